#!/bin/bash
# Round 3's full GPU session: all parity tests, smoke, every bench line quoted in DESIGN.md / BASELINE.md, rocprofv3 stats + HBM PMC.
# usage: tools/gpu_round3_final.sh [tag]   (outputs: gpurun_out/<tag>/)
TAG=${1:-r03f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
tail -3 $OUT/pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
( timeout 300 python tools/rules_bench.py > $OUT/rules_bench.log 2>&1 )
for a in "1 20000 0 0 0" "1 20000 0 0 1" "1 20000 2 0 0" "1 20000 2 0 1" "1 20000 0 0 0" "1 20000 0 0 1"; do ./tools/ubench/mfma_peak $a | tail -1; done > $OUT/mfma_peak_f16.log 2>&1
B="timeout 600 python bench.py"
$B > $OUT/bench_default.json 2> $OUT/bench_default.err
$B --steps 20 --warmup 5 > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err
$B --dtype bf16 --no-cpu-baseline > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
$B --eval-cache --no-cpu-baseline > $OUT/bench_cache.json 2> $OUT/bench_cache.err
$B --games 4096 --playout 400 --steps 1200 --age-steps 300 --no-cpu-baseline > $OUT/bench_cfg1_4096x400.json 2> $OUT/bench_cfg1.err
$B --games 4096 --playout 400 --steps 1200 --age-steps 300 --dtype bf16 --no-cpu-baseline > $OUT/bench_cfg1_4096x400_bf16.json 2> $OUT/bench_cfg1_bf16.err
$B --blocks 19 --steps 300 --steady-steps 300 --no-cpu-baseline > $OUT/bench_19blk_fp16.json 2> $OUT/bench_19blk.err
$B --terminal-extra 0 --no-cpu-baseline > $OUT/bench_default_te0.json 2> $OUT/bench_default_te0.err
$B --selfplay --steps 6400 --warmup 16 --age-steps 0 --steady-steps 0 --no-cpu-baseline > $OUT/bench_selfplay_p1600.json 2> $OUT/bench_selfplay_p1600.err
$B --selfplay --playout 100 --steps 20000 --warmup 64 --advance-every 4 --age-steps 0 --steady-steps 0 --no-cpu-baseline > $OUT/bench_selfplay_p100.json 2> $OUT/bench_selfplay_p100.err
$B --playout 100 --steps 20000 --warmup 64 --advance-every 4 --age-steps 100 --steady-steps 0 --no-cpu-baseline > $OUT/bench_search_p100.json 2> $OUT/bench_search_p100.err
ROOT=$(pwd); mkdir -p $OUT/train && cd $OUT/train && ( SECONDS=0; timeout 900 python $ROOT/main.py --mode train --games 2048 --train_playout 100 --batch_size 512 --res_block_nums 7 --processor gpu --max_batches 3 > train.log 2> train.err; echo "wall seconds: $SECONDS" >> train.log ); cd $ROOT; rm -rf $OUT/train/gpu_models $OUT/train/models* 2>/dev/null; tail -n 5 $OUT/train/train.log | cut -c1-220
for f in $OUT/bench_*.json; do echo "== $f"; python tools/jline.py $f 2>&1 | head -9; done
bash tools/profile_round.sh $TAG/prof > $OUT/profile_round.log 2>&1
cat $OUT/rules_bench.log $OUT/mfma_peak_f16.log
