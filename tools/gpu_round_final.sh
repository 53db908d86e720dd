#!/bin/bash
# The round's full GPU session: all parity tests, every bench line quoted in DESIGN.md / BASELINE.md, rocprofv3 stats, HBM PMC,
# SQ PMC of the trunk and of the tree kernels.  usage: tools/gpu_round_final.sh [tag]   (outputs: gpurun_out/<tag>/)
TAG=${1:-r02f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
( timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
tail -3 $OUT/pytest_gpu.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
( timeout 300 python tools/rules_bench.py > $OUT/rules_bench.log 2>&1 )
B="timeout 600 python bench.py"
$B > $OUT/bench_default.json 2> $OUT/bench_default.err
$B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err
$B --games 4096 --playout 400 --steps 1200 --no-cpu-baseline > $OUT/bench_cfg1_4096x400.json 2> $OUT/bench_cfg1.err
$B --blocks 19 --dtype fp16 --steps 300 --no-cpu-baseline > $OUT/bench_19blk_fp16.json 2> $OUT/bench_19blk.err
$B --terminal-extra 0 --no-cpu-baseline > $OUT/bench_default_te0.json 2> $OUT/bench_default_te0.err
$B --selfplay --playout 100 --steps 20000 --warmup 64 --advance-every 4 --no-cpu-baseline > $OUT/bench_selfplay_p100.json 2> $OUT/bench_selfplay_p100.err
$B --playout 100 --steps 20000 --warmup 64 --advance-every 4 --no-cpu-baseline > $OUT/bench_search_p100.json 2> $OUT/bench_search_p100.err
$B --selfplay --start-position --playout 100 --steps 20000 --warmup 64 --advance-every 4 --no-cpu-baseline > $OUT/bench_selfplay_p100_startpos.json 2> $OUT/bench_selfplay_p100_startpos.err
$B --selfplay --steps 6400 --warmup 16 --no-cpu-baseline > $OUT/bench_selfplay_p1600.json 2> $OUT/bench_selfplay_p1600.err
$B --gpus 2 --all-on-device0 --dist-backend gloo --games 1024 --selfplay --timed-gather --playout 40 --steps 4800 --warmup 64 --advance-every 4 > $OUT/bench_2ranks_selfplay_gather.json 2> $OUT/bench_2ranks_selfplay_gather.err
$B --gpus 2 --all-on-device0 --dist-backend gloo --games 2048 --steps 100 --warmup 8 > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks_gloo.err
$B --force-dist --dist-backend nccl --games 1024 --selfplay --timed-gather --playout 40 --steps 1600 --warmup 64 --advance-every 4 --no-cpu-baseline > $OUT/bench_rccl_world1_selfplay_gather.json 2> $OUT/bench_rccl_world1.err
ROOT=$(pwd); mkdir -p $OUT/train && cd $OUT/train && ( SECONDS=0; timeout 900 python $ROOT/main.py --mode train --games 2048 --train_playout 100 --batch_size 512 --res_block_nums 7 --processor gpu --max_batches 2 > train.log 2> train.err; echo "wall seconds: $SECONDS" >> train.log ); cd $ROOT; rm -rf $OUT/train/gpu_models $OUT/train/models* 2>/dev/null; tail -n 4 $OUT/train/train.log | cut -c1-200
for f in $OUT/bench_*.json; do echo "== $f"; python tools/jline.py $f 2>&1 | head -12; done
bash tools/profile_round.sh $TAG/prof > $OUT/profile_round.log 2>&1
bash tools/pmc_ubench.sh 8 > $OUT/pmc_ubench8.log 2>&1; cp gpurun_out/pmc_sq_8.json $OUT/ 2>/dev/null; tail -7 $OUT/pmc_ubench8.log | cut -c1-300
bash tools/pmc_tree.sh $OUT/pmc_tree > $OUT/pmc_tree.log 2>&1; tail -6 $OUT/pmc_tree.log | cut -c1-400
cat $OUT/rules_bench.log
