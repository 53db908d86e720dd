#!/bin/bash
# Final check of the round: full GPU suite + the evaluation-cache lines quoted in DESIGN.md.  usage: tools/gpu_round_cache.sh [tag]
TAG=${1:-r02q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
( timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
grep -E "evaluation cache|real net|passed|failed|Error|rc=|^E " $OUT/pytest_gpu.log | cut -c1-240 | tail -12
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -n 1 $OUT/smoke.log
B="timeout 600 python bench.py --no-cpu-baseline"
$B --eval-cache > $OUT/bench_cache_default.json 2> $OUT/bench_a.err
$B > $OUT/bench_nocache_default.json 2> $OUT/bench_b.err
$B --eval-cache --steps 6400 > $OUT/bench_cache_4plies.json 2> $OUT/bench_c.err
$B --steps 6400 > $OUT/bench_nocache_4plies.json 2> $OUT/bench_d.err
$B --selfplay --eval-cache --playout 100 --steps 20000 --warmup 64 --advance-every 4 > $OUT/bench_selfplay_p100_cache.json 2> $OUT/bench_e.err
$B --selfplay --playout 100 --steps 20000 --warmup 64 --advance-every 4 > $OUT/bench_selfplay_p100_nocache.json 2> $OUT/bench_f.err
$B --selfplay --start-position --eval-cache --playout 400 --steps 20000 --warmup 64 > $OUT/bench_selfplay_startpos_p400_cache.json 2> $OUT/bench_g.err
$B --selfplay --start-position --playout 400 --steps 20000 --warmup 64 > $OUT/bench_selfplay_startpos_p400_nocache.json 2> $OUT/bench_h.err
ROOT=$(pwd); mkdir -p $OUT/train && cd $OUT/train && ( SECONDS=0; timeout 900 python $ROOT/main.py --mode train --games 2048 --train_playout 100 --batch_size 512 --res_block_nums 7 --processor gpu --max_batches 2 > train.log 2> train.err; echo "wall seconds: $SECONDS" >> train.log ); cd $ROOT; rm -rf $OUT/train/gpu_models $OUT/train/models* 2>/dev/null; grep "batch i\|wall" $OUT/train/train.log | cut -c1-200
for f in $OUT/bench_*.json; do echo "== $f"; python tools/jline.py $f 2>&1 | head -14 | grep -v "per rank"; done
for f in $OUT/*.err; do grep -v amdgpu.ids $f | tail -n 3 | cut -c1-300; done
