#!/bin/bash
# One GPU-box session of a round: parity tests, measured net errors, the bench lines.  Everything lands in gpurun_out/.
# usage: tools/gpu_round.sh [tag]
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
( timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
tail -5 $OUT/pytest_gpu.log
( timeout 600 python tests/measure_net_errors.py $OUT/net_errors.json > $OUT/net_errors.log 2>&1; echo "rc=$?" >> $OUT/net_errors.log )
( timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err )
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err )
( timeout 600 python bench.py --selfplay --playout 100 --steps 300 --warmup 10 --no-cpu-baseline > $OUT/bench_selfplay_p100.json 2> $OUT/bench_selfplay_p100.err; echo "rc=$?" >> $OUT/bench_selfplay_p100.err )
( timeout 300 python bench.py --playout 100 --steps 3000 --warmup 16 --no-cpu-baseline > $OUT/bench_search_p100.json 2> $OUT/bench_search_p100.err )
( timeout 600 python bench.py --gpus 2 --all-on-device0 --dist-backend gloo --games 2048 --steps 100 --warmup 8 > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks_gloo.err; echo "rc=$?" >> $OUT/bench_2ranks_gloo.err )
( timeout 600 python bench.py --gpus 2 --all-on-device0 --dist-backend gloo --games 1024 --selfplay --timed-gather --playout 40 --steps 120 --warmup 4 > $OUT/bench_2ranks_selfplay_gather.json 2> $OUT/bench_2ranks_selfplay_gather.err; echo "rc=$?" >> $OUT/bench_2ranks_selfplay_gather.err )
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
tail -2 $OUT/smoke.log
for f in $OUT/bench_*.json; do echo "== $f"; python tools/jline.py $f 2>&1 | head -12; done
