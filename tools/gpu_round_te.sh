#!/bin/bash
# Terminal simulations inside select + asynchronous plies: parity tests, then a same-box sweep of --terminal-extra.
# usage: tools/gpu_round_te.sh [tag]
TAG=${1:-r02te}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
[ -n "$SKIP_TESTS" ] || ( timeout 1500 python -m pytest tests -m gpu -q -s -x -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
grep -E "terminal_extra|asynchronous plies|passed|failed|Error|rc=" $OUT/pytest_gpu.log | cut -c1-240 | tail -20
B="timeout 600 python bench.py --no-cpu-baseline"
for te in ${TE_SWEEP:-0 2 4 8 16 64 0 8}; do
  $B --terminal-extra $te > $OUT/bench_te${te}_$SECONDS.json 2> $OUT/bench_te${te}.err
done
$B --terminal-extra 8 --playout 100 --steps 3000 --warmup 16 > $OUT/bench_search_p100_te8.json 2> $OUT/bench_search_p100.err
$B --terminal-extra 0 --playout 100 --steps 3000 --warmup 16 > $OUT/bench_search_p100_te0.json 2>> $OUT/bench_search_p100.err
$B --selfplay --playout 100 --steps 3000 --warmup 16 --terminal-extra 8 --advance-every 4 > $OUT/bench_selfplay_p100_te8.json 2> $OUT/bench_selfplay_p100.err
$B --selfplay --playout 100 --steps 3000 --warmup 16 --terminal-extra 0 --advance-every 4 > $OUT/bench_selfplay_p100_te0.json 2>> $OUT/bench_selfplay_p100.err
$B --selfplay --start-position --steps 3200 --warmup 16 --terminal-extra 8 > $OUT/bench_selfplay_startpos_p1600_te8.json 2> $OUT/bench_selfplay_p1600.err
$B --selfplay --start-position --steps 3200 --warmup 16 --terminal-extra 0 > $OUT/bench_selfplay_startpos_p1600_te0.json 2>> $OUT/bench_selfplay_p1600.err
for f in $OUT/bench_*.json; do echo "== $f"; python tools/jline.py $f 2>&1 | head -14; done
for f in $OUT/*.err; do tail -n 3 $f | cut -c1-300; done
