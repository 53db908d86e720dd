#!/bin/bash
# Evaluation cache: parity tests, then a same-box comparison.  usage: tools/gpu_round_ec.sh [tag]
TAG=${1:-r02ec}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
( timeout 1500 python -m pytest tests -m gpu -q -s -x -p no:cacheprovider --timeout 900 ${PYTEST_K:+-k "$PYTEST_K"} > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
grep -E "evaluation cache|real net|passed|failed|Error|rc=|^E " $OUT/pytest_gpu.log | cut -c1-240 | tail -20
B="timeout 600 python bench.py --no-cpu-baseline"
[ -n "$SKIP_BENCH" ] && exit 0
$B > $OUT/bench_a_te4.json 2> $OUT/bench_a.err
$B --eval-cache > $OUT/bench_b_te4_cache.json 2> $OUT/bench_b.err
$B --eval-cache --terminal-extra 8 > $OUT/bench_c_te8_cache.json 2> $OUT/bench_c.err
$B > $OUT/bench_d_te4.json 2> $OUT/bench_d.err
$B --eval-cache --steps 4800 > $OUT/bench_e_te4_cache_3plies.json 2> $OUT/bench_e.err
$B --steps 4800 > $OUT/bench_f_te4_3plies.json 2> $OUT/bench_f.err
$B --selfplay --start-position --eval-cache --steps 4800 > $OUT/bench_g_selfplay_cache.json 2> $OUT/bench_g.err
$B --selfplay --start-position --steps 4800 > $OUT/bench_h_selfplay.json 2> $OUT/bench_h.err
for f in $OUT/bench_*.json; do echo "== $f"; python tools/jline.py $f 2>&1 | head -14 | grep -v "per rank"; done
for f in $OUT/*.err; do grep -v amdgpu.ids $f | tail -n 3 | cut -c1-300; done
