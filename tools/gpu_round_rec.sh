#!/bin/bash
# After the per-tree-record refactor: full GPU suite, then cache / no-cache bench lines.  usage: tools/gpu_round_rec.sh [tag]
TAG=${1:-r02r}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
( timeout 1500 python -m pytest tests -m gpu -q -s -x -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
grep -E "passed|failed|Error|rc=|^E " $OUT/pytest_gpu.log | cut -c1-240 | tail -12
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -n 1 $OUT/smoke.log
B="timeout 600 python bench.py --no-cpu-baseline"
$B > $OUT/bench_nocache_default.json 2> $OUT/bench_a.err
$B --eval-cache > $OUT/bench_cache_default.json 2> $OUT/bench_b.err
$B --terminal-extra 0 > $OUT/bench_te0.json 2> $OUT/bench_c.err
$B --eval-cache --terminal-extra 8 > $OUT/bench_cache_te8.json 2> $OUT/bench_d.err
for f in $OUT/bench_*.json; do echo "== $f"; python tools/jline.py $f 2>&1 | head -14 | grep -v "per rank"; done
for f in $OUT/*.err; do grep -v amdgpu.ids $f | tail -n 3 | cut -c1-300; done
