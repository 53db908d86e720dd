#!/bin/bash
# round 3, call c: GPU tests, bench with the LDS advance kernel (A/B against the global-memory one), rocprof stats + HBM PMC passes
set -u
O=gpurun_out/r03c; mkdir -p $O
python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
B="python bench.py --no-cpu-baseline"
$B > $O/bench_a.json 2>$O/err1; python tools/jline.py $O/bench_a.json | head -3
CCHESS_ADVANCE_GLOBAL=1 $B > $O/bench_adv_global.json 2>/dev/null; python tools/jline.py $O/bench_adv_global.json | head -2
$B > $O/bench_b.json 2>/dev/null; python tools/jline.py $O/bench_b.json | head -2
bash tools/profile_round.sh r03c/prof > $O/profile.log 2>&1; tail -5 $O/profile.log
