#!/bin/bash
# GPU box, round 6 session 2: the tests touched since session 1, then the small-batch latency lines and the rules kernels
O=gpurun_out/r06c; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "test_hip_rules or test_selfplay_device or test_multirank or test_hip_search or test_abi" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -4 $O/pytest_gpu.log
for po in 400 1600; do
  timeout 600 python bench.py --games 1 --search-threads 16 --playout $po --steps 400 --warmup 16 --age-steps 64 --steady-steps 1000 --alt-steps 200 --no-cpu-baseline > $O/latency_1x16_p$po.json 2> $O/latency_1x16_p$po.err
  python - <<PY
import json
l=[x for x in open("$O/latency_1x16_p$po.json") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); c=d["config"]
    print("playout $po: engine %s  %.0f sims/s  %.3f ms per lock-step  %.2f ms per move (%s launches per lock-step, %.1f lock-steps per move); fast engine leg %s sims/s" % (d["dtype"], d["value"], d["ms_per_step"], c["ms_per_move"], c["launches_per_lock_step"], c["lock_steps_per_move"], d.get("fast_engine") and round(d["fast_engine"]["value"])))
else:
    print("no line:", open("$O/latency_1x16_p$po.err").read()[-600:])
PY
done
timeout 600 python tools/rules_bench.py > $O/rules_bench.log 2>&1; tail -14 $O/rules_bench.log
