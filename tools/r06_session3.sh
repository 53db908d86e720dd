#!/bin/bash
# GPU box, round 6 session 3: the whole GPU suite at HEAD, the default bench line, the --mode play shape with and without the captured
# graph, the 19-block line, rocprofv3 kernel stats + HBM PMC passes of the default (strict) bench
O=gpurun_out/r06f; mkdir -p $O
( timeout 1700 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -4 $O/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log ); tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/jline.py $O/bench_default.json | tee $O/bench_default.txt | head -14; cp gpurun_out/bench_detail_mx6_n1.json $O/bench_default_detail.json
for po in 400 1600; do for gr in "" "--graph"; do
  n=latency_1x16_p${po}${gr:+_graph}
  timeout 600 python bench.py --games 1 --search-threads 16 --playout $po --steps 400 --warmup 16 --age-steps 64 --steady-steps 1000 --alt-steps 0 --no-cpu-baseline $gr > $O/$n.json 2> $O/$n.err
  python - <<PY
import json
l=[x for x in open("$O/$n.json") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); c=d["config"]
    print("$n: engine %s  %.0f sims/s  %.3f ms per lock-step  %.2f ms per move (%s launches per lock-step, %.1f lock-steps per move) trunk %.0f us" % (d["dtype"], d["value"], d["ms_per_step"], c["ms_per_move"], c["launches_per_lock_step"], c["lock_steps_per_move"], d["roofline"]["us_per_launch"]))
else:
    print("$n: no line:", open("$O/$n.err").read()[-400:])
PY
done; done
timeout 900 python bench.py --blocks 19 --no-cpu-baseline > $O/bench_19_blocks.json 2> $O/bench_19_blocks.err; python tools/jline.py $O/bench_19_blocks.json | head -4; cp gpurun_out/bench_detail_*_n1.json $O/ 2>/dev/null
tools/gpu_session.sh r06f profile:mx6
