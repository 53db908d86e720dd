#!/bin/bash
# GPU box: whole-game self-play (160 000 lock-steps from the start position, playout 1600, 8192 games, fast fp16 engine as in round 5's
# r05t runs) without a cache, and with both cache levels at 2^22 / 2^24 entries under the round-6 replacement rule
O=gpurun_out/${1:-r06d}; mkdir -p $O
STEPS=${2:-160000}
run() { n=$1; shift
  timeout 1500 python bench.py --selfplay --start-position --dtype fp16 --steps $STEPS --warmup 16 --age-steps 0 --steady-steps 0 --alt-steps 0 --no-cpu-baseline "$@" > $O/sp_p1600_${STEPS}_$n.json 2> $O/sp_p1600_${STEPS}_$n.err
  python - <<PY
import json
l=[x for x in open("$O/sp_p1600_${STEPS}_$n.json") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); c=d["config"]; sp=c.get("selfplay") or {}
    print("$n: %.0f sims/s  %.3f ms/step  %.3f sims per net row  games %s records %s" % (d["value"], d["ms_per_step"], c["simulations_per_net_row"], sp.get("games_finished"), sp.get("records")))
else:
    print("$n: no line", open("$O/sp_p1600_${STEPS}_$n.err").read()[-500:])
PY
  cp gpurun_out/bench_detail_fp16_n1.json $O/detail_$n.json 2>/dev/null
}
run nocache
run xcache22 --eval-cache --xcache 22
run xcache24 --eval-cache --xcache 24
python - <<PY
import json
for n in ("xcache22", "xcache24"):
    try:
        d=json.load(open("$O/detail_%s.json" % n)); print(n, "eval_cache", d["config"]["eval_cache"], "xcache", d["config"]["xcache"])
    except Exception as e: print(n, e)
PY
