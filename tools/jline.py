#!/usr/bin/env python3
"""Print selected fields of a bench.py JSON line: tools/jline.py [file] (default: stdin)"""
import json
import sys
src = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
lines = [l for l in src.strip().splitlines() if l.startswith("{")]
d = json.loads(lines[-1])
r = d["roofline"]
print("%.0f sims/s  %.3f ms/step  n_gpus %d  trunk %.1f us  %.1f TF/s (frac %.3f)  errors %s" % (
    d["value"], d["ms_per_step"], d["n_gpus"], r.get("us_per_launch", float("nan")), r["achieved"], r["frac"], d["config"].get("trees_with_error_status")))
ss = d.get("steady_state")
if ss:
    print("  steady state: %.0f sims/s over %d steps (%.3f ms/step, %.0f net rows/s, %.3f sims per row)" % (ss["value"], ss["steps"], ss["ms_per_step"], ss["net_rows_per_s"], ss["simulations_per_net_row"]))
ne = d.get("net_error")
if ne and "as_benchmarked_glorot" in ne:
    for k in ("as_benchmarked_glorot", "trained_like"):
        e = ne[k]
        print("  net_error %-22s dlogit %.3g (rel %.3g of %.3g) dsoftmax %.3g dvalue %.3g argmax %.3f" % (k, e["dlogit"], e["dlogit_rel"], e["max_abs_logit"], e["dsoftmax"], e["dvalue"], e["argmax_agree"]))
elif ne:
    print("  net_error:", ne)
rr = d.get("roofline_rules")
if rr and "achieved" in rr:
    print("  rules K1: %.2f G positions/s, %.0f GB/s algorithmic (frac %.3f), ABI %.0f GB/s" % (rr["positions_per_s"] / 1e9, rr["achieved"], rr["frac"], rr["abi_GBps"]))
t = d.get("roofline_tree")
if t:
    print("  tree side: select %.1f us, expand %.1f us, %.0f GB/s (frac %.4f)" % (t["us_select"], t["us_expand_backup"], t["achieved"], t["frac"]))
c = d.get("cpu_baseline")
if c:
    print("  cpu: %s sims/s on %s cores; single core %s" % (c.get("value"), c.get("cores"), (c.get("single_core") or {}).get("value")))
sp = d["config"].get("selfplay")
if sp:
    print("  selfplay:", sp)
print("  sims per net row %s  terminal_extra %s  eval_cache %s  net rows/s %s" % (d["config"].get("simulations_per_net_row"), d["config"].get("terminal_extra"), d["config"].get("eval_cache"), d["config"].get("net_rows_per_s")))
print("  per rank:", d["config"].get("per_rank_sims_per_s"), "backend", d["config"].get("dist_backend"), "gather", d["config"].get("record_gather"),
      "status", d["config"].get("status_bits"))
