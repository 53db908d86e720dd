#!/usr/bin/env python3
"""Print selected fields of a bench.py JSON line: tools/jline.py [file] (default: stdin)"""
import json
import sys
src = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
lines = [l for l in src.strip().splitlines() if l.startswith("{")]
d = json.loads(lines[-1])
if "detail_file" in d:   # the compact line of round 5: the complete record sits in the side file
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for cand in (d["detail_file"], os.path.join(root, d["detail_file"]), os.path.join(os.path.dirname(sys.argv[1]) if len(sys.argv) > 1 else ".", os.path.basename(d["detail_file"]))):
        if os.path.exists(cand):
            d = json.load(open(cand))
            break
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
    ol = rr.get("ordered_list_kernel")
    if ol:
        print("    ordered-list kernel: %.2f G positions/s (frac %.3f), ABI %.0f GB/s" % (ol["positions_per_s"] / 1e9, ol["frac"], ol.get("abi_GBps", float("nan"))))
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
cs = d.get("contract_steps")
if cs:
    print("  value from %s; contract steps: %.0f sims/s over %d steps (%.3f ms/step)" % (d.get("value_source"), cs["value"], cs["steps"], cs["ms_per_step"]))
if "effective_clock_GHz" in r:
    print("  trunk clock %.3f GHz (in-kernel stamps, %d launches), clock-scaled peak %.0f TF -> frac %.3f (MFMA issue %.3f); power %s W, driver sclk %s MHz (%s samples)" % (
        r["effective_clock_GHz"], r["clock_probe"]["launches_probed"], r["clock_scaled_peak"], r["frac_of_clock_scaled_peak"], r["mfma_issue_frac_of_clock_scaled_peak"],
        r.get("power_W"), r.get("sclk_MHz_driver"), (r.get("telemetry") or {}).get("samples")))
mp = r.get("mfma_peak_measured")
if mp and "dense_random_operands" in mp:
    print("  measured MFMA ceiling of this box: %.0f TF dense random, %.0f TF half zeros" % (mp["dense_random_operands"]["tflops"], mp["half_zero_operands"]["tflops"]))
elif mp:
    print("  mfma_peak_measured:", mp)
se = d.get("strict_engine")
if se:
    print("  strict engine: %.0f sims/s over %d steps (%.3f ms/step), target met: %s" % (se["value"], se["steps"], se["ms_per_step"], se["meets_target_1e6_sims_per_s_per_gpu"]))
    if "roofline" in se:
        print("    trunk %.1f us, %.0f TF algorithmic (frac %.3f)" % (se["roofline"]["us_per_launch"], se["roofline"]["achieved"], se["roofline"]["frac"]))
    for k, e in (se.get("net_error") or {}).items():
        if isinstance(e, dict):
            print("    net_error %-22s dlogit %.3g dvalue %.3g argmax %.3f" % (k, e["dlogit"], e["dvalue"], e["argmax_agree"]))
        else:
            print("    %s: %s" % (k, e))
