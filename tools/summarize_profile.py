#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/<tag>/) into the committed
artefacts under profiles/: <round>_kernel_stats.csv, <round>_summary.md, <round>_bench_under_rocprof.json
and pmc_traffic.json (the file bench.py attaches as roofline.traffic when its configuration matches).

usage: tools/summarize_profile.py gpurun_out/prof r01b [kernel-substring [suffix]]
       suffix (e.g. "_strict"): pmc_traffic<suffix>.json / pmc_tree_traffic<suffix>.json (bench.py looks the strict engine's up there)
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, pat):
    hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
    if not hits:
        raise SystemExit("no %s under %s" % (pat, d))
    return hits[0]


def pmc_mean(path, counter, kernel_sub):
    vals = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and kernel_sub in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    ksub = sys.argv[3] if len(sys.argv) > 3 else "k_tower8_c128"
    sfx = sys.argv[4] if len(sys.argv) > 4 else ""
    prof = os.environ.get("PROFILES_OUT") or os.path.join(ROOT, "profiles")   # PROFILES_OUT: summarise on the GPU box into gpurun_out/
    os.makedirs(prof, exist_ok=True)
    stats = find(os.path.join(src, "stats"), "*kernel_stats.csv")
    shutil.copy(stats, os.path.join(prof, rnd + "_kernel_stats.csv"))
    bline = None
    for ln in open(os.path.join(src, "bench_under_rocprof.json")):
        if ln.startswith("{"):
            bline = json.loads(ln)
    json.dump(bline, open(os.path.join(prof, rnd + "_bench_under_rocprof.json"), "w"), indent=1)
    rows = list(csv.DictReader(open(stats)))
    dom = [r for r in rows if ksub in r["Name"]][0]
    md = ["# %s — rocprofv3 --kernel-trace --stats of the default bench" % rnd, "",
          "Command on the MI355X box (tools/profile_round.sh): `rocprofv3 --kernel-trace --stats --output-format csv -- "
          "python bench.py --no-cpu-baseline --strict-steps 0 --steps 100 --warmup 8 --steady-steps 0` (dtype %s)" % bline["dtype"], "",
          "| kernel | calls | avg us | % of GPU time |", "|---|---|---|---|"]
    for r in rows[:12]:
        md.append("| `%s` | %s | %.1f | %s |" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    md += ["", "Dominant kernel `%s`: rocprof average %.1f us vs bench.py's live HIP-event average in the same run %.1f us "
           "(roofline.achieved %.1f TFLOP/s, frac %.3f; whole-job %.0f sims/s under the profiler)."
           % (ksub, float(dom["AverageNs"]) / 1e3, bline["roofline"]["us_per_launch"], bline["roofline"]["achieved"],
              bline["roofline"]["frac"], bline["value"])]
    # the same average over the launches of the TIMED region only: the stats line above mixes in the 800 untimed ageing steps and
    # the small net-error probes.  (Under the profiler the timed region is 100 steps = 0.23 s right after a host-side pause:
    # the power governor lets the first ~0.3 s after an idle gap run 3-4 % faster than the settled rate the ageing launches
    # show — which is why bench.py's headline is its 2000-step leg, not a short one.)
    try:
        tr = []
        with open(find(os.path.join(src, "stats"), "*kernel_trace.csv")) as f:
            for r in csv.DictReader(f):
                if ksub in r["Kernel_Name"]:
                    tr.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"])))
        tr.sort()
        full = max(t[2] for t in tr)
        big = [t for t in tr if t[2] == full]
        # bench order: ageing, warm-up, K timed steps, then the MFMA probe / net-error probes: the timed launches are the last K
        # full-size ones that are followed by no other full-size launch of the loop
        K = int(bline["steps"])
        timed = big[-K:]
        md += ["", "Over the %d launches of the timed region alone (kernel trace, the last %d full-batch launches): **%.1f us** "
               "(min %.1f, max %.1f); the %d ageing / warm-up launches before them average %.1f us (the settled rate of this box: the 100 timed steps under the profiler are 0.23 s right after a host-side pause, which the power governor lets run faster)."
               % (len(timed), K, sum(t[1] for t in timed) / len(timed) / 1e3, min(t[1] for t in timed) / 1e3, max(t[1] for t in timed) / 1e3,
                  len(big) - len(timed), sum(t[1] for t in big[:-K]) / max(1, len(big) - K) / 1e3)]
    except Exception as e:   # older outputs without the trace
        md += ["", "(no per-launch trace: %s)" % e]
    fetch, nf = pmc_mean(find(os.path.join(src, "pmc_f"), "*counter_collection.csv"), "FETCH_SIZE", ksub)
    write, nw = pmc_mean(find(os.path.join(src, "pmc_w"), "*counter_collection.csv"), "WRITE_SIZE", ksub)
    cfg = bline["config"]
    G, blocks = cfg["games_per_gpu"], cfg["res_block_nums"]
    # algorithmic bytes of one launch: planes in (bf16 [G][90][16]) + all folded weights once
    # (first conv 9*16*128, 2*blocks layers of 9*128*128, bf16) + biases (f32) + head conv output (f32 [G][90][3])
    rows_ = cfg.get("net_rows_per_step", G)   # compact batches: fewer rows than trees
    halves = 2 if bline["dtype"].endswith("x2") else 1   # the x3 strict engine streams every weight as hi + lo
    wbytes = (9 * 16 * 128 + 2 * blocks * 9 * 128 * 128) * 2 * halves
    if bline["dtype"] == "mx6":   # first layer hi + lo fp16; tower: 36 slabs per layer of fp16 hi (8 KB) + fp6 blocks (6 KB) + scale dwords (1 KB)
        wbytes = 9 * 16 * 128 * 2 * 2 + 2 * blocks * 36 * 15360
    alg = int(rows_ * 90 * 16 * 2 + wbytes + (2 * blocks + 1) * 128 * 4 + rows_ * 90 * 3 * 4)
    tj = {"kernel": ksub, "config": {"B": G, "res_block_nums": blocks, "dtype": bline["dtype"], "compact": bool(cfg.get("compact_batches", False)),
                                     "net_rows_per_step": rows_},
          "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes over `python bench.py "
                    "--steps 40 --warmup 4 --no-cpu-baseline` (tools/profile_round.sh); mean over %d/%d launches; counters are in KiB" % (nf, nw),
          "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
          "fetch_bytes_corrected_x2": fetch * 1024 * 2, "write_bytes": write * 1024,
          "traffic_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
          "algorithmic_bytes_per_launch": alg,
          "note": "gfx950 correction from MI355X_MICROARCH.md (FETCH_SIZE reports 1/2 of wide coalesced reads; uncalibrated for the "
                  "LDS-DMA pattern, so the truth lies between x1 and x2). WRITE_SIZE equals the head conv outputs only: the trunk never "
                  "leaves the CU. The read side is far above the algorithmic input because every workgroup streams the whole folded weight "
                  "set (L2 / Infinity Cache hits for all but the first sweep)."}
    json.dump(tj, open(os.path.join(prof, "pmc_traffic%s.json" % sfx), "w"), indent=1)
    md += ["", "HBM PMC (separate passes): FETCH_SIZE %.0f KiB raw, WRITE_SIZE %.0f KiB per launch -> traffic %.3e B "
           "(algorithmic %.3e B); see pmc_traffic%s.json." % (fetch, write, tj["traffic_bytes_per_launch"], alg, sfx)]
    # the HBM-bound tree / rules kernels of the same runs: counter traffic per launch and the GB/s that implies
    dur = {r["Name"]: float(r["AverageNs"]) for r in rows}
    tree = {}
    fpath = find(os.path.join(src, "pmc_f"), "*counter_collection.csv")
    wpath = find(os.path.join(src, "pmc_w"), "*counter_collection.csv")
    for key, sub in (("k_select", "k_select"), ("k_expand_backup", "k_expand_backup"), ("k_value_fc", "k_value_fc")):
        f_, n1 = pmc_mean(fpath, "FETCH_SIZE", sub)
        w_, n2 = pmc_mean(wpath, "WRITE_SIZE", sub)
        ns = [v for k, v in dur.items() if sub in k]
        if f_ is None or w_ is None or not ns:
            continue
        t_ns = ns[0]
        tree[key] = {"FETCH_SIZE_KiB_raw": f_, "WRITE_SIZE_KiB_raw": w_, "fetch_bytes_x1": f_ * 1024, "fetch_bytes_x2": f_ * 2048,
                     "write_bytes": w_ * 1024, "avg_us": t_ns / 1e3, "launches": [n1, n2],
                     "GBps_x1": (f_ * 1024 + w_ * 1024) / t_ns, "GBps_x2": (f_ * 2048 + w_ * 1024) / t_ns}
    if tree:
        tj2 = {"config": tj["config"], "kernels": tree,
               "method": "same two rocprofv3 --pmc passes as pmc_traffic.json; avg_us from the --kernel-trace --stats pass; FETCH_SIZE is given "
                         "raw (x1) and with the gfx950 x2 correction for wide coalesced reads (these kernels mix narrow gathers and wide "
                         "loads: the truth lies between)"}
        json.dump(tj2, open(os.path.join(prof, "pmc_tree_traffic%s.json" % sfx), "w"), indent=1)
        md += ["", "Tree / rules kernels (HBM-bound; counter traffic per launch, x1 .. x2 FETCH correction):", "",
               "| kernel | avg us | fetch KiB raw | write KiB | GB/s (x1 .. x2) |", "|---|---|---|---|---|"]
        for k, v in tree.items():
            md.append("| `%s` | %.1f | %.0f | %.0f | %.0f .. %.0f |" % (k, v["avg_us"], v["FETCH_SIZE_KiB_raw"], v["WRITE_SIZE_KiB_raw"], v["GBps_x1"], v["GBps_x2"]))
    open(os.path.join(prof, rnd + "_summary.md"), "w").write("\n".join(md) + "\n")
    print("\n".join(md))


if __name__ == "__main__":
    main()
