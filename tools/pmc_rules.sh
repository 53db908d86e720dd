#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the stand-alone rules kernels over tools/rules_bench.py (1 M positions), two PMC passes,
# kernel-trace only.  usage: tools/pmc_rules.sh <outdir>   -> <outdir>/pmc_rules.txt (per kernel: counters per wave of 64 positions)
OUT=$(pwd)/$1; ROOT=$(pwd); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
B="SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
(timeout 200 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $OUT/a -o p -- python $ROOT/tools/rules_bench.py > $OUT/a.out 2>&1) < /dev/null
(timeout 200 rocprofv3 --kernel-trace --pmc $B --output-format csv -d $OUT/b -o p -- python $ROOT/tools/rules_bench.py > $OUT/b.out 2>&1) < /dev/null
# HBM-side traffic: FETCH_SIZE and WRITE_SIZE in their own passes (MI355X_MICROARCH.md)
(timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o p -- python $ROOT/tools/rules_bench.py > $OUT/f.out 2>&1) < /dev/null
(timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o p -- python $ROOT/tools/rules_bench.py > $OUT/w.out 2>&1) < /dev/null
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in ("a", "b", "f", "w"):
    for f in glob.glob(out + "/" + d + "/**/p_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            for key in ("k_movegen_mask", "k_movegen_list<true, true>", "k_movegen_list<true, false>", "k_movegen_list<false, true>", "k_movegen_list<false, false>", "k_hash"):
                if key in k:
                    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
traffic = {}
with open(out + "/pmc_rules.txt", "w") as fo:
    for key, c in acc.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        groups = (1 << 20) / 64.0
        us = sum(dur[key]) / len(dur[key]) / 1e3
        line = ["%s: %.1f us per 1 M positions; waves launched %.0f" % (key, us, m.get("SQ_WAVES", 0))]
        line.append("  per group of 64 positions: VALU %.0f SALU %.0f LDS %.0f VMEM %.0f instructions" % (m["SQ_INSTS_VALU"] / groups, m["SQ_INSTS_SALU"] / groups, m["SQ_INSTS_LDS"] / groups, m["SQ_INSTS_VMEM"] / groups))
        wc = m["SQ_WAVE_CYCLES"] * 4 / groups
        line.append("  wave-cycles per group %.0f = active %.0f + issue-stalled %.0f + parked (s_waitcnt) %.0f; LDS-issue-stall %.0f" % (wc, m["SQ_ACTIVE_INST_ANY"] * 4 / groups, m["SQ_WAIT_INST_ANY"] * 4 / groups, m["SQ_WAIT_ANY"] * 4 / groups, m["SQ_WAIT_INST_LDS"] * 4 / groups))
        line.append("  active VALU %.0f, active LDS %.0f quad-cycles x4 per group; LDS idx active %.0f, bank conflicts %.0f; clock %.2f GHz" % (m["SQ_ACTIVE_INST_VALU"] * 4 / groups, m["SQ_ACTIVE_INST_LDS"] * 4 / groups, m["SQ_LDS_IDX_ACTIVE"] / groups, m["SQ_LDS_BANK_CONFLICT"] / groups, m["GRBM_GUI_ACTIVE"] / 8 / (us * 1e3)))
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            line.append("  HBM-side PMC per launch: FETCH_SIZE %.0f KiB raw (x2 correction: %.1f MB), WRITE_SIZE %.0f KiB (%.1f MB)" % (m["FETCH_SIZE"], m["FETCH_SIZE"] * 2048 / 1e6, m["WRITE_SIZE"], m["WRITE_SIZE"] * 1024 / 1e6))
            traffic[key] = {"FETCH_SIZE_KiB_raw": m["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": m["WRITE_SIZE"], "fetch_bytes_x2": m["FETCH_SIZE"] * 2048,
                            "write_bytes": m["WRITE_SIZE"] * 1024, "traffic_bytes_per_launch": m["FETCH_SIZE"] * 2048 + m["WRITE_SIZE"] * 1024, "avg_us": us}
        fo.write("\n".join(line) + "\n")
        print("\n".join(line))
import json
json.dump({"positions": 1 << 20, "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over tools/rules_bench.py (1 M positions per launch); FETCH x2 gfx950 correction for wide coalesced reads",
           "kernels": traffic}, open(out + "/pmc_rules_traffic.json", "w"), indent=1)
PY
rm -rf $OUT/a $OUT/b $OUT/f $OUT/w
