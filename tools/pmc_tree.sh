#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the tree / rules kernels over a short bench run + the rules micro-benchmark.
# usage: tools/pmc_tree.sh <outdir>
OUT=$1; ROOT=$(pwd); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
(timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $ROOT/$OUT/sq_a -o p -- python $ROOT/bench.py --no-cpu-baseline --steps 24 --warmup 4 --age-steps 200 --steady-steps 0 > $ROOT/$OUT/sq_a.out 2>&1) < /dev/null
C2="SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
(timeout 200 rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d $ROOT/$OUT/sq_b -o p -- python $ROOT/bench.py --no-cpu-baseline --steps 24 --warmup 4 --age-steps 200 --steady-steps 0 > $ROOT/$OUT/sq_b.out 2>&1) < /dev/null
cd $ROOT
python3 - $OUT <<'PY'
import csv, sys, collections, glob, json
out = sys.argv[1]
res = collections.defaultdict(dict)
for d in ("sq_a", "sq_b"):
    f = glob.glob("%s/%s/**/*counter_collection.csv" % (out, d), recursive=True)
    if not f:
        print(d, "no csv")
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        for key in ("k_select", "k_expand_backup", "k_movegen", "k_value_fc", "k_advance_lds", "k_pick_ready"):
            if key in k:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, cs in acc.items():
        for c, v in cs.items():
            res[key][c] = sum(v) / len(v)
summ = {}
for key, c in res.items():
    w = c.get("SQ_WAVES", 0) or 1
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    summ[key] = {"waves_per_launch": w,
                 "per_wave": {"cycles": 4 * wc / w, "valu": c.get("SQ_INSTS_VALU", 0) / w, "salu": c.get("SQ_INSTS_SALU", 0) / w,
                              "lds": c.get("SQ_INSTS_LDS", 0) / w, "vmem": c.get("SQ_INSTS_VMEM", 0) / w, "smem": c.get("SQ_INSTS_SMEM", 0) / w,
                              "branch": c.get("SQ_INSTS_BRANCH", 0) / w},
                 "share_of_wave_cycles": {"issuing (SQ_ACTIVE_INST_ANY)": c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                                          "issue-stalled (SQ_WAIT_INST_ANY)": c.get("SQ_WAIT_INST_ANY", 0) / wc,
                                          "parked in s_waitcnt / barrier (SQ_WAIT_ANY)": c.get("SQ_WAIT_ANY", 0) / wc},
                 "raw": c}
    print(key, json.dumps(summ[key]["per_wave"]), json.dumps(summ[key]["share_of_wave_cycles"]))
json.dump({"method": "rocprofv3 --kernel-trace --pmc, two separate passes over `python bench.py --no-cpu-baseline --steps 24 --warmup 4` "
                     "(tools/pmc_tree.sh); means over the launches of each kernel; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles",
           "kernels": summ}, open(out + "/pmc_tree_sq.json", "w"), indent=1)
PY
rm -rf $OUT/sq_a $OUT/sq_b   # raw rocprofv3 output (tens of MB): only the summary travels back
