#!/bin/bash
# SQ counters of the stand-alone rules kernels (tools/rules_bench.py): where k_movegen_mask's time goes
O=gpurun_out/r04s; mkdir -p $O; ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
(timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $ROOT/$O/a -o p -- python $ROOT/tools/rules_bench.py > $ROOT/$O/a.out 2>&1) < /dev/null
C2="SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU"
(timeout 200 rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d $ROOT/$O/b -o p -- python $ROOT/tools/rules_bench.py > $ROOT/$O/b.out 2>&1) < /dev/null
cd $ROOT
python3 - $O <<'PY'
import csv, sys, collections, glob
out = sys.argv[1]
for d in ("a", "b"):
    f = glob.glob("%s/%s/**/*counter_collection.csv" % (out, d), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "movegen" in k:
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
tail -12 $O/a.out
rm -rf $O/a $O/b
