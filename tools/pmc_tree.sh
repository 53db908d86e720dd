#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the tree / rules kernels over a short bench run + the rules micro-benchmark.
# usage: tools/pmc_tree.sh <outdir>
OUT=$1; ROOT=$(pwd); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
(timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $ROOT/$OUT/sq_a -o p -- python $ROOT/bench.py --no-cpu-baseline --steps 24 --warmup 4 > $ROOT/$OUT/sq_a.out 2>&1) < /dev/null
C2="SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
(timeout 200 rocprofv3 --kernel-trace --pmc $C2 --output-format csv -d $ROOT/$OUT/sq_b -o p -- python $ROOT/bench.py --no-cpu-baseline --steps 24 --warmup 4 > $ROOT/$OUT/sq_b.out 2>&1) < /dev/null
cd $ROOT
python3 - $OUT <<'PY'
import csv, sys, collections, glob
out = sys.argv[1]
for d in ("sq_a", "sq_b"):
    f = glob.glob("%s/%s/**/*counter_collection.csv" % (out, d), recursive=True)
    if not f: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        for key in ("k_select", "k_expand_backup", "k_movegen", "k_value_fc", "k_advance"):
            if key in k:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, cs in acc.items():
        print(d, key, {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()})
PY
