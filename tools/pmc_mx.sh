#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of one trunk engine over tools/mx_check.py --time (8192 corpus positions, 7 blocks), two PMC
# passes, kernel-trace only.  usage: tools/pmc_mx.sh <outdir> <engine: fp16|x3|mx>
OUT=$1; ENG=$2; ROOT=$(pwd); mkdir -p $OUT
case $ENG in mx) export PMC_KERNEL=trunk_mx PMC_WG_POSITIONS=2 PMC_SLABS_PER_LAYER=36;; x3) export PMC_KERNEL=trunk_split PMC_WG_POSITIONS=2 PMC_SLABS_PER_LAYER=36;; *) export PMC_KERNEL=tower8;; esac
cd /tmp && export TMPDIR=/tmp
A="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
Bc="SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT"
(timeout 200 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $ROOT/$OUT/a -o p -- python $ROOT/tools/mx_check.py --blocks "" --time --engines $ENG --launches 12 > $ROOT/$OUT/a.out 2>&1) < /dev/null
(timeout 200 rocprofv3 --kernel-trace --pmc $Bc --output-format csv -d $ROOT/$OUT/b -o p -- python $ROOT/tools/mx_check.py --blocks "" --time --engines $ENG --launches 12 > $ROOT/$OUT/b.out 2>&1) < /dev/null
cd $ROOT
DA=$(dirname $(find $OUT/a -name 'p_counter_collection.csv' | head -1)); DB=$(dirname $(find $OUT/b -name 'p_counter_collection.csv' | head -1))
python3 tools/pmc_summary.py $DA $DB 8192 7 $OUT/pmc_sq_$ENG.json "tools/mx_check.py --time, 8192 corpus positions per launch, 7 blocks, engine $ENG"
rm -rf $OUT/a $OUT/b
