#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the trunk kernel AS THE BENCH RUNS IT (fp16, persistent workgroups, real search data in
# steady state) — two separate PMC passes, kernel-trace only.  usage: tools/pmc_trunk_bench.sh <outdir> [bench args]
OUT=$1; shift; ROOT=$(pwd); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"
Bc="SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT"
(timeout 300 rocprofv3 --kernel-trace --pmc $A --output-format csv -d $ROOT/$OUT/a -o p -- python $ROOT/bench.py --no-cpu-baseline --steps 24 --warmup 4 --age-steps 200 --steady-steps 0 "$@" > $ROOT/$OUT/a.out 2>&1) < /dev/null
(timeout 300 rocprofv3 --kernel-trace --pmc $Bc --output-format csv -d $ROOT/$OUT/b -o p -- python $ROOT/bench.py --no-cpu-baseline --steps 24 --warmup 4 --age-steps 200 --steady-steps 0 "$@" > $ROOT/$OUT/b.out 2>&1) < /dev/null
cd $ROOT
DA=$(dirname $(find $OUT/a -name 'p_counter_collection.csv' | head -1)); DB=$(dirname $(find $OUT/b -name 'p_counter_collection.csv' | head -1))
python3 tools/pmc_summary.py $DA $DB 8192 7 $OUT/pmc_sq_bench.json "bench.py search loop, 8192 positions per launch, 7 blocks, steady-state search data; args: $*"
du -sh $OUT/a $OUT/b | tr '\n' ' '; rm -rf $OUT/a $OUT/b   # raw rocprofv3 output (tens of MB): only the summary travels back
