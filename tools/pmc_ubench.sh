#!/bin/bash
# Runs ON THE GPU BOX: two PMC passes over the tower micro-benchmark for one kernel variant (8 | 4 | 1).
# usage: tools/pmc_ubench.sh <variant> [B] [blocks];  output: gpurun_out/pmcu_<variant>_{a,b}/ + a printed summary
set -u
V=${1:-8}; B=${2:-8192}; NB=${3:-7}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for pass in a b; do
  if [ $pass = a ]; then C="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES";
  else C="SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT"; fi
  O=$ROOT/gpurun_out/pmcu_${V}_$pass
  mkdir -p $O
  (timeout 120 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O -o p -- $ROOT/tools/ubench/tower_base $B $NB 3 $V > $O/log.txt 2>&1) < /dev/null
done
python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/pmcu_${V}_a $ROOT/gpurun_out/pmcu_${V}_b $B $NB $ROOT/gpurun_out/pmc_sq_${V}.json
