#!/bin/bash
# GPU box: k_trunk_mx_c128 (product) vs the 12-wave experiment k_trunk_mx12_c128 (tools/ab/lib_mx_mx12.so, CCHESS_MX_KERNEL=12)
O=gpurun_out/${1:-mxab6}; mkdir -p $O
L=$(realpath tools/ab/lib_mx_mx12.so)
( CCHESS_HIP_LIB=$L CCHESS_MX_KERNEL=12 timeout 200 python tools/mx_check.py --blocks 1,2,7 --wsets trained_like,glorot --n 37 ) 2>&1 | cut -c1-330 | tail -6 | tee $O/check.txt
T() { ( timeout 200 python tools/mx_check.py --blocks "" --time --engines mx --launches 40 ) 2>&1 | grep "per launch" | tail -1; }
for r in 1 2 3; do
  echo "v1  : $(T)" | tee -a $O/times.txt
  echo "mx12: $(CCHESS_HIP_LIB=$L CCHESS_MX_KERNEL=12 T)" | tee -a $O/times.txt
done
[ -f tools/ab/lib_mx_timing12.so ] && CCHESS_HIP_LIB=$(realpath tools/ab/lib_mx_timing12.so) CCHESS_MX_KERNEL=12 CCHESS_MX_TIMING_V1=1 timeout 200 python tools/mx_timing.py 2>&1 | tail -1 | tee $O/timing.txt
