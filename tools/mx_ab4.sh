#!/bin/bash
# GPU box: the working tree's k_trunk_mx_c128 against tools/ab/lib_mx_old.so (HEAD's), interleaved; then the timing split
O=gpurun_out/${1:-mxab}; mkdir -p $O
( timeout 200 python tools/mx_check.py --blocks 2,7 --wsets trained_like,glorot --n 37 ) 2>&1 | cut -c1-330 | tail -4 | tee $O/check.txt
T() { ( timeout 200 python tools/mx_check.py --blocks "" --time --engines mx --launches 40 ) 2>&1 | grep "per launch" | tail -1; }
for r in 1 2 3; do
  echo "old: $(CCHESS_HIP_LIB=$(realpath tools/ab/lib_mx_old.so) T)" | tee -a $O/times.txt
  echo "new: $(T)" | tee -a $O/times.txt
done
[ -f tools/ab/lib_mx_timing1.so ] && CCHESS_HIP_LIB=$(realpath tools/ab/lib_mx_timing1.so) timeout 200 python tools/mx_timing.py 2>&1 | tail -1 | tee $O/timing.txt
