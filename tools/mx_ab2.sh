#!/bin/bash
# Runs ON THE GPU BOX: launch times of k_trunk_mx_c128 (v1), k_trunk_mx2_c128 and the latter's timing ablations in tools/ab/ (same box).
O=gpurun_out/${1:-mxab2}; mkdir -p $O
( timeout 200 python tools/mx_check.py --blocks 2,7 --wsets glorot,trained_like --n 37 ) 2>&1 | cut -c1-330 | tail -5 > $O/check.txt; cat $O/check.txt
T() { ( timeout 200 python tools/mx_check.py --blocks "" --time --engines mx --launches 30 ) 2>&1 | grep "per launch" | tail -1; }
for r in 1 2; do
  echo "v1: $(CCHESS_MX_KERNEL=1 T)" | tee -a $O/times.txt
  echo "v2: $(T)" | tee -a $O/times.txt
  for l in tools/ab/lib_mx_*.so; do echo "$(basename $l): $(CCHESS_HIP_LIB=$(realpath $l) T)" | tee -a $O/times.txt; done
done
