#!/bin/bash
# GPU box: v1 vs v2 launch times + the cycle split of the timing build (tools/ab/lib_mx_timing.so)
O=gpurun_out/${1:-mxab3}; mkdir -p $O
( timeout 200 python tools/mx_check.py --blocks 2,7 --wsets trained_like --n 37 ) 2>&1 | cut -c1-330 | tail -2 | tee $O/check.txt
T() { ( timeout 200 python tools/mx_check.py --blocks "" --time --engines mx --launches 30 ) 2>&1 | grep "per launch" | tail -1; }
for r in 1 2; do
  echo "v1: $(CCHESS_MX_KERNEL=1 T)" | tee -a $O/times.txt
  echo "v2: $(T)" | tee -a $O/times.txt
done
for l in tools/ab/lib_mx_*.so; do case $l in *timing*) ;; *) echo "$(basename $l): $(CCHESS_HIP_LIB=$(realpath $l) T)" | tee -a $O/times.txt;; esac; done
[ -f tools/ab/lib_mx_timing.so ] && CCHESS_HIP_LIB=$(realpath tools/ab/lib_mx_timing.so) timeout 200 python tools/mx_timing.py 2>&1 | tail -2 | tee $O/timing.txt
