#!/bin/bash
# Runs ON THE GPU BOX: k_trunk_mx_c128 (CCHESS_MX_KERNEL=1) against k_trunk_mx2_c128 (default) — parity with the CPU emulation,
# then launch times, interleaved, same box.   usage: tools/mx_ab.sh <tag> [rounds]
O=gpurun_out/${1:-mxab}; mkdir -p $O
R=${2:-3}
( timeout 300 python tools/mx_check.py --blocks 1,2,7 --n 37 ) > $O/mx2_check.txt 2>&1; echo "check rc=$?" >> $O/mx2_check.txt
cut -c1-330 $O/mx2_check.txt | tail -12
for r in $(seq $R); do
  for k in 1 2; do
    echo "kernel $k" >> $O/mx_ab_times.txt
    ( CCHESS_MX_KERNEL=$k timeout 200 python tools/mx_check.py --blocks "" --time --engines mx --launches 30 ) 2>&1 | grep "per launch" >> $O/mx_ab_times.txt
  done
done
cat $O/mx_ab_times.txt
