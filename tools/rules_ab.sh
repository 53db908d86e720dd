#!/bin/bash
# Runs ON THE GPU BOX: tools/rules_bench.py (1 M positions) with the working tree's library against every tools/ab/lib_rules_*.so,
# interleaved, 3 rounds.  usage: tools/rules_ab.sh <outfile>
OUT=$1; : > $OUT
for r in 1 2 3; do
  for l in tree tools/ab/lib_rules_*.so; do
    if [ $l = tree ]; then unset CCHESS_HIP_LIB; else export CCHESS_HIP_LIB=$(realpath $l); fi
    echo "== round $r: $l" | tee -a $OUT
    timeout 300 python tools/rules_bench.py 2>&1 | grep -E "K1 movegen|Zobrist|K3 planes|K2 apply" | tee -a $OUT
  done
done
