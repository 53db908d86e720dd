#!/bin/bash
# Runs ON THE GPU BOX: same-box A/B of two builds of libcchess_hip.so through the bench (interleaved, 3 rounds).
# usage: tools/ab_lib.sh <old.so> [bench args...]
OLD=$(realpath $1); shift
for r in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then export CCHESS_HIP_LIB=$OLD; else unset CCHESS_HIP_LIB; fi
    printf "%s round %d: " $which $r
    python bench.py --no-cpu-baseline --steps 400 --warmup 16 "$@" 2>/dev/null | python tools/jline.py | head -1
  done
done
