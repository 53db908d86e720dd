#!/bin/bash
# Runs ON THE GPU BOX: same-box A/B of tower micro-benchmark variants, interleaved, 3 rounds.  usage: tools/ubench_exp.sh tag1 tag2 ...
cd "$(dirname "$0")/ubench"
for r in 1 2 3; do
  for t in "$@"; do
    printf "%-8s round %d: " $t $r; ./tower_$t 8192 7 10 8 2>&1 | tail -1
  done
done
