#!/bin/bash
# (Record of a round-3 experiment: the grid-stride / persistent trunk kernel it compares lives in git history, commits 3f397c6 ..
# the commit that reverted it; on the current tree CCHESS_TOWER_PERSIST has no effect.)
# Runs ON THE GPU BOX: interleaved A/B/C of the trunk launch shapes through the bench (4 rounds):
#   r02   = the round-2 trunk kernel (first-layer weights prefetched into registers, one workgroup per batch)   [tools/ab/lib_r02_trunk.so]
#   grid  = this round's kernel (first-layer weights staged in LDS), one workgroup per batch                      [CCHESS_TOWER_PERSIST=0]
#   pers  = this round's kernel, one persistent workgroup per CU                                                  [CCHESS_TOWER_PERSIST=1]
B="python bench.py --no-cpu-baseline --steps 400 --warmup 16 --steady-steps 0"
for r in 1 2 3 4; do
  printf "r02  %d: " $r; CCHESS_HIP_LIB=$(realpath tools/ab/lib_r02_trunk.so) $B 2>/dev/null | python tools/jline.py | head -1
  printf "grid %d: " $r; CCHESS_TOWER_PERSIST=0 $B 2>/dev/null | python tools/jline.py | head -1
  printf "pers %d: " $r; CCHESS_TOWER_PERSIST=1 $B 2>/dev/null | python tools/jline.py | head -1
done
