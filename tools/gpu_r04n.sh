#!/bin/bash
# same box, same harness (tools/experiments/tower_skip_ubench.hip, 8192 positions, 7 blocks, fp16): the trunk kernel of the
# previous commit (natural cell order) / rank-major order without the skip / rank-major order with the dy = -1 skip
O=gpurun_out/r04n; mkdir -p $O
for rep in 1 2 3; do for b in tower_head tower_perm_noskip tower_perm_skip; do echo -n "$b: "; ./tools/experiments/bin/$b 8192 7 20 | tail -1; done; done 2>&1 | tee $O/trunk_ab.txt
