#!/bin/bash
# round 4, GPU session j: zero-work elasticity (skip loop of its own), interleaved
O=gpurun_out/r04j; mkdir -p $O
for rep in 1 2 3; do for sk in 0 3 9; do ./tools/experiments/bin/tower_skip$sk 8192 7 20 | tail -1; done; done > $O/tower_skip.log 2>&1; cat $O/tower_skip.log
