#!/bin/bash
# round 4, GPU session k: the trunk kernel with the rank-major cell order and the dy = -1 skip — net parity (numbers must equal
# r04a's to the digit: adding 0 * w is exact), bench default, ubench vs the skip-0 experiment binary (same box)
O=gpurun_out/r04k; mkdir -p $O
( timeout 1200 python -m pytest tests/test_net.py -m gpu -q -s -p no:cacheprovider > $O/test_net.log 2>&1; echo "pytest rc=$?" >> $O/test_net.log )
grep -n "fp16 \|bf16 \|passed\|failed\|rc=" $O/test_net.log | cut -c1-200 | head -40
for rep in 1 2; do ./tools/experiments/bin/tower_skip0 8192 7 20 | tail -1; done > $O/tower_now.log 2>&1; cat $O/tower_now.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; python tools/jline.py $O/bench_default.json | head -14
