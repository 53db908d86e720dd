#!/bin/bash
# k_movegen_mask with persistent waves + prefetch of the next group's boards: parity, then the rules micro-benchmark
O=gpurun_out/r04t; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_rules.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -12 | tee $O/tests.log
for r in 1 2; do timeout 300 python tools/rules_bench.py 2>&1 | grep "K1\|K3\|Zob"; done | tee $O/rules_bench.log
