#!/bin/bash
# the class-tiled trunk kernel (top / left / right / bottom tiles skipped, explicit swizzle keys): parity, the same-box A/B
# against the rank-major kernel of the previous commits (tools/experiments/bin/tower_perm_skip), the default bench line
O=gpurun_out/r04B; mkdir -p $O
timeout 900 python -m pytest tests/test_net.py tests/test_bench_path.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4 | tee $O/tests.log
for rep in 1 2 3; do for b in tower_perm_skip tower_ct; do echo -n "$b: "; ./tools/experiments/bin/$b 8192 7 20 | tail -1; done; done 2>&1 | tee $O/trunk_ab.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; python tools/jline.py $O/bench_default.json | tee $O/bench_default.txt | head -3
