#!/bin/bash
# the rank-major trunk kernel with the dy = -1 skip, SCC clobber declared: net + bench-path parity, then the default bench line
O=gpurun_out/r04m; mkdir -p $O
timeout 900 python -m pytest tests/test_net.py tests/test_bench_path.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 | tee $O/tests.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/jline.py $O/bench_default.json | tee $O/bench_default.txt
