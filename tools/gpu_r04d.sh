#!/bin/bash
# round 4, GPU session d: tree / self-play / multirank tests after the node layout change (+2 B per node: src|dst), default bench
O=gpurun_out/r04d; mkdir -p $O
( timeout 1800 python -m pytest tests/test_hip_search.py tests/test_width.py tests/test_selfplay_device.py tests/test_selfplay_golden.py tests/test_bench_path.py tests/test_multirank_gpu.py tests/test_facade.py tests/test_play.py -m gpu -q -s -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log )
grep -n "agreement\|passed\|failed\|rc=" $O/pytest_gpu.log | cut -c1-250 | tail -15
timeout 600 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; python tools/jline.py $O/bench_default.json | head -30
ls gpurun_out/testlogs 2>/dev/null && tail -30 gpurun_out/testlogs/*.txt | cut -c1-300
