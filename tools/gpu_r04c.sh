#!/bin/bash
# round 4, GPU session c: the whole GPU suite without -x (new: mask-only K1 kernel, N>1 training test with its logs kept), the
# stand-alone rules bench
O=gpurun_out/r04c; mkdir -p $O
( timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log )
tail -40 $O/pytest_gpu.log | cut -c1-300
( timeout 300 python tools/rules_bench.py > $O/rules_bench.log 2>&1 ); cat $O/rules_bench.log
ls gpurun_out/testlogs 2>/dev/null && tail -60 gpurun_out/testlogs/*.txt | cut -c1-400
