#!/bin/bash
TAG=${1:-r02l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
( timeout 900 python -m pytest tests/test_net.py tests/test_hip_search.py tests/test_bench_path.py -m gpu -q -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
tail -3 $OUT/pytest_gpu.log | cut -c1-200
bash tools/ab_lib.sh tools/ubench/libcchess_hip_old.so > $OUT/ab_lib.log 2>&1; cat $OUT/ab_lib.log
