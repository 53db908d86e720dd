#!/bin/bash
TAG=${1:-r02e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
( timeout 900 python -m pytest tests/test_hip_rules.py tests/test_hip_search.py tests/test_scale_properties.py tests/test_width.py -m gpu -q -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
tail -4 $OUT/pytest_gpu.log | cut -c1-200
( timeout 300 python tools/rules_bench.py > $OUT/rules_bench.log 2>&1 ); cat $OUT/rules_bench.log
( timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "rc=$?" >> $OUT/bench_default.err )
python tools/jline.py $OUT/bench_default.json
