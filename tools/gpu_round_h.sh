#!/bin/bash
TAG=${1:-r02h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
( timeout 900 python -m pytest tests/test_hip_rules.py tests/test_hip_search.py tests/test_bench_path.py tests/test_scale_properties.py tests/test_width.py tests/test_selfplay_golden.py tests/test_selfplay_device.py tests/test_facade.py -m gpu -q -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
tail -4 $OUT/pytest_gpu.log | cut -c1-200
( timeout 300 python tools/rules_bench.py > $OUT/rules_bench.log 2>&1 ); grep -v amdgpu.ids $OUT/rules_bench.log
cd /tmp; (timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/stats -o s -- python $OLDPWD/bench.py --no-cpu-baseline --steps 100 --warmup 8 > $OLDPWD/$OUT/bench_under_rocprof.json 2> $OLDPWD/$OUT/stats.err) < /dev/null; cd $OLDPWD
find $OUT -name '*_kernel_trace.csv' -size +20M -delete
head -7 $(find $OUT/stats -name "*kernel_stats.csv" | head -1) | cut -c1-60,200-330
bash tools/pmc_tree.sh $OUT/pmc_tree > $OUT/pmc_tree.log 2>&1; tail -5 $OUT/pmc_tree.log | cut -c1-330
