#!/bin/bash
TAG=${1:-r02c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
( timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
tail -6 $OUT/pytest_gpu.log | cut -c1-200
bash tools/ubench_exp.sh base prio pad1 pad2 pad3 al6 al8 > $OUT/ubench_exp.log 2>&1; cat $OUT/ubench_exp.log
( timeout 300 python bench.py --steps 200 --warmup 16 --no-cpu-baseline > $OUT/bench_200.json 2> $OUT/bench_200.err )
( timeout 300 python tools/rules_bench.py > $OUT/rules_bench.log 2>&1 ); cat $OUT/rules_bench.log
cd /tmp; (timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/stats -o s -- python $OLDPWD/bench.py --no-cpu-baseline --steps 100 --warmup 8 > $OLDPWD/$OUT/bench_under_rocprof.json 2> $OLDPWD/$OUT/stats.err) < /dev/null; cd $OLDPWD
find $OUT -name '*_kernel_trace.csv' -size +20M -delete
head -6 $(find $OUT/stats -name "*kernel_stats.csv" | head -1) | cut -c1-100,180-330
for f in $OUT/bench_*.json; do echo "== $f"; python tools/jline.py $f 2>&1 | head -12; done
