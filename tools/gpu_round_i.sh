#!/bin/bash
TAG=${1:-r02i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
( timeout 900 python -m pytest tests/test_facade.py tests/test_hip_rules.py -m gpu -q -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log )
tail -3 $OUT/pytest_gpu.log | cut -c1-200
( timeout 300 python tools/rules_bench.py > $OUT/rules_bench.log 2>&1 ); grep -v amdgpu.ids $OUT/rules_bench.log
mkdir -p $OUT/train && cd $OUT/train && ( SECONDS=0; timeout 900 python $ROOT/main.py --mode train --games 2048 --train_playout 100 --batch_size 512 --res_block_nums 7 --processor gpu --max_batches 2 > train.log 2> train.err; echo "wall seconds: $SECONDS" >> train.log ); cd $ROOT
grep -E "batch i|kl:|wall seconds|Error|error" $OUT/train/train.log $OUT/train/train.err | cut -c1-250 | tail -12
rm -rf $OUT/train/gpu_models $OUT/train/models
cd /tmp; (timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/stats -o s -- python $ROOT/bench.py --no-cpu-baseline --steps 100 --warmup 8 > $ROOT/$OUT/bench_under_rocprof.json 2> $ROOT/$OUT/stats.err) < /dev/null; cd $ROOT
find $OUT -name '*_kernel_trace.csv' -size +20M -delete
python tools/jline.py $OUT/bench_under_rocprof.json
