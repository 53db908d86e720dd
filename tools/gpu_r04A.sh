#!/bin/bash
# round 4, GPU session z (HEAD of the round): the whole GPU suite, smoke(),
# the default bench line (with the CPU baseline) and the driver-shaped one
O=gpurun_out/r04A; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -4 $O/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log ); tail -3 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/jline.py $O/bench_default.json | tee $O/bench_default.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err; python tools/jline.py $O/bench_driver_shape.json | head -3
