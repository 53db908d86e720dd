#!/bin/bash
# round 4, GPU session b: the whole GPU suite (new: strict engine, N>1 path), smoke, the new bench line (telemetry, clock probe,
# MFMA ceiling probe, strict leg), the driver-shaped run
O=gpurun_out/r04b; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log )
tail -25 $O/pytest_gpu.log | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
B="timeout 600 python bench.py"
$B > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err
$B --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err
for f in $O/bench_*.json; do echo "== $f"; python tools/jline.py $f 2>&1 | head -30; done
