#!/bin/bash
# the strict trunk kernel with the rank-major cell order + dy = -1 skip: parity tests, then strict bench A/B against the previous
# commit's library (tools/bin/libcchess_hip_head.so, same box)
O=gpurun_out/r04o; mkdir -p $O
timeout 900 python -m pytest tests/test_net.py tests/test_bench_path.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -5 | tee $O/tests.log
for rep in 1 2; do
for lib in tools/bin/libcchess_hip_head.so cchess_zero_amd/libcchess_hip.so; do
  n=$(basename $lib .so)_$rep
  CCHESS_HIP_LIB=$(pwd)/$lib timeout 600 python bench.py --dtype strict --steps 120 --warmup 20 --steady-steps 240 --no-cpu-baseline > $O/strict_$n.json 2> $O/strict_$n.err
  echo "== $lib"; python tools/jline.py $O/strict_$n.json | grep -i "sims/s\|trunk" | head -4
done; done 2>&1 | tee $O/strict_ab.txt
