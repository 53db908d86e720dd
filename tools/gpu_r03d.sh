#!/bin/bash
# persistent-workgroup experiment on the trunk kernel: ubench (bf16, tower only) and the bench (fp16, whole step), interleaved
O=gpurun_out/r03d; mkdir -p $O
for r in 1 2 3; do
  for p in 0 1; do ./tools/ubench/tower_persist 8192 7 20 8 0 $p; done
done 2>&1 | tee $O/ubench_persist.log
B="python bench.py --no-cpu-baseline --steps 400 --steady-steps 0"
for r in 1 2; do
  $B > $O/bench_base_$r.json 2>/dev/null; python tools/jline.py $O/bench_base_$r.json | head -1
  CCHESS_TOWER_PERSIST=1 $B > $O/bench_persist_$r.json 2>/dev/null; python tools/jline.py $O/bench_persist_$r.json | head -1
done
CCHESS_TOWER_PERSIST=1 python -m pytest tests/test_net.py tests/test_bench_path.py -m gpu -x -q 2>&1 | tail -3
