#!/bin/bash
# round 4, GPU session I: the 19-block lines on the final kernels
O=gpurun_out/r04I; mkdir -p $O
B="timeout 600 python bench.py --no-cpu-baseline"
$B --blocks 19 --steps 300 --steady-steps 300 --strict-steps 60 > $O/bench_19blk_fp16.json 2> $O/bench_19blk.err
python tools/jline.py $O/bench_19blk_fp16.json | grep -i "sims/s\|trunk" | head -8
