#!/bin/bash
# round 4, GPU session r: rocprofv3 evidence for the FINAL kernels of the round (rank-major + dy = -1 skip) — kernel stats + HBM PMC of the
# default bench and of --dtype strict, SQ counters of both trunk kernels as the bench runs them, the strict bench lines (7 and 19 blocks)
O=gpurun_out/r04r; mkdir -p $O
bash tools/profile_round.sh r04r/prof > $O/profile_round.log 2>&1
bash tools/profile_round.sh r04r/prof_strict --dtype strict > $O/profile_round_strict.log 2>&1
bash tools/pmc_trunk_bench.sh $O/sq_fp16 > $O/sq_fp16.log 2>&1; mv $O/sq_fp16/pmc_sq_bench.json $O/pmc_sq_bench_fp16.json
PMC_KERNEL=trunk_split PMC_WG_POSITIONS=2 PMC_SLABS_PER_LAYER=36 bash tools/pmc_trunk_bench.sh $O/sq_strict --dtype strict > $O/sq_strict.log 2>&1; mv $O/sq_strict/pmc_sq_bench.json $O/pmc_sq_bench_strict.json
tail -8 $O/sq_fp16.log $O/sq_strict.log | cut -c1-300
B="timeout 600 python bench.py --no-cpu-baseline"
$B --dtype strict > $O/bench_strict.json 2> $O/bench_strict.err
$B --dtype strict --blocks 19 --steps 200 --steady-steps 300 > $O/bench_strict_19blk.json 2> $O/bench_strict_19blk.err
$B --blocks 19 --steps 300 --steady-steps 300 > $O/bench_19blk_fp16.json 2> $O/bench_19blk.err
for f in $O/bench_*.json; do echo "== $f"; python tools/jline.py $f 2>&1 | head -24; done
ls -R $O | head -60
