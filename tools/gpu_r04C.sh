#!/bin/bash
# round 4, GPU session C (HEAD with the class-tiled trunk kernel): the whole GPU suite, smoke(), the default bench line, then the
# rocprofv3 evidence of the same kernels (kernel stats + HBM PMC of the default bench, SQ counters of the fp16 trunk)
O=gpurun_out/r04C; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log ); tail -4 $O/pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log ); tail -3 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/jline.py $O/bench_default.json | tee $O/bench_default.txt
bash tools/profile_round.sh r04C/prof > $O/profile_round.log 2>&1
bash tools/pmc_trunk_bench.sh $O/sq_fp16 > $O/sq_fp16.log 2>&1; mv $O/sq_fp16/pmc_sq_bench.json $O/pmc_sq_bench_fp16.json; tail -4 $O/sq_fp16.log | cut -c1-300
