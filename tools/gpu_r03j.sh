#!/bin/bash
# round 3, final state of the kernels: full GPU suite, default bench, rocprof stats + HBM PMC, SQ PMC of the trunk in the bench
O=gpurun_out/r03j; mkdir -p $O
python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/bench_default.json 2> $O/bench_default.err; python tools/jline.py $O/bench_default.json | head -3
python bench.py --steps 20 --warmup 5 > $O/bench_driver_shape.json 2>/dev/null; python tools/jline.py $O/bench_driver_shape.json | head -2
bash tools/profile_round.sh r03j/prof > $O/profile.log 2>&1
find $O/prof -name '*kernel_trace.csv' -delete
bash tools/pmc_trunk_bench.sh $O/sq_fp16 2>&1 | head -3
bash tools/pmc_trunk_bench.sh $O/sq_bf16 --dtype bf16 2>&1 | head -3
du -sh gpurun_out
