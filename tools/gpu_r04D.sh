#!/bin/bash
# class-tiled trunk kernel with the lanes relabelled to the ds_read_b128 lane groups: parity, same-box A/B (rank-major /
# class-tiled in lane order / class-tiled relabelled), SQ counters of the bench's trunk launches
O=gpurun_out/r04D; mkdir -p $O
timeout 900 python -m pytest tests/test_net.py tests/test_bench_path.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3 | tee $O/tests.log
for rep in 1 2 3; do for b in tower_perm_skip tower_ct tower_ct2; do echo -n "$b: "; ./tools/experiments/bin/$b 8192 7 20 | tail -1; done; done 2>&1 | tee $O/trunk_ab.txt
bash tools/pmc_trunk_bench.sh $O/sq_fp16 > $O/sq_fp16.log 2>&1; mv $O/sq_fp16/pmc_sq_bench.json $O/pmc_sq_bench_fp16.json; tail -5 $O/sq_fp16.log | head -3 | cut -c1-300
