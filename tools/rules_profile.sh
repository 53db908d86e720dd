#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 --kernel-trace --stats over tools/rules_bench.py (the stand-alone rules kernels on 1 M positions):
# a kernel_stats.csv that lists k_movegen_list, k_movegen_mask, k_hash, k_apply_move, k_encode_planes.  usage: tools/rules_profile.sh <outdir>
OUT=$(pwd)/$1; ROOT=$(pwd); mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python $ROOT/tools/rules_bench.py > $OUT/rules_bench_under_rocprof.log 2>&1) < /dev/null
find $OUT -name '*_kernel_trace.csv' -size +20M -delete
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/rules_kernel_stats.csv 2>/dev/null
head -12 $OUT/rules_kernel_stats.csv | cut -c1-160
