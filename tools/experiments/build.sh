#!/bin/bash
# tools/experiments/build.sh — cross-compiles (here) the checker / timing tools of the trunk-kernel variants that were measured
# and not adopted (cz_trunk_experiments.h); the binaries go to tools/experiments/bin/ (git-ignored, travels with gpurun).
#   variants_check B blocks fp16(0|1) iters variant(1 = sk, 2 = d)   bit-equality against k_tower8_c128 + alternating timings
#   tower_ubench   B blocks iters variant(8 = product, 4 = 4w, 1 = pw) zero(0|1)
#   tower_trace{1,2} (tower_trace.sh run)                               shader-clock timeline of a layer
#   tower_skip{0,3,9} B blocks iters                                    zero-work elasticity: 0 / 11 % / 33 % of the MFMAs removed
#   tower_product B blocks iters                                        the library's trunk kernel in the same harness
set -e
cd "$(dirname "$0")"
python3 gen_experiments_asm.py
mkdir -p bin
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off"
$H -o bin/variants_check variants_check.hip &
$H -o bin/tower_ubench tower_ubench.hip &
for lv in 1 2; do $H -DCZ_T8_TRACE=$lv -o bin/tower_trace$lv tower_trace.hip & done
for sk in 0 3 9; do $H -DCZ_T8_SKIPTEST=$sk -o bin/tower_skip$sk tower_skip_ubench.hip & done
$H -DCZ_T8_PRODUCT -o bin/tower_product tower_skip_ubench.hip &   # the kernel exactly as the library builds it (A/B against older binaries)
wait
ls -la bin
