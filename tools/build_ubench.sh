#!/bin/bash
# builds tools/ubench/conv_<tag> variants (cross-compiles here, runs on the GPU box)
set -e
cd "$(dirname "$0")"
mkdir -p ubench
build() { tag=$1; shift; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" -o ubench/conv_$tag conv_ubench.hip & }
build base
build noact -DCZ_ABL=1
build noepi -DCZ_ABL=2
build noactepi -DCZ_ABL=3
build nomfma -DCZ_ABL=4
build nolds -DCZ_ABL=8
build nowstream -DCZ_ABL=16
build mfmaonly -DCZ_ABL=27
build_t() { tag=$1; shift; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" -o ubench/tower_$tag tower_ubench.hip & }
build_t base
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o ubench/fc_base fc_ubench.hip &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench/mfma_peak mfma_peak.hip &
wait
ls -la ubench
