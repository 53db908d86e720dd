#!/bin/bash
# Builds timing-ablation variants of the library (here, no GPU): tools/ab/lib_mx_<name>.so with MX_ABLATE=<flags> in the generated
# slab bodies of k_trunk_mx_c128.  Results of the variants are WRONG on purpose; only their launch time means anything.
#   usage: tools/experiments/mx_ablate.sh name=flag,flag ...      e.g.  nobarrier=nobarrier noc=noc dmaBC=place2:BC noxch=define:MX2_ABLATE_NO_EXCHANGE
#   (the experiments' slab bodies cz_trunk_mx2_asm.inc / cz_trunk_mx12_asm.inc are generated: tools/gen_tower_asm.py writes them next to
#   the sources it is given — here the temporary copy — and they are not kept in git)
#   mx2 / mx2:<DEFINE>: a library that also holds the round-6 experiment k_trunk_mx2_c128 (tools/experiments/cz_trunk_mx2.h; run with
#   CCHESS_MX_KERNEL=2), e.g.  mx2=mx2  timing=mx2:MX2_TIMING  noxch=mx2:MX2_ABLATE_NO_EXCHANGE
# On the GPU box:  for l in tools/ab/lib_mx_*.so; do CCHESS_HIP_LIB=$(realpath $l) python tools/mx_check.py --blocks "" --time --engines mx; done
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/ab
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  top=$(mktemp -d /tmp/mxab.XXXX); d=$top/pkg/csrc; mkdir -p $d $top/include
  cp -r cchess_zero_amd/csrc/. $d/; cp include/cchess_hip.h $top/include/; cp tools/experiments/cz_trunk_mx2.h tools/experiments/cz_trunk_mx12.h $d/
  DEF=""
  case $flags in place:*) MX_DMA_PLACE=${flags#place:} python3 tools/gen_tower_asm.py $d > /dev/null;;
                 place2:*) MX2_DMA_PLACE=${flags#place2:} python3 tools/gen_tower_asm.py $d > /dev/null;;
                 mx2) DEF="-DCZ_EXPERIMENT_MX2"; python3 tools/gen_tower_asm.py $d > /dev/null;;
                 mx12) DEF="-DCZ_EXPERIMENT_MX12"; python3 tools/gen_tower_asm.py $d > /dev/null;;
                 mx12:*) DEF="-DCZ_EXPERIMENT_MX12 -D${flags#mx12:}"; python3 tools/gen_tower_asm.py $d > /dev/null;;
                 mx2:*) DEF="-DCZ_EXPERIMENT_MX2 -D${flags#mx2:}"; python3 tools/gen_tower_asm.py $d > /dev/null;;
                 define:*) DEF="-D${flags#define:}"; python3 tools/gen_tower_asm.py $d > /dev/null;;
                 *) MX_ABLATE=$flags python3 tools/gen_tower_asm.py $d > /dev/null;; esac
  ( cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt $DEF \
      -Wno-unused-function -o $OLDPWD/tools/ab/lib_mx_$name.so cz_api.hip cz_tables.hip cz_rules.hip cz_search.hip cz_selfplay.hip cz_conv.hip cz_heads.hip cz_probe.hip ) &
done
wait
ls -la tools/ab/
