#!/bin/bash
# VERDICT r4 item 6, measured: what removing the padding rows of k_tower8_c128's tile 0 can buy AT MOST.  Tile 0 = 24 padding rows +
# the 8 rank-0 cells of files 8, 9; a 16-row MFMA for it (368 instead of 384 GEMM rows) would halve its MFMA time in taps 3..8.
# This builds tools/ab/lib_t8_tile0skip.so in which cell group 0 skips tile 0's MFMAs in ALL nine taps (wrong results for 8 cells
# per 4 positions; timing only): TWICE the saving a 16-row tile could reach.  A/B on the GPU box:
#   for l in "" tools/ab/lib_t8_tile0skip.so; do CCHESS_HIP_LIB=${l:+$(realpath $l)} python tools/mx_check.py --blocks "" --time --engines fp16; done
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/ab
top=$(mktemp -d /tmp/t8ab.XXXX); d=$top/pkg/csrc; mkdir -p $d $top/include
cp -r cchess_zero_amd/csrc/. $d/; cp include/cchess_hip.h $top/include/
sed -i 's/wr < 2 ? 0x15 : (wr == 2 ? 0x21861 : 0x15000)/wr == 0 ? 0x15555 : (wr == 1 ? 0x15 : (wr == 2 ? 0x21861 : 0x15000))/' $d/cz_conv_kernel.h
grep -q "wr == 0 ? 0x15555" $d/cz_conv_kernel.h
( cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -Wno-unused-function -o $OLDPWD/tools/ab/lib_t8_tile0skip.so cz_api.hip cz_tables.hip cz_rules.hip cz_search.hip cz_selfplay.hip cz_conv.hip cz_heads.hip cz_probe.hip )
ls -la tools/ab/lib_t8_tile0skip.so
