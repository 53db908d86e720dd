#!/bin/bash
# debug: the rank-major trunk kernel with the skip disabled (tools/bin/libcchess_hip_noskip.so) vs the skip build
O=gpurun_out/r04l; mkdir -p $O
for lib in tools/bin/libcchess_hip_noskip.so cchess_zero_amd/libcchess_hip.so; do
  echo "== $lib"
  CCHESS_HIP_LIB=$(pwd)/$lib timeout 600 python -m pytest tests/test_net.py -m gpu -q -s -p no:cacheprovider -k "fused_net_vs_fp32_restatement and fp16 and 2-glorot or fused_net_kernel_paths_agree" 2>&1 | grep "fp16 \|passed\|failed" | cut -c1-200
done
