#!/bin/bash
# builds (here) or runs (on the GPU box) tools/tower_trace.hip at both trace levels:  tools/tower_trace.sh build | run
set -e
cd "$(dirname "$0")"
mkdir -p ubench
if [ "$1" = build ]; then
  for lv in 1 2; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCZ_T8_TRACE=$lv -o bin/tower_trace$lv tower_trace.hip & done
  wait; ls -la bin/tower_trace*
else
  for dt in 1 0; do ./bin/tower_trace1 8192 7 $dt; done
  ./bin/tower_trace2 8192 7 1
  ./bin/tower_trace1 8192 7 1 300 1     # the skewed variant
  ./ubench/tower_base 8192 7 20 8
fi
