#!/bin/bash
# experiment: one full batch vs two / four part batches on separate streams (tools/experiments/dual_stream_probe.py)
O=gpurun_out/r04p; mkdir -p $O
timeout 900 python tools/experiments/dual_stream_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/dual_stream.txt
