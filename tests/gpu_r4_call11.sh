#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python tests/gpu_convbench.py --sweep > gpurun_out/r4_conv_tiling_sweep_h3.txt 2>&1
cat gpurun_out/r4_conv_tiling_sweep_h3.txt
