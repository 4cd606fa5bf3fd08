#!/bin/bash
# the M-split form forced wherever it exists, 32-wide tiles forced: full-size network goldens on the GPU
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( MUGD_CONV_TN=32 MUGD_CONV_WIDE=1 timeout 300 python -m pytest tests/test_nets.py -x -q -m gpu -p no:cacheprovider -k "unet_forward or ddim_tiny or vae_decode" 2>&1 | tail -3 ) > gpurun_out/r4_wide_nets_gpu.log 2>&1
cat gpurun_out/r4_wide_nets_gpu.log
