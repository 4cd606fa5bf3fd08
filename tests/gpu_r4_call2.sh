#!/bin/bash
# round 4, call 2: first run of the XCD-resident executor on the GPU: parity tests, then the A/B at batch 8 / 16
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python mug-diffusion_amd/build.py > gpurun_out/r4_build2.log 2>&1 || { tail -30 gpurun_out/r4_build2.log; exit 1; }
timeout 600 python -m pytest tests/test_nets.py -x -q -m gpu -p no:cacheprovider -k "executor" > gpurun_out/r4_t2.log 2>&1; tail -25 gpurun_out/r4_t2.log
timeout 300 python tests/gpu_xexec_ab.py --B 8 16 > gpurun_out/r4_xexec_ab.txt 2>&1; grep -E "RESULT|Error|error" gpurun_out/r4_xexec_ab.txt
timeout 200 python -m pytest tests/test_nets.py tests/test_ops.py -x -q -m gpu -p no:cacheprovider -k "unet_forward or attention or s4_conv or conv1d" > gpurun_out/r4_t2b.log 2>&1; tail -3 gpurun_out/r4_t2b.log
