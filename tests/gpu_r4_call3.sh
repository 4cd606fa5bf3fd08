#!/bin/bash
# round 4, call 3: executor with inlined bodies (no callee-saved traffic): parity + A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python mug-diffusion_amd/build.py > gpurun_out/r4_build3.log 2>&1 || { tail -30 gpurun_out/r4_build3.log; exit 1; }
timeout 600 python -m pytest tests/test_nets.py -x -q -m gpu -p no:cacheprovider -k "executor" > gpurun_out/r4_t3.log 2>&1; tail -5 gpurun_out/r4_t3.log
timeout 300 python tests/gpu_xexec_ab.py --B 8 16 > gpurun_out/r4_xexec_ab3.txt 2>&1; grep -E "RESULT|Error|error" gpurun_out/r4_xexec_ab3.txt
