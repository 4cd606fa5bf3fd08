#!/bin/bash
# round 4, call 5: executor phases run twice (second pass = hot operands): what could prefetching buy?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python mug-diffusion_amd/build.py > gpurun_out/r4_build5.log 2>&1 || { tail -30 gpurun_out/r4_build5.log; exit 1; }
rm -f gpurun_out/r4_x8t*.csv
MUGD_XEXEC_TWICE=1 timeout 300 python tests/gpu_xexec_ab.py --B 8 --rounds 1 --reps 1 --profile gpurun_out/r4_x8t > gpurun_out/r4_xexec_ab5.txt 2>&1; grep -E "RESULT|PROFILE|Error|error" gpurun_out/r4_xexec_ab5.txt
