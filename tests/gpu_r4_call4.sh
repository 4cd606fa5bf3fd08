#!/bin/bash
# round 4, call 4: executor per-phase timeline against the per-op event timings (batch 8)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python mug-diffusion_amd/build.py > gpurun_out/r4_build4.log 2>&1 || { tail -30 gpurun_out/r4_build4.log; exit 1; }
rm -f gpurun_out/r4_x8*.csv
timeout 300 python tests/gpu_xexec_ab.py --B 8 --rounds 1 --profile gpurun_out/r4_x8 > gpurun_out/r4_xexec_ab4.txt 2>&1; grep -E "RESULT|PROFILE|Error|error" gpurun_out/r4_xexec_ab4.txt
ls -la gpurun_out/r4_x8*
