#!/bin/bash
# phase timeline with the "first chunk parked" phase split into issue / arrive / park, batch 4 and 16
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python tests/gpu_timeline.py --z 512 --B 4 --out gpurun_out/r4_tl2_b4.csv 2>&1 | grep -v amdgpu.ids | head -80
timeout 300 python tests/gpu_timeline.py --z 512 --B 16 --out gpurun_out/r4_tl2_b16.csv 2>&1 | grep -v amdgpu.ids | head -40
