#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/r4_tl_z512_b4.csv
timeout 300 python tests/gpu_timeline.py --z 512 --B 4 --out gpurun_out/r4_tl_z512_b4.csv > gpurun_out/r4_timeline_z512_b4.txt 2>&1
head -16 gpurun_out/r4_timeline_z512_b4.txt
