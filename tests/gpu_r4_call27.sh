#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tests/gpu_icache_probe.hip -o /tmp/icp 2>/dev/null && timeout 60 /tmp/icp > gpurun_out/r4_icache_probe.txt 2>&1
cat gpurun_out/r4_icache_probe.txt
