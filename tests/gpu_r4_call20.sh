#!/bin/bash
# concurrency across queues: N contexts x batch 4 (threads) against one context at batch 4 N
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r4_streams_ab.log
: > $L
timeout 400 python tests/gpu_streams_ab.py --b 4 --graph 0 2>&1 | grep -v amdgpu.ids >> $L
timeout 400 python tests/gpu_streams_ab.py --b 4 --graph 2 --nmax 2 2>&1 | grep -v amdgpu.ids >> $L
cat $L
