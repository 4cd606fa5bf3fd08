#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 45 python -m pytest tests/test_ops.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2 ) > gpurun_out/r4_last_ops_gpu.log 2>&1
cat gpurun_out/r4_last_ops_gpu.log
