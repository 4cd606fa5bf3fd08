#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( timeout 60 python -m pytest tests/test_ops.py -x -q -m gpu -p no:cacheprovider -k "host_rule" 2>&1 | grep -v "^$" | tail -12 ) > gpurun_out/r4_rule_test_gpu.log 2>&1
cat gpurun_out/r4_rule_test_gpu.log
