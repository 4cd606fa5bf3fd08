#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_train.py -x -q -m gpu -p no:cacheprovider -s -k "timed_geometry" ) > gpurun_out/r4_train_parity_z512.log 2>&1; tail -40 gpurun_out/r4_train_parity_z512.log
