#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python mug-diffusion_amd/build.py > gpurun_out/r3_build.log 2>&1
(echo "# round 3, final tree -- training-step parity on the GPU (pytest tests/test_train.py -m gpu -s, MI355X): per-block-type error tables"; timeout 900 python -m pytest tests/test_train.py -x -q -m gpu -s -p no:cacheprovider 2>&1 | grep -v "^$" ) > gpurun_out/r3_train_parity.txt
tail -3 gpurun_out/r3_train_parity.txt
timeout 300 python tests/gpu_train_probe.py --B 32 --reps 6 --bf16 --adamw 2>&1 | grep "step" | tail -2
timeout 300 python tests/gpu_train_probe.py --B 32 --reps 4 --adamw 2>&1 | grep "step" | tail -1
