#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python mug-diffusion_amd/build.py > gpurun_out/r3_build.log 2>&1
timeout 900 python -m pytest tests/test_train.py -x -q -m gpu -p no:cacheprovider -k "bf16 or bracket or resblock or shipped or whole" > gpurun_out/r3_train_tests_gpu_g.log 2>&1
tail -2 gpurun_out/r3_train_tests_gpu_g.log
for v in 1 0 1 0; do
  MUGD_TRAIN_ACT_FP32=$v timeout 300 python tests/gpu_train_probe.py --B 32 --reps 6 --bf16 --adamw 2>&1 | grep "step 5" | sed "s/^/act_fp32=$v /"
done
