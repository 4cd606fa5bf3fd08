#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python mug-diffusion_amd/build.py > gpurun_out/r3_build.log 2>&1
timeout 100 python -m pytest tests/test_train.py tests/test_ops.py -x -q -m gpu -p no:cacheprovider -k "s4 or shipped_training_step_vs" > gpurun_out/r3_t16a.log 2>&1; tail -1 gpurun_out/r3_t16a.log
timeout 70 python -m pytest tests/test_nets.py -x -q -m gpu -p no:cacheprovider -k "unet_forward or ddim_full" > gpurun_out/r3_t16b.log 2>&1; tail -1 gpurun_out/r3_t16b.log
timeout 60 python tests/gpu_train_probe.py --B 32 --reps 5 --bf16 --adamw 2>&1 | grep "step 4"
