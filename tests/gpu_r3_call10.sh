#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python mug-diffusion_amd/build.py > gpurun_out/r3_build.log 2>&1
timeout 600 python -m pytest tests/test_train.py -x -q -m gpu -k "conv_layer_bf16" > gpurun_out/r3_tests_gpu_f.log 2>&1; echo "rc $?" >> gpurun_out/r3_tests_gpu_f.log
echo "== FAST staging, window 2 stages ahead" > gpurun_out/r3_tgemm_bench.txt
timeout 300 python tests/gpu_tgemm_bench.py >> gpurun_out/r3_tgemm_bench.txt 2>&1
timeout 300 python tests/gpu_train_probe.py --B 32 --reps 4 --adamw --bf16 > gpurun_out/r3_train_probe_bf16.log 2>&1
tail -2 gpurun_out/r3_tests_gpu_f.log; grep -v amdgpu gpurun_out/r3_tgemm_bench.txt; grep step gpurun_out/r3_train_probe_bf16.log
