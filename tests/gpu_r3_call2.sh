#!/bin/bash
# round-3 GPU call 2: training tests on the GPU (fp32 + bf16), step time fp32 vs bf16 at batch 32, kernel stats of the bf16 step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_train.py -x -q -m gpu -s > gpurun_out/r3_train_tests_gpu.log 2>&1; echo "rc $?" >> gpurun_out/r3_train_tests_gpu.log
timeout 300 python tests/gpu_train_probe.py --B 32 --reps 3 > gpurun_out/r3_train_probe_fp32.log 2>&1
timeout 300 python tests/gpu_train_probe.py --B 32 --reps 3 --bf16 > gpurun_out/r3_train_probe_bf16.log 2>&1
rm -rf /tmp/trp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trp -- python $GRAFT_REPO_ROOT/tests/gpu_train_probe.py --B 32 --reps 2 --bf16) > gpurun_out/r3_train_probe_bf16_prof.log 2>&1
f=$(find /tmp/trp -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r3_train_step_b32_bf16_kernel_stats.csv
tail -4 gpurun_out/r3_train_tests_gpu.log; grep step gpurun_out/r3_train_probe_fp32.log gpurun_out/r3_train_probe_bf16.log; head -25 gpurun_out/r3_train_step_b32_bf16_kernel_stats.csv | cut -c1-150
