#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python mug-diffusion_amd/build.py > gpurun_out/r3_build.log 2>&1
timeout 900 python -m pytest tests/test_train.py -x -q -m gpu -p no:cacheprovider -k "s4 or shipped or whole or bracket" > gpurun_out/r3_train_tests_gpu_g.log 2>&1
tail -2 gpurun_out/r3_train_tests_gpu_g.log
rm -rf /tmp/trp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trp -- python $GRAFT_REPO_ROOT/tests/gpu_train_probe.py --B 32 --reps 4 --bf16 --adamw) > gpurun_out/r3_train_probe_bf16_prof.log 2>&1
f=$(find /tmp/trp -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r3_train_step_b32_bf16_kernel_stats.csv
grep step gpurun_out/r3_train_probe_bf16_prof.log | tail -1
