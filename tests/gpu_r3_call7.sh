#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python mug-diffusion_amd/build.py > gpurun_out/r3_build.log 2>&1
timeout 600 python -m pytest tests/test_train.py -x -q -m gpu -k "transformer or shipped_size" > gpurun_out/r3_tests_gpu_e.log 2>&1; echo "rc $?" >> gpurun_out/r3_tests_gpu_e.log
timeout 300 python tests/gpu_train_probe.py --B 32 --reps 4 --adamw --bf16 > gpurun_out/r3_train_probe_bf16.log 2>&1
rm -rf /tmp/trp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trp -- python $GRAFT_REPO_ROOT/tests/gpu_train_probe.py --B 32 --reps 3 --bf16 --adamw) > gpurun_out/r3_train_probe_bf16_prof.log 2>&1
f=$(find /tmp/trp -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r3_train_step_b32_bf16_kernel_stats.csv
f=$(find /tmp/trp -name "*kernel_trace.csv" | head -1); python tests/pp_tgemm_trace.py "$f" > gpurun_out/r3_tgemm_by_shape.txt 2>&1
# per-layer tables of the final tree (U-Net, VAE, wave encoder) + the probe
rm -f gpurun_out/r3_per_layer_z512_b4.csv
MUGD_PROFILE_CSV=gpurun_out/r3_per_layer_z512_b4.csv timeout 300 python tests/gpu_probe.py --out gpurun_out/r3_probe_z512.json > gpurun_out/r3_probe_z512.txt 2>&1
timeout 200 python tests/gpu_probe.py --B 8 --quick > gpurun_out/r3_probe_z512_b8.txt 2>&1
timeout 300 python tests/gpu_job_transcript.py /tmp/mug_job_demo > gpurun_out/r3_job_transcript.txt 2>&1
tail -3 gpurun_out/r3_tests_gpu_e.log; grep step gpurun_out/r3_train_probe_bf16.log; grep "ddim\|unet forward" gpurun_out/r3_probe_z512.txt gpurun_out/r3_probe_z512_b8.txt; tail -12 gpurun_out/r3_job_transcript.txt
