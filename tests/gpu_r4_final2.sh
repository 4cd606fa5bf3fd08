#!/bin/bash
# round 4, final tree: the whole GPU suite, smoke, the driver-style bench, rocprofv3 kernel stats of the bench, counter traffic of conv_gemm, per-layer table
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python mug-diffusion_amd/build.py > gpurun_out/r4f2_build.log 2>&1 || { tail -30 gpurun_out/r4f2_build.log; exit 1; }
( time timeout 1000 python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > gpurun_out/r4f2_gputest.log 2>&1; tail -6 gpurun_out/r4f2_gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4f2_smoke.log 2>&1; tail -1 gpurun_out/r4f2_smoke.log
( time python bench.py --gpus 1 --steps 20 --warmup 3 ) > gpurun_out/r4f2_bench_driver_cmd.log 2>&1; grep '^{' gpurun_out/r4f2_bench_driver_cmd.log > gpurun_out/r4f2_bench_driver_cmd.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4f2_bench_driver_cmd.json'))
print('bench', d['value'], d['ms_per_step'], 'ddim', d['ddim_loop_ms'], 'clock', d.get('shader_clock_mhz_under_matrix_load'), 'roofline', d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
print('tp2', d.get('throughput_mode',{}).get('value'), 'tp4', d.get('throughput_mode_4_songs',{}).get('value'), d.get('throughput_mode_4_songs',{}).get('unet_sample_steps_per_s'))
print('train', (d.get('training_step') or {}).get('value'), (d.get('training_step_fp32') or {}).get('value'), 'reduced', (d.get('reduced_precision_mode') or {}).get('value'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
rm -rf /tmp/prof4
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4 -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-training-step --no-throughput-mode --no-reduced-mode) > gpurun_out/r4f2_bench_under_rocprof.log 2>&1
f=$(find /tmp/prof4 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r4f2_bench_kernel_stats.csv; head -8 gpurun_out/r4f2_bench_kernel_stats.csv | cut -c1-160
rm -rf /tmp/pmc4
(cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc4/fetch -- python $GRAFT_REPO_ROOT/tests/gpu_unet_once.py --n 2) > gpurun_out/r4f2_pmc_fetch.log 2>&1
(cd /tmp && timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc4/write -- python $GRAFT_REPO_ROOT/tests/gpu_unet_once.py --n 2) > gpurun_out/r4f2_pmc_write.log 2>&1
python tests/pmc_summary.py /tmp/pmc4 gpurun_out/r4f2_conv_traffic.json | tail -8
rm -f gpurun_out/r4f2_per_layer_z512_b4.csv
MUGD_PROFILE_CSV=gpurun_out/r4f2_per_layer_z512_b4.csv timeout 200 python tests/gpu_probe.py --B 4 > gpurun_out/r4f2_probe_b4.txt 2>&1; grep -E "ddim|vae decode|wave encode|log-mel" gpurun_out/r4f2_probe_b4.txt
