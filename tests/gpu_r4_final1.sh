#!/bin/bash
# round 4: the whole GPU suite, the driver-style bench, rocprofv3 kernel stats of the bench, counter traffic of conv_gemm, per-layer tables
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python mug-diffusion_amd/build.py > gpurun_out/r4_buildf.log 2>&1 || { tail -30 gpurun_out/r4_buildf.log; exit 1; }
( time timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > gpurun_out/r4_gputest.log 2>&1; tail -6 gpurun_out/r4_gputest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4_smoke.log 2>&1; tail -1 gpurun_out/r4_smoke.log
( time python bench.py --gpus 1 --steps 20 --warmup 3 ) > gpurun_out/r4_bench_driver_cmd.log 2>&1; grep '^{' gpurun_out/r4_bench_driver_cmd.log > gpurun_out/r4_bench_driver_cmd.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4_bench_driver_cmd.json'))
print('bench', d['value'], d['ms_per_step'], 'ddim', d['ddim_loop_ms'], 'roofline', d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
print('tp2', d.get('throughput_mode',{}).get('value'), 'tp4', d.get('throughput_mode_4_songs',{}).get('value'), d.get('throughput_mode_4_songs',{}).get('unet_sample_steps_per_s'))
print('train', (d.get('training_step') or {}).get('value'), (d.get('training_step_fp32') or {}).get('value'), 'reduced', (d.get('reduced_precision_mode') or {}).get('value'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
rm -rf /tmp/prof4
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof4 -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-training-step --no-throughput-mode --no-reduced-mode) > gpurun_out/r4_bench_under_rocprof.log 2>&1
f=$(find /tmp/prof4 -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r4_bench_kernel_stats.csv; head -8 gpurun_out/r4_bench_kernel_stats.csv | cut -c1-160
grep '^{' gpurun_out/r4_bench_under_rocprof.log > gpurun_out/r4_bench_under_rocprof.json
rm -rf /tmp/pmc4
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc4/fetch -- python $GRAFT_REPO_ROOT/tests/gpu_unet_once.py --n 2) > gpurun_out/r4_pmc_fetch.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc4/write -- python $GRAFT_REPO_ROOT/tests/gpu_unet_once.py --n 2) > gpurun_out/r4_pmc_write.log 2>&1
python tests/pmc_summary.py /tmp/pmc4 gpurun_out/r4_conv_traffic.json | tail -8
rm -rf /tmp/pmc4b
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d /tmp/pmc4b -- python $GRAFT_REPO_ROOT/tests/gpu_unet_once.py --n 2) > gpurun_out/r4_pmc_mfma.log 2>&1
python tests/pmc_mfma_summary.py /tmp/pmc4b gpurun_out/r4_pmc_unet_mfma.txt | head -12
rm -f gpurun_out/r4_per_layer_z512_b4.csv
MUGD_PROFILE_CSV=gpurun_out/r4_per_layer_z512_b4.csv timeout 200 python tests/gpu_probe.py --B 4 > gpurun_out/r4_probe_b4.txt 2>&1; grep -E "ddim|vae decode|wave encode|log-mel" gpurun_out/r4_probe_b4.txt
