#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
( time python bench.py --gpus 1 --steps 20 --warmup 3 ) > gpurun_out/r4_bench_driver_cmd2.log 2>&1; grep '^{' gpurun_out/r4_bench_driver_cmd2.log > gpurun_out/r4_bench_driver_cmd2.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4_bench_driver_cmd2.json'))
print('bench', d['value'], d['ms_per_step'], 'ddim', d['ddim_loop_ms'], 'clock', d.get('shader_clock_mhz_under_matrix_load'), 'roofline', d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
print('tp2', d.get('throughput_mode',{}).get('value'), 'tp4', d.get('throughput_mode_4_songs',{}).get('value'), d.get('throughput_mode_4_songs',{}).get('unet_sample_steps_per_s'))
print('train', (d.get('training_step') or {}).get('value'), (d.get('training_step_fp32') or {}).get('value'), 'reduced', (d.get('reduced_precision_mode') or {}).get('value'), 'cpu', (d.get('cpu_baseline') or {}).get('value'))
PY
MUGD_LIB_PATH=$GRAFT_REPO_ROOT/tests/var/fp32mfma/libmugd.so python bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-training-step --no-reduced-mode > gpurun_out/r4_bench_fp32mfma_same_box.log 2>&1; grep '^{' gpurun_out/r4_bench_fp32mfma_same_box.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('fp32mfma build same box:', d['value'], d['ms_per_step'], d['ddim_loop_ms'], d.get('throughput_mode',{}).get('value'), d.get('throughput_mode_4_songs',{}).get('value'), d['roofline']['achieved'], d['roofline']['avg_launch_us'])"
