#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python mug-diffusion_amd/build.py > gpurun_out/r3_build.log 2>&1
timeout 900 python -m pytest tests/test_train.py tests/test_nets.py -x -q -m gpu -k "bf16 or reduced or graph_and_eager" > gpurun_out/r3_tests_gpu_d.log 2>&1; echo "rc $?" >> gpurun_out/r3_tests_gpu_d.log
# headline A/B within one box: eager (default) vs per-step graph vs eager again
for m in 0 1 0 1; do MUGD_GRAPH=$m timeout 300 python bench.py --steps 5 --warmup 2 --no-training-step --no-cpu-baseline --no-throughput-mode --no-reduced-mode --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('MUGD_GRAPH=$m', round(d['value'],2), 'charts/s', round(d['ms_per_step'],2), 'ms/step  ddim', round(d['ddim_loop_ms'],2))" >> gpurun_out/r3_bench_graph_ab.txt; done
timeout 300 python tests/gpu_train_probe.py --B 32 --reps 4 --adamw --bf16 > gpurun_out/r3_train_probe_bf16.log 2>&1
rm -rf /tmp/trp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trp -- python $GRAFT_REPO_ROOT/tests/gpu_train_probe.py --B 32 --reps 3 --bf16 --adamw) > gpurun_out/r3_train_probe_bf16_prof.log 2>&1
f=$(find /tmp/trp -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r3_train_step_b32_bf16_kernel_stats.csv
f=$(find /tmp/trp -name "*kernel_trace.csv" | head -1); python tests/pp_tgemm_trace.py "$f" > gpurun_out/r3_tgemm_by_shape.txt 2>&1
tail -3 gpurun_out/r3_tests_gpu_d.log; cat gpurun_out/r3_bench_graph_ab.txt; grep step gpurun_out/r3_train_probe_bf16.log
