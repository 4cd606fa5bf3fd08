#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python mug-diffusion_amd/build.py > gpurun_out/r3_build.log 2>&1
(time timeout 1500 python -m pytest tests/ -x -q -m gpu) > gpurun_out/r3_gpu_suite.log 2>&1; echo "rc $?" >> gpurun_out/r3_gpu_suite.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke.log 2>&1; echo "rc $?" >> gpurun_out/r3_smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.log; echo "bench rc $?" >> gpurun_out/r3_bench.log
rm -rf /tmp/bprof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bprof -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-throughput-mode --no-reduced-mode --no-training-step) > gpurun_out/r3_bench_under_rocprof.json 2> gpurun_out/r3_bench_under_rocprof.log
f=$(find /tmp/bprof -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r3_bench_kernel_stats.csv
timeout 300 python tests/gpu_train_probe.py --B 32 --reps 6 --bf16 --adamw > gpurun_out/r3_train_probe_bf16.log 2>&1
rm -rf /tmp/trp
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trp -- python $GRAFT_REPO_ROOT/tests/gpu_train_probe.py --B 32 --reps 4 --bf16 --adamw) > gpurun_out/r3_train_probe_bf16_prof.log 2>&1
f=$(find /tmp/trp -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r3_train_step_b32_bf16_kernel_stats.csv
f=$(find /tmp/trp -name "*kernel_trace.csv" | head -1); python tests/pp_tgemm_trace.py "$f" > gpurun_out/r3_tgemm_by_shape.txt 2>&1
tail -4 gpurun_out/r3_gpu_suite.log; tail -2 gpurun_out/r3_smoke.log; tail -3 gpurun_out/r3_bench.log; head -c 400 gpurun_out/r3_bench.json; echo; grep "step" gpurun_out/r3_train_probe_bf16.log | tail -2
