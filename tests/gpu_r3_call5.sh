#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python mug-diffusion_amd/build.py > gpurun_out/r3_build.log 2>&1
timeout 900 python -m pytest tests/test_train.py -x -q -m gpu -k "transformer or shipped or adamw" > gpurun_out/r3_train_tests_gpu_c.log 2>&1; echo "rc $?" >> gpurun_out/r3_train_tests_gpu_c.log
timeout 300 python tests/gpu_train_probe.py --B 32 --reps 4 --adamw --bf16 > gpurun_out/r3_train_probe_bf16.log 2>&1
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r3_bench_a.json 2> gpurun_out/r3_bench_a.log; echo "bench rc $?" >> gpurun_out/r3_bench_a.log
tail -3 gpurun_out/r3_train_tests_gpu_c.log; grep step gpurun_out/r3_train_probe_bf16.log; tail -5 gpurun_out/r3_bench_a.log; cat gpurun_out/r3_bench_a.json | head -c 3000
