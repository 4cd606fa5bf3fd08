#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python mug-diffusion_amd/build.py > gpurun_out/r3_build.log 2>&1
(time timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5) > gpurun_out/r3_bench_driver_cmd.json 2> gpurun_out/r3_bench_driver_cmd.log
tail -4 gpurun_out/r3_bench_driver_cmd.log; head -c 300 gpurun_out/r3_bench_driver_cmd.json
