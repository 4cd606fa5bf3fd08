#!/bin/bash
# round 4, call 1: phase-0 fixes on the GPU, the RCCL path at world = 1 (verdict item 4), executor design probe, batch-8 baseline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python mug-diffusion_amd/build.py > gpurun_out/r4_build.log 2>&1
timeout 120 tests/bin/xcd_exec_probe > gpurun_out/r4_xcd_exec_probe.txt 2>&1; echo "probe rc $?"; cat gpurun_out/r4_xcd_exec_probe.txt
timeout 300 python -m pytest tests/test_shard.py tests/test_train.py -x -q -m gpu -p no:cacheprovider -k "rccl or fit_three or replacing or bracket" > gpurun_out/r4_t1.log 2>&1; tail -3 gpurun_out/r4_t1.log
# bench under torch.distributed.run with ONE rank: nccl init, barrier, gather_grids, BucketedAllReduce (even_single) in the training leg
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-reduced-mode ) > gpurun_out/r4_rccl_world1_bench.txt 2>&1; echo "dist bench rc $?"
( time python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-reduced-mode --no-training-step ) > gpurun_out/r4_plain_bench.txt 2>&1; echo "plain bench rc $?"
grep -h '^{' gpurun_out/r4_rccl_world1_bench.txt gpurun_out/r4_plain_bench.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], d['ms_per_step'], d.get('throughput_mode', {}).get('value'), d.get('throughput_mode_4_songs', {}).get('value'), (d.get('training_step') or {}).get('value'))"
MUGD_PROFILE_CSV=gpurun_out/r4_per_layer_z512_b8_base.csv timeout 200 python tests/gpu_probe.py --B 8 --quick > gpurun_out/r4_probe_b8.txt 2>&1; tail -15 gpurun_out/r4_probe_b8.txt
