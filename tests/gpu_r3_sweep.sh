#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python mug-diffusion_amd/build.py > gpurun_out/r3_build.log 2>&1
out=gpurun_out/r3_twgrad_split_sweep.txt
: > $out
run() { echo "== $*" >> $out; env "$@" timeout 200 python tests/gpu_train_probe.py --B 32 --reps 5 --bf16 --adamw 2>&1 | grep "step 4" >> $out; }
run A=0
run MUGD_TWGRAD_MINSLABS=4
run MUGD_TWGRAD_MINSLABS=2
run MUGD_TWGRAD_FREE_MB=16
run MUGD_TWGRAD_FREE_MB=64
run MUGD_TWGRAD_FREE_MB=64 MUGD_TWGRAD_MINSLABS=4
run MUGD_TWGRAD_FREE_MB=64 MUGD_TWGRAD_WGS=1024
run MUGD_TWGRAD_FREE_MB=64 MUGD_TWGRAD_WGS=512
cat $out
