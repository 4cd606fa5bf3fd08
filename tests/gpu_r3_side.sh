#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python mug-diffusion_amd/build.py > gpurun_out/r3_build.log 2>&1
out=gpurun_out/r3_side_stream.txt
: > $out
MUGD_TRAIN_SIDE=1 timeout 600 python -m pytest tests/test_train.py -x -q -m gpu -p no:cacheprovider -k "bracket or bf16" 2>&1 | tail -2 >> $out
for m in 0 1 0 1; do
  echo "== MUGD_TRAIN_SIDE=$m" >> $out
  MUGD_TRAIN_SIDE=$m timeout 300 python tests/gpu_train_probe.py --B 32 --reps 6 --bf16 --adamw 2>&1 | grep "step 5" >> $out
done
echo "== B=2" >> $out
for m in 0 1; do MUGD_TRAIN_SIDE=$m timeout 300 python tests/gpu_train_probe.py --B 2 --reps 6 --bf16 --adamw 2>&1 | grep "step 5" >> $out; done
cat $out
