#!/bin/bash
# kernel arguments in device memory (HIP_FORCE_DEV_KERNARG=1) vs the runtime default, one box, A/B/A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r4_kernarg_ab.log
: > $L
for v in default 1 0 default 1 0; do
  if [ $v = default ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  echo "== HIP_FORCE_DEV_KERNARG=$v" >> $L
  timeout 300 python tests/gpu_probe.py --B 4 --quick 2>&1 | grep -E "ms/step|total \(event|unet forward" >> $L
done
unset HIP_FORCE_DEV_KERNARG
cat $L
