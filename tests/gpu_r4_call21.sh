#!/bin/bash
# k_conv16.hip folded into the conv_tile template: parity on the GPU + A/B against the two-file build (tests/var/head)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r4_fold_ab.log
: > $L
( timeout 400 python -m pytest tests/test_ops.py -x -q -m gpu -p no:cacheprovider -k "conv or linear or xattn or glu" 2>&1 | tail -3 ) >> $L 2>&1
( timeout 400 python -m pytest tests/test_nets.py -x -q -m gpu -p no:cacheprovider -k "unet_vs_reference or vae_decode or reduced_precision" 2>&1 | tail -3 ) >> $L 2>&1
for v in new head new head; do
  if [ $v = head ]; then export MUGD_LIB_PATH=$PWD/tests/var/head/libmugd.so; else unset MUGD_LIB_PATH; fi
  echo "== B=4 lib=$v" >> $L
  timeout 300 python tests/gpu_probe.py --B 4 --quick 2>&1 | grep -E "ms/step|total \(event" >> $L
done
unset MUGD_LIB_PATH
cat $L
