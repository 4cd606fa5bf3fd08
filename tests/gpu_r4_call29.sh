#!/bin/bash
# the M-split ("wide") conv_gemm form, forced wherever it exists, against the K-split default: per-layer tables at batch 8 and 16
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r4_wide_ab.log
: > $L
( MUGD_CONV_WIDE=1 timeout 300 python -m pytest tests/test_ops.py -x -q -m gpu -p no:cacheprovider -k "wide" 2>&1 | tail -2 ) >> $L 2>&1
for B in 16 8; do
for v in 0 1 0 1; do
  echo "== B=$B MUGD_CONV_WIDE=$v" >> $L
  rm -f gpurun_out/r4_wide_layers_b${B}_w$v.csv
  MUGD_CONV_WIDE=$v MUGD_PROFILE_CSV=gpurun_out/r4_wide_layers_b${B}_w$v.csv timeout 300 python tests/gpu_probe.py --B $B --quick 2>&1 | grep -E "ms/step|total \(event|conv_gemm" >> $L
done
done
cat $L
