#!/bin/bash
# argument loads batched and held (SGPR_HOLD) + KARG_WARM, against the build before both (tests/var/nowarm)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r4_sgpr_hold_ab.log
: > $L
( timeout 300 python -m pytest tests/test_ops.py -x -q -m gpu -p no:cacheprovider -k "conv or linear or xattn or glu or norm" 2>&1 | tail -2 ) >> $L 2>&1
for B in 4 8 16; do
for v in new before new before; do
  if [ $v = before ]; then export MUGD_LIB_PATH=$PWD/tests/var/nowarm/libmugd.so; else unset MUGD_LIB_PATH; fi
  echo "== B=$B lib=$v" >> $L
  timeout 300 python tests/gpu_probe.py --B $B --quick 2>&1 | grep -E "ms/step|total \(event" >> $L
done
done
unset MUGD_LIB_PATH
cat $L
