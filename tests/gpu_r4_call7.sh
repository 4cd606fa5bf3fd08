#!/bin/bash
# round 4, call 7: are the two waves of a SIMD in lockstep in the conv K loop?  variants: s_setprio around the MFMA cluster, initial half-iteration skew
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for rep in 1 2; do
for v in base prio skew20; do
  if [ $v = base ]; then unset MUGD_LIB_PATH; else export MUGD_LIB_PATH=$GRAFT_REPO_ROOT/tests/var/$v/libmugd.so; fi
  for B in 4 8; do
    timeout 200 python tests/gpu_probe.py --B $B --quick > gpurun_out/r4_v_${v}_b$B.txt 2>&1
    echo "rep$rep $v B=$B: $(grep -E 'ddim_graph' gpurun_out/r4_v_${v}_b$B.txt | cut -c1-90) | $(grep -E '  conv_gemm  ' gpurun_out/r4_v_${v}_b$B.txt | cut -c1-100)"
  done
done
done
