#!/bin/bash
# round 4, call 10: conv_gemm on the f16 matrix cores with split operands (H3) against the fp32-MFMA build: parity, then A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_ops.py tests/test_nets.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r4_t10.log 2>&1; tail -4 gpurun_out/r4_t10.log
for rep in 1 2; do
for v in fp32mfma h3; do
  if [ $v = h3 ]; then unset MUGD_LIB_PATH; else export MUGD_LIB_PATH=$GRAFT_REPO_ROOT/tests/var/$v/libmugd.so; fi
  for B in 4 8 16; do
    timeout 200 python tests/gpu_probe.py --B $B --quick > gpurun_out/r4_h3_${v}_b$B.txt 2>&1
    echo "rep$rep $v B=$B: $(grep -E 'ddim_graph' gpurun_out/r4_h3_${v}_b$B.txt | cut -c1-90) | $(grep -E '  conv_gemm  ' gpurun_out/r4_h3_${v}_b$B.txt | cut -c1-100)"
  done
done
done
unset MUGD_LIB_PATH
