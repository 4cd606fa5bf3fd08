#!/bin/bash
# round 4, call 6: 64 x 32 (TALL) conv tiles: parity, then A/B at batch 16 / 8 and on the wave encoder / VAE
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python mug-diffusion_amd/build.py > gpurun_out/r4_build6.log 2>&1 || { tail -30 gpurun_out/r4_build6.log; exit 1; }
timeout 300 python -m pytest tests/test_ops.py -x -q -m gpu -p no:cacheprovider -k "tall or conv1d or norm_conv" > gpurun_out/r4_t6.log 2>&1; tail -3 gpurun_out/r4_t6.log
for tall in 0 auto; do
  if [ $tall = 0 ]; then export MUGD_CONV_TALL=0; else unset MUGD_CONV_TALL; fi
  for B in 16 8; do
    timeout 200 python tests/gpu_probe.py --B $B --quick > gpurun_out/r4_tall_${tall}_b$B.txt 2>&1
    echo "tall=$tall B=$B: $(grep -E 'ddim_graph' gpurun_out/r4_tall_${tall}_b$B.txt) | $(grep -E '  conv_gemm  ' gpurun_out/r4_tall_${tall}_b$B.txt)"
  done
  timeout 200 python tests/gpu_probe.py --B 4 > gpurun_out/r4_tall_${tall}_b4_full.txt 2>&1
  echo "tall=$tall B=4: $(grep -E 'ddim_eager|vae decode|wave encode' gpurun_out/r4_tall_${tall}_b4_full.txt | tr '\n' '|')"
done
unset MUGD_CONV_TALL
timeout 400 python -m pytest tests/test_nets.py -x -q -m gpu -p no:cacheprovider -k "wave or vae or unet_forward" > gpurun_out/r4_t6b.log 2>&1; tail -3 gpurun_out/r4_t6b.log
