#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python mug-diffusion_amd/build.py > gpurun_out/r3_build.log 2>&1
rm -rf /tmp/pmc
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc/fetch -- python $GRAFT_REPO_ROOT/tests/gpu_unet_once.py --n 2) > gpurun_out/r3_pmc_fetch.log 2>&1
(cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc/write -- python $GRAFT_REPO_ROOT/tests/gpu_unet_once.py --n 2) > gpurun_out/r3_pmc_write.log 2>&1
python tests/pmc_summary.py /tmp/pmc gpurun_out/r3_conv_traffic.json | head -5
tail -2 gpurun_out/r3_pmc_fetch.log
