#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python mug-diffusion_amd/build.py > gpurun_out/r3_build.log 2>&1
rm -rf /tmp/pmct
(cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmct/fetch -- python $GRAFT_REPO_ROOT/tests/gpu_train_probe.py --B 32 --reps 2 --bf16 --adamw) > gpurun_out/r3_pmc_train_fetch.log 2>&1
(cd /tmp && timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmct/write -- python $GRAFT_REPO_ROOT/tests/gpu_train_probe.py --B 32 --reps 2 --bf16 --adamw) > gpurun_out/r3_pmc_train_write.log 2>&1
python tests/pp_train_pmc.py /tmp/pmct 2 > gpurun_out/r3_train_hbm_traffic.txt 2>&1
cat gpurun_out/r3_train_hbm_traffic.txt
grep step gpurun_out/r3_pmc_train_fetch.log | tail -1
