#!/bin/bash
# executed instructions per wave of a small (proj 1x1, 256 tiles) and a GroupNorm 3-tap (res.l0) conv_gemm launch
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for shape in 10 4; do
rm -rf /tmp/pmc28
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_IFETCH"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc28/p$i -- python $GRAFT_REPO_ROOT/tests/gpu_convbench.py --pmc --shape $shape --tn 32 --wk 8 > /tmp/pmc28_$i.log 2>&1
  tail -1 /tmp/pmc28_$i.log
done
echo "== shape $shape" >> $GRAFT_REPO_ROOT/gpurun_out/r4_pmc_instr.txt
python $GRAFT_REPO_ROOT/tests/pmc_raw_summary.py /tmp/pmc28 conv_gemm >> $GRAFT_REPO_ROOT/gpurun_out/r4_pmc_instr.txt
done
cat $GRAFT_REPO_ROOT/gpurun_out/r4_pmc_instr.txt
