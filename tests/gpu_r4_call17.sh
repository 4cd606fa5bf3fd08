#!/bin/bash
# paired-frame STFT: parity on the GPU, log-mel time old (tests/var/head) vs new, kernel trace of the new one
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r4_stft.log
: > $L
( timeout 300 python -m pytest tests/test_ops.py tests/test_mirror.py -x -q -m gpu -p no:cacheprovider -k "mel or resampl or audio" 2>&1 | tail -4 ) >> $L 2>&1
cat > /tmp/melt.py <<'PY'
import sys, time, torch
sys.path.insert(0, "mug-diffusion_amd"); sys.path.insert(0, ".")
from mug import _native
from oracle import host
lib = _native.get_lib()
pcm = torch.from_numpy(host.synth_audio(180.0)).cuda()
out = lib.log_mel(pcm); torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
with lib.on_stream():
    for _ in range(3): lib.log_mel(pcm)
    ev[0].record()
    for _ in range(20): lib.log_mel(pcm)
    ev[1].record()
torch.cuda.synchronize()
print("log-mel of 180 s (%d frames): %.1f us per call" % (out.shape[1], ev[0].elapsed_time(ev[1]) / 20 * 1e3), flush=True)
print("checksum %.6f" % out.double().sum().item())
PY
for v in new head new head; do
  if [ $v = head ]; then export MUGD_LIB_PATH=$PWD/tests/var/head/libmugd.so; else unset MUGD_LIB_PATH; fi
  echo "== lib=$v" >> $L
  timeout 200 python /tmp/melt.py >> $L 2>&1
done
unset MUGD_LIB_PATH
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_stft -o stft -- python $GRAFT_REPO_ROOT/../melt.py > /dev/null 2>&1 || timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_stft -o stft -- bash -c "cd $GRAFT_REPO_ROOT && python /tmp/melt.py" > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_stft -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -8 "$f" | cut -c1-200 >> $L
cat $L
