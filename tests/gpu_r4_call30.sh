#!/bin/bash
# sanity of the final library after the M-split form went in (default paths): operator tests, network goldens, smoke, a short bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
L=gpurun_out/r4_final_sanity.log
: > $L
( timeout 200 python -m pytest tests/test_ops.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2 ) >> $L 2>&1
( timeout 300 python -m pytest tests/test_nets.py -x -q -m gpu -p no:cacheprovider -k "unet_forward or ddim or vae_decode or wave_encoder" 2>&1 | tail -2 ) >> $L 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> $L
python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-training-step --no-reduced-mode 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('bench', round(d['value'],2), round(d['ms_per_step'],1), 'ddim', round(d['ddim_loop_ms'],1), 'tp2', round(d.get('throughput_mode',{}).get('value',0),1), 'tp4', round(d.get('throughput_mode_4_songs',{}).get('value',0),1))" >> $L
cat $L
