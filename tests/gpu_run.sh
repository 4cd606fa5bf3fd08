#!/bin/bash
# The ONE GPU-side runner (development tool; run through gpurun from the repo root):
#     gpurun --timeout 900 -- 'bash tests/gpu_run.sh TAG task [task ...]'
# Every task appends to gpurun_out/TAG_<task>.log / writes gpurun_out/TAG_<task>.*; what is worth keeping is copied to profiles/ by hand.
# Tasks (each bounded by its own `timeout`):
#   ops            pytest tests/test_ops.py -m gpu
#   suite          the whole -m gpu suite + smoke()
#   parity         the headline parity transcript (pytest -s -k "headline or ten_minute or ddim_full")
#   ab:L1,L2[@B1,B2]  DDIM step of the in-tree library against tests/var/L*/libmugd.so, alternating, twice per batch size (default 4,8,16)
#   wide           per-launch tables with MUGD_CONV_WIDE = 0 | 1 | 2 at batch 8 and 16, A/B/A/B
#   bench          the driver's command (bench.py --gpus 1 --steps 20 --warmup 3)
#   stats          rocprofv3 --kernel-trace --stats of a short bench run
#   kstats:LIB[@B] rocprofv3 --kernel-trace --stats of tests/gpu_probe.py --quick, in-tree library and tests/var/LIB on one box
#   shapepmc:LIB@I,J   executed instructions per wave of single launch shapes, in-tree library vs tests/var/LIB
#   convbench:L1,L2    us per launch of every hot conv shape (tests/gpu_convbench.py --compare), in-tree library and tests/var/L*
#   traffic        FETCH_SIZE / WRITE_SIZE of conv_gemm (separate --pmc passes) -> TAG_conv_traffic.json
#   train_traffic  FETCH_SIZE / WRITE_SIZE of a bf16 batch-32 training step by kernel class (tests/pp_train_pmc.py)
#   train_stats    rocprofv3 --kernel-trace --stats of the bf16 batch-32 training step (tests/gpu_train_probe.py)
#   mfma           matrix-pipe busy / VALU counters of one U-Net evaluation
#   instr          executed instructions per wave by class (the r4_pmc_instr_per_wave table)
#   layers:B       per-launch table of one U-Net / VAE / wave evaluation at batch B (MUGD_PROFILE_CSV)
#   probe:B        tests/gpu_probe.py at batch B (all pieces of the pipeline)
#   timeline:B     per-wave phase timeline of one U-Net evaluation at batch B (development library tests/tl/libmugd_tl.so)
#   overlap        tests/gpu_overlap_probe.hip: a chain of dependent conv-shaped launches in order vs released early + epoch polling (round 6)
#   icache         tests/gpu_convbench.py --icache: per hot shape, launches with warm vs evicted instruction caches (round 6)
#   forms          tests/gpu_convbench.py --forms: M-split geometries vs the host's choice on the tall-M launches (round 6)
#   hostq          host microseconds to enqueue one U-Net evaluation (mugd_net_host_enqueue), in-tree library
#   envab:K=a,b[@B1,B2]  DDIM step (tests/gpu_probe.py --quick) with K=a and K=b alternating, six times per batch size (default 4,8,16)
#   env:K=V        export K=V for the tasks that follow
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=$1; shift
O=gpurun_out/$TAG
# the library is built in the authoring container and travels with the snapshot; build here only if it is missing
[ -f mug-diffusion_amd/libmugd.so ] || python mug-diffusion_amd/build.py > ${O}_build.log 2>&1 || { tail -30 ${O}_build.log; exit 1; }

probe_line() { grep -E "ms/step|total \(event|conv_gemm|attention|s4_conv|unet forward"; }

for task in "$@"; do
  arg=${task#*:}; name=${task%%:*}
  echo "=== $task"
  case $name in
    env) export "$arg" ;;
    envab)
      spec=${arg%%@*}; bs=4,8,16; [ "$spec" != "$arg" ] && bs=${arg#*@}
      var=${spec%%=*}; vals=${spec#*=}
      F=${O}_envab_$var.txt; : > $F
      for B in $(echo $bs | tr ',' ' '); do for rep in 1 2 3 4 5 6; do for v in $(echo $vals | tr ',' ' '); do
        echo "== B=$B $var=$v" >> $F
        env $var=$v timeout 300 python tests/gpu_probe.py --B $B --quick 2>&1 | grep -E "ms/step" >> $F
      done; done; done
      python - "$F" <<'PY'
import re, sys, collections
d, k = collections.OrderedDict(), None
for line in open(sys.argv[1]):
    m = re.match(r'== (B=\d+) (\S+)', line)
    if m:
        k = (m.group(1), m.group(2)); continue
    m = re.search(r'([\d.]+) ms/step', line)
    if m and k:
        d.setdefault(k, []).append(float(m.group(1)))
with open(sys.argv[1], 'a') as f:
    for k, v in d.items():
        line = "%s %s: mean %.4f ms/step over %d runs %s" % (k[0], k[1], sum(v) / len(v), len(v), v)
        print(line); f.write("# " + line + "\n")
PY
      ;;
    overlap)
      ( hipcc --offload-arch=gfx950 -O3 -o /tmp/overlap_probe tests/gpu_overlap_probe.hip && timeout 300 /tmp/overlap_probe 400 ) > ${O}_overlap.txt 2>&1; cat ${O}_overlap.txt ;;
    icache)
      timeout 600 python tests/gpu_convbench.py --icache > ${O}_icache.txt 2>&1; cat ${O}_icache.txt
      for v in $IC_OTHERS; do MUGD_LIB_LENIENT=1 MUGD_LIB_PATH=$PWD/tests/var/$v/libmugd.so timeout 600 python tests/gpu_convbench.py --icache > ${O}_icache_$v.txt 2>&1; cat ${O}_icache_$v.txt; done ;;
    forms)
      timeout 600 python tests/gpu_convbench.py --forms > ${O}_forms.txt 2>&1; cat ${O}_forms.txt ;;
    hostq)
      timeout 300 python - > ${O}_hostq.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "mug-diffusion_amd"))
import torch
from oracle import cases, weights
from mug._native import get_lib
lib = get_lib(); case = cases.FULL
man = weights.load_manifest(os.path.join(cases.GOLDEN, case["manifest"]))
for B in (4, 16):
    sd = weights.set_s4_lengths(weights.make_state_dict(man, 0), case["unet"], 512)
    u = lib.unet(case["unet"]); u.set_params(sd, "model.unet_model.")
    x = cases.x_T(1, B, 512).to(lib.device); t = torch.full((B,), 501, dtype=torch.long, device=lib.device)
    c = cases.context(case, 1, B).to(lib.device); w = [m.to(lib.device) for m in cases.audio_maps(case, 1, 1, 512)]
    u.forward(x, t, c, w); torch.cuda.synchronize()
    for rep in range(3):
        us, n = u.host_enqueue(5)
        print("B=%d: host enqueue %.1f us per U-Net evaluation, %d ops, %.2f us per launch" % (B, us, n, us / n), flush=True)
    u.close()
PY
      cat ${O}_hostq.txt ;;
    ops)
      ( time timeout 900 python -m pytest tests/test_ops.py -x -q -m gpu -p no:cacheprovider ) > ${O}_ops.log 2>&1; tail -5 ${O}_ops.log ;;
    suite)
      ( time timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > ${O}_suite.log 2>&1; tail -6 ${O}_suite.log
      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > ${O}_smoke.log 2>&1; tail -1 ${O}_smoke.log ;;
    parity)
      ( timeout 1200 python -m pytest tests/test_nets.py -q -s -m gpu -p no:cacheprovider -k "headline or ten_minute or ddim_full" ) > ${O}_parity.log 2>&1
      grep -E "flip|latent|passed|failed|error" ${O}_parity.log | tail -40 ;;
    ab)
      # ab:LIB1,LIB2,...[@B1,B2,...]: the in-tree library and each tests/var/LIB/libmugd.so alternating, twice, per batch size
      libs=${arg%%@*}; bs=4,8,16; [ "$libs" != "$arg" ] && bs=${arg#*@}
      L=${O}_ab_$(echo $libs | tr ',' '_').log; : > $L
      for B in $(echo $bs | tr ',' ' '); do for rep in 1 2; do for v in new $(echo $libs | tr ',' ' '); do
        echo "== B=$B lib=$v" >> $L
        if [ $v = new ]; then timeout 300 python tests/gpu_probe.py --B $B --quick 2>&1 | probe_line >> $L
        else MUGD_LIB_LENIENT=1 MUGD_LIB_PATH=$PWD/tests/var/$v/libmugd.so timeout 300 python tests/gpu_probe.py --B $B --quick 2>&1 | probe_line >> $L; fi
      done; done; done
      grep -E "^==|ms/step" $L ;;
    wide)
      : > ${O}_wide.log
      for B in 16 8; do for v in 0 1 2 0 1 2; do
        echo "== B=$B MUGD_CONV_WIDE=$v" >> ${O}_wide.log
        rm -f ${O}_wide_layers_b${B}_w$v.csv
        MUGD_CONV_WIDE=$v MUGD_PROFILE_CSV=${O}_wide_layers_b${B}_w$v.csv timeout 300 python tests/gpu_probe.py --B $B --quick 2>&1 | probe_line >> ${O}_wide.log
      done; done
      grep -E "^==|ms/step" ${O}_wide.log ;;
    bench)
      ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 ) > ${O}_bench.log 2>&1; grep '^{' ${O}_bench.log > ${O}_bench.json
      python - "$O" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + '_bench.json'))
r = d['roofline']
print('bench', d['value'], d['unit'], d['ms_per_step'], 'ms; ddim loop', d.get('ddim_loop_ms'), 'clock', d.get('shader_clock_mhz_under_matrix_load'))
print('roofline', r['achieved'], r['unit'], 'frac', r['frac'], 'avg launch us', r.get('avg_launch_us'), 'traffic', r.get('traffic'))
for k in ('throughput_mode', 'throughput_mode_4_songs', 'cfg_scale_5', 'training_step', 'training_step_fp32', 'reduced_precision_mode', 'cpu_baseline'):
    v = d.get(k) or {}
    print(k, v.get('value'), v.get('unit'), v.get('unet_sample_steps_per_s', ''))
PY
      ;;
    stats)
      rm -rf /tmp/prof_$TAG
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $OLDPWD/bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-training-step --no-throughput-mode --no-reduced-mode) > ${O}_bench_under_rocprof.log 2>&1
      f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1); cp "$f" ${O}_bench_kernel_stats.csv; head -8 ${O}_bench_kernel_stats.csv | cut -c1-180
      grep '^{' ${O}_bench_under_rocprof.log > ${O}_bench_under_rocprof.json ;;
    kstats)
      # kstats:LIB[@B]: rocprofv3 per-kernel statistics of tests/gpu_probe.py --quick for the in-tree library and tests/var/LIB, same box
      lb=${arg%%@*}; B=4; [ "$lb" != "$arg" ] && B=${arg#*@}
      for v in new $lb; do
        rm -rf /tmp/ks_${TAG}_$v
        if [ $v = new ]; then (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_${TAG}_$v -- python $OLDPWD/tests/gpu_probe.py --B $B --quick) > ${O}_kstats_$v.log 2>&1
        else (cd /tmp && MUGD_LIB_LENIENT=1 MUGD_LIB_PATH=$OLDPWD/tests/var/$v/libmugd.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_${TAG}_$v -- python $OLDPWD/tests/gpu_probe.py --B $B --quick) > ${O}_kstats_$v.log 2>&1; fi
        f=$(find /tmp/ks_${TAG}_$v -name "*kernel_stats.csv" | head -1); cp "$f" ${O}_kstats_${v}_b$B.csv
        grep "ms/step" ${O}_kstats_$v.log
      done ;;
    shapepmc)
      # shapepmc:LIB@I,J,...: executed instructions per wave of single conv_gemm launch shapes (tests/gpu_convbench.py --pmc --shape I), in-tree library vs tests/var/LIB
      lb=${arg%%@*}; shapes=${arg#*@}; : > ${O}_shape_pmc.txt
      for shape in $(echo $shapes | tr ',' ' '); do for v in new $lb; do
        for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVES SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM" "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
          d=/tmp/spmc_${TAG}_${shape}_${v}_$(echo $set | tr ' ' '_'); rm -rf $d
          if [ $v = new ]; then (cd /tmp && timeout 120 rocprofv3 --pmc $set --output-format csv -d $d -- python $OLDPWD/tests/gpu_convbench.py --pmc --shape $shape) > /dev/null 2>&1
          else (cd /tmp && MUGD_LIB_LENIENT=1 MUGD_LIB_PATH=$OLDPWD/tests/var/$v/libmugd.so timeout 120 rocprofv3 --pmc $set --output-format csv -d $d -- python $OLDPWD/tests/gpu_convbench.py --pmc --shape $shape) > /dev/null 2>&1; fi
          echo "== shape $shape lib $v" >> ${O}_shape_pmc.txt
          python tests/pmc_generic_summary.py $d 2>/dev/null | grep -E "^kernel|conv_gemm" >> ${O}_shape_pmc.txt
        done
      done; done
      cat ${O}_shape_pmc.txt ;;
    convbench)
      # convbench:LIB1,LIB2: tests/gpu_convbench.py --compare (us per launch of every hot shape) for the in-tree library and each tests/var/LIB
      python tests/gpu_convbench.py --compare > ${O}_cb_new.txt 2>&1
      for v in $(echo $arg | tr ',' ' '); do MUGD_LIB_LENIENT=1 MUGD_LIB_PATH=$PWD/tests/var/$v/libmugd.so python tests/gpu_convbench.py --compare > ${O}_cb_$v.txt 2>&1; done
      paste ${O}_cb_*.txt | head -40 ;;
    traffic)
      rm -rf /tmp/pmc_$TAG
      (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_$TAG/fetch -- python $OLDPWD/tests/gpu_unet_once.py --n 2) > ${O}_pmc_fetch.log 2>&1
      (cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_$TAG/write -- python $OLDPWD/tests/gpu_unet_once.py --n 2) > ${O}_pmc_write.log 2>&1
      python tests/pmc_summary.py /tmp/pmc_$TAG ${O}_conv_traffic.json | tail -8 ;;
    train_stats)
      rm -rf /tmp/ts_$TAG
      (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ts_$TAG -- python $OLDPWD/tests/gpu_train_probe.py --B 32 --reps 4 --bf16 --adamw) > ${O}_train_stats.log 2>&1
      f=$(find /tmp/ts_$TAG -name "*kernel_stats.csv" | head -1); cp "$f" ${O}_train_kernel_stats.csv; grep -i "step" ${O}_train_stats.log | tail -3
      python tests/pp_train_trace.py /tmp/ts_$TAG ${O}_train_step_digest.txt; head -60 ${O}_train_step_digest.txt ;;
    train_traffic)
      rm -rf /tmp/pmct_$TAG
      (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmct_$TAG/fetch -- python $OLDPWD/tests/gpu_train_probe.py --B 32 --reps 2 --bf16 --adamw) > ${O}_pmc_train_fetch.log 2>&1
      (cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmct_$TAG/write -- python $OLDPWD/tests/gpu_train_probe.py --B 32 --reps 2 --bf16 --adamw) > ${O}_pmc_train_write.log 2>&1
      python tests/pp_train_pmc.py /tmp/pmct_$TAG 2 > ${O}_train_hbm_traffic.txt 2>&1; cat ${O}_train_hbm_traffic.txt; grep step ${O}_pmc_train_fetch.log | tail -1 ;;
    mfma)
      rm -rf /tmp/pmcm_$TAG
      (cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d /tmp/pmcm_$TAG -- python $OLDPWD/tests/gpu_unet_once.py --n 2) > ${O}_pmc_mfma.log 2>&1
      python tests/pmc_mfma_summary.py /tmp/pmcm_$TAG ${O}_pmc_unet_mfma.txt | head -14 ;;
    instr)
      rm -rf /tmp/pmci_$TAG; : > ${O}_pmc_instr_per_wave.txt
      for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_MFMA" "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
        d=/tmp/pmci_$TAG/$(echo $set | tr ' ' '_')
        (cd /tmp && timeout 300 rocprofv3 --pmc $set --output-format csv -d $d -- python $OLDPWD/tests/gpu_unet_once.py --n 2) >> ${O}_pmc_instr.log 2>&1
        python tests/pmc_generic_summary.py $d >> ${O}_pmc_instr_per_wave.txt 2>&1
      done
      head -60 ${O}_pmc_instr_per_wave.txt ;;
    layers)
      rm -f ${O}_per_layer_z512_b$arg.csv
      MUGD_PROFILE_CSV=${O}_per_layer_z512_b$arg.csv timeout 300 python tests/gpu_probe.py --B $arg > ${O}_probe_b$arg.txt 2>&1
      grep -E "ddim|vae decode|wave encode|log-mel|unet forward|TFLOP" ${O}_probe_b$arg.txt ;;
    timeline)
      # timeline:B -- per-wave phase stamps of every conv_gemm launch of one U-Net evaluation (tests/tl/libmugd_tl.so: build.py --tl, built in the authoring container)
      timeout 600 python tests/gpu_timeline.py --z 512 --B $arg --out ${O}_timeline_z512_b$arg.csv > /dev/null 2>${O}_timeline_err.log; head -16 ${O}_timeline_z512_b$arg.txt
      for v in $TL_OTHERS; do      # env:TL_OTHERS="r5tl ..." -- the same with tests/var/<name>/libmugd.so (a -DMUGD_TL build of another revision)
        timeout 600 python tests/gpu_timeline.py --z 512 --B $arg --lib tests/var/$v/libmugd.so --out ${O}_timeline_${v}_z512_b$arg.csv > /dev/null 2>>${O}_timeline_err.log; head -16 ${O}_timeline_${v}_z512_b$arg.txt
      done ;;
    probe)
      timeout 300 python tests/gpu_probe.py --B $arg > ${O}_probe_b$arg.txt 2>&1; grep -E "ddim|vae decode|wave encode|log-mel|unet forward|TFLOP" ${O}_probe_b$arg.txt ;;
    *) echo "unknown task $task" ;;
  esac
done
