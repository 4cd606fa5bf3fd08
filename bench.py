#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): charts/s for 3-minute audio, 50 DDIM steps, batch 4, on N MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`, or as the plain
   command above -- without WORLD_SIZE in the environment bench.py starts its own N ranks through torch.distributed.run)

One STEP = one batch of `--batch` charts for one synthetic 3-minute 22.05 kHz song, i.e. one full pass of
the hot path with the PCM already resident in HBM:
    log-mel (HIP FFT + mel GEMM) -> wave encoder (once per song, shared by the seeds) -> prompt embedding
    -> 50-step DDIM loop over the U-Net (one native call, eager launches) -> VAE decode -> thresholded 4K note grid.
Every rank works on its own (song, seeds) units; there is no data-path collective (weak scaling).
Weights are seeded synthetic values of the shipped architecture (no checkpoint exists offline), with the
reference's zero-initialised tensors randomised so no branch is a no-op.  Tensors and accumulation are fp32 end to end like
the reference; conv / Linear products run on the f16 matrix cores with split, block-scaled operands (fp32-equivalent over the
fp32 range: include/mugd.h, mugd_version).

Prints ONE JSON line (rank 0) with the contract's fields plus
  "roofline"     : the dominant kernel (conv_gemm, MFMA-bound) measured live with HIP events,
  "cpu_baseline" : the oracle (PyTorch-CPU fp32 restatement of the reference path) timed on this host's cores
                   on a bounded sample of the same workload (a reported baseline, not the target).
"""
import argparse
import faulthandler
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mug-diffusion_amd"))
sys.path.insert(1, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense f16 / bf16 MFMA (v_mfma_f32_32x32x16_f16)
# conv_gemm since round 4: every fp32-equivalent product block is THREE f16 MFMAs on split operands (csrc/conv_body.h: H3), so the matrix
# pipe's ceiling for ALGORITHMIC (2 M K N) flops is a third of the dense f16 peak
PEAK_H3_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0
PEAK_HBM_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E, 8 TB/s nominal (~6.3 TB/s achievable on a streaming kernel)
UNET_GFLOP_PER_SAMPLE_STEP = 22.37     # BASELINE.md section 2 (z = 512)

SHIPPED = dict(          # configs/mug/mug_diffusion.yaml of the reference (shapes only)
    unet=dict(in_channels=16, model_channels=128, out_channels=16, attention_resolutions=[8, 4, 2], num_res_blocks=2,
              channel_mult=[1, 2, 3, 4], num_heads=8, context_dim=128, dropout=0.0, lstm_last=False, lstm_layer=False,
              s4_layer=True, audio_channels=[256, 512, 512, 512], use_checkpoint=False),
    vae=dict(x_channels=16, middle_channels=64, z_channels=16, num_groups=8, channel_mult=[1, 2, 4, 4], num_res_blocks=1),
    wave=dict(n_freq=128, middle_channels=128, attention_resolutions=[128, 256, 512], num_res_blocks=2, num_heads=8,
              num_groups=32, dropout=0.0, use_checkpoint=True, channel_mult=[1, 1, 1, 1, 2, 2, 2, 4, 4, 4]),
    sr=22050, n_fft=512, n_mels=128, max_audio_frame=32768, z_length=512)

FEATURE_YAML = os.path.join(ROOT, "tests", "golden", "mania_beatmap_features.yaml")


def model_config():
    return dict(target="mug.diffusion.diffusion.DDPM", params=dict(
        linear_start=0.0001, linear_end=0.02, log_every_t=100, timesteps=1000, z_channels=16, z_length=512,
        parameterization="eps", loss_type="smooth_l1", monitor="val/loss_simple",
        unet_config=dict(target="mug.diffusion.unet.UNetModel", params=SHIPPED["unet"]),
        first_stage_config=dict(target="mug.firststage.autoencoder.AutoencoderKL",
                                params=dict(monitor="val/loss", kl_weight=1e-6, ddconfig=SHIPPED["vae"],
                                            lossconfig=dict(target="torch.nn.Identity"))),
        cond_stage_config=dict(target="mug.cond.feature.BeatmapFeatureEmbedder",
                               params=dict(path_to_yaml=FEATURE_YAML, embed_dim=128)),
        wave_stage_config=dict(target="mug.cond.wave.MelspectrogramScaleEncoder1D", params=SHIPPED["wave"])))


_T0 = time.perf_counter()


def note(msg):
    """progress line on stderr (the JSON line on stdout stays alone)"""
    print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def host_threads():
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return os.cpu_count() or 1


def pin_near_gpu(dev):
    """Best effort: restrict this rank's threads (the enqueue thread above all) to the CPUs of the NUMA node its GPU hangs off, so that eight
    ranks sharing a host do not migrate across sockets between launches.  Returns a description for the JSON line (None: nothing to do)."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(dev)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            cpus = set()
            for part in f.read().strip().split(","):
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= set(os.sched_getaffinity(0))
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return {"pci": bdf, "numa_node": node, "cpus": len(cpus)}
    except Exception:                                  # noqa: BLE001 -- no sysfs / no such property: leave the affinity alone
        return None


def synth_audio(seconds, sr, seed):
    """0.5 sin(2pi 440 t) + 0.25 chirp(110 -> 3110 Hz) + N(0, 0.01)."""
    n = np.arange(int(round(seconds * sr)), dtype=np.float64)
    N = len(n)
    y = 0.5 * np.sin(2 * np.pi * 440.0 * n / sr) + 0.25 * np.sin(2 * np.pi * (110.0 + n / N * 3000.0) * n / sr)
    return (y + np.random.default_rng(seed).normal(0.0, 0.01, N)).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--seconds", type=float, default=180.0)
    ap.add_argument("--audio-sr", type=int, default=44100,
                    help="sample rate of the synthetic PCM (BASELINE configs[1]: 44.1 kHz); converted to the model's 22.05 kHz on the "
                         "device inside every timed step.  22050 skips the conversion")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--cfg-scale", type=float, default=1.0, help="1.0 = no guidance (scripts/mapping.py default); webui default is 5.0")
    ap.add_argument("--audios-per-rank", type=int, default=1, help="audios per rank and step (each with --batch seeds): 2 with --pack-songs 2 = the throughput mode; "
                                                                    "--gpus 8 --audios-per-rank 2 = BASELINE configs[2] (64 units)")
    ap.add_argument("--pack-songs", type=int, default=1, help="2: two audios of equal length share one batch-2B U-Net launch (mug/job.py)")
    ap.add_argument("--no-training-step", action="store_true", help="skip the extra training-step measurement (configs[4] shape, every N)")
    ap.add_argument("--train-batch", type=int, default=32, help="per-GPU batch of the training-step measurement (configs[4]: 256 over 8 GPUs)")
    ap.add_argument("--no-throughput-mode", action="store_true", help="skip the extra 2-songs-per-launch measurement (N = 1 only)")
    ap.add_argument("--weights", choices=["f32", "bf16"], default="f32", help="bf16: the reduced-precision mode (packed conv / linear weights in bfloat16; "
                                                                                "NOT the reference's arithmetic) as the timed configuration")
    ap.add_argument("--no-reduced-mode", action="store_true", help="skip the extra bf16-weight measurement (N = 1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the CPU baseline (torch scales badly past ~16 on these small tensors)")
    ap.add_argument("--cpu-timeout", type=int, default=150)
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--launcher-selftest", action="store_true", help=argparse.SUPPRESS)        # tests/test_bench_launcher.py: the rank plumbing on gloo, no GPU
    ap.add_argument("--z", type=int, default=512, help=argparse.SUPPRESS)
    ap.add_argument("--n-unet-steps", type=int, default=50, help=argparse.SUPPRESS)
    a = ap.parse_args()
    faulthandler.dump_traceback_later(180, repeat=True, file=sys.stderr)      # a hung stage shows up in the log
    if a.cpu_baseline_worker:
        return cpu_baseline_worker(a)

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(a.gpus)                       # plain `python bench.py --gpus N`: start the N ranks ourselves
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (use --nproc-per-node equal to --gpus)" % (a.gpus, world))
    if a.launcher_selftest:
        return launcher_selftest(a, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libmugd has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    affinity = pin_near_gpu(dev)
    import torch.distributed as dist
    grouped = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)      # under torch.distributed.run also at N = 1: the same
    if grouped:                                                                          # RCCL code path (init, barrier, gather) then runs on one GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)          # RCCL over xGMI; only barriers / final gather use it

    from mug._native import get_lib
    from mug.diffusion.ddim import DDIMSampler
    from mug.model.paramtree import seed_all_parameters
    from mug.util import instantiate_from_config, feature_dict_to_embedding_ids
    import yaml

    lib = get_lib()
    h3 = "f16x3" in lib.version()            # the library's conv arithmetic (csrc/conv_body.h: MUGD_CONV_H3)
    if a.weights == "bf16":
        lib.set_weight_precision(True)
    note("library loaded")
    model = instantiate_from_config(model_config()).eval()
    unet = model.model.unet_model
    z_cfg = SHIPPED["z_length"]
    seed_all_parameters(model, seed=0, s4_length_of=lambda k: unet.s4_length_of(k[len("model.unet_model."):], z_cfg))
    model = model.to(dev)
    sampler = DDIMSampler(model)
    note("model instantiated, seeded and moved to %s" % dev)

    from mug import job, shard
    B, S, sr, hop = a.batch, a.ddim_steps, SHIPPED["sr"], SHIPPED["n_fft"] // 4
    with open(FEATURE_YAML) as f:
        fy = yaml.safe_load(f)
    prompts = [{"sr": 4.0, "rank_status": "ranked"}, {"sr": 2.5, "ln_ratio": 0.4}, {"sr": 6.0, "ln": 1}, {"sr": 3.2}]
    # The work list (BASELINE configs[2] shape): `units_per_rank` (audio, seed) units per rank, B seeds per audio, partitioned
    # by mug/shard.py so that the seeds of one audio stay on one rank.  Default = one audio x B seeds per rank and step
    # (configs[1] on every GPU: weak scaling).  Every rank's PCM is resident in HBM before the timed region.
    n_audio = world * a.audios_per_rank
    units = job.make_units(n_audio, B, prompts=prompts, seed0=1000)
    mine = shard.partition(len(units), world, rank)
    my_audios = sorted(set(units[u]["audio"] for u in mine))
    pcm_in = {au: torch.from_numpy(synth_audio(a.seconds, a.audio_sr, seed=au)).to(dev) for au in my_audios}

    def mel_of(au):
        pcm = lib.resample_poly(pcm_in[au], sr, a.audio_sr) if a.audio_sr != sr else pcm_in[au]    # 44.1 kHz -> 22.05 kHz, polyphase FIR
        return lib.log_mel(pcm, sr=sr, n_fft=SHIPPED["n_fft"], hop=hop, n_mels=SHIPPED["n_mels"])     # (128, frames), fp16-rounded

    z = job.z_length_for(1 + int(round(a.seconds * sr)) // hop, SHIPPED["max_audio_frame"], z_cfg)   # webui.py:349-356
    ddim_ms, ev = [], [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]

    class _Timed:        # the DDIM loop alone (sampler.sample) for the U-Net steps/s figure: events on torch's stream around it
        def __init__(self, inner):
            self.inner = inner
            self.ddim_timesteps = None

        def sample(self, **kw):
            ev[0].record()
            r = self.inner.sample(**kw)
            ev[1].record()
            self.ddim_timesteps = self.inner.ddim_timesteps
            return r

    timed = _Timed(sampler)

    def one_step(pack=1):
        grids, stats = job.run_job(model, timed, units, mel_of, fy, steps=S, scale=a.cfg_scale, eta=0.0, batch=B, pack_songs=pack,
                                   max_audio_frame=SHIPPED["max_audio_frame"], z_length_cfg=z_cfg, gather=False)
        return grids, stats

    def sync_all():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        one_step(a.pack_songs)
        torch.cuda.synchronize()
        note("warmup step %d done" % i)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        grids, stats = one_step(a.pack_songs)
    if grouped:                         # the job's only data movement between ranks: note grids (8 x T bits per chart) to every rank
        all_grids = shard.gather_grids(torch.stack(grids), len(units), device=dev)
        assert all_grids.shape[0] == len(units)
    torch.cuda.synchronize()
    ddim_ms.append(ev[0].elapsed_time(ev[1]))
    sync_all()
    elapsed = time.perf_counter() - t0
    note("timed region done: %.1f ms per step (%d launches per rank and step)" % (elapsed / a.steps * 1e3, stats["launches"]))
    notes = int(sum(int(g.sum()) for g in grids))
    if grouped:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    Bl = B * (a.pack_songs if a.audios_per_rank >= a.pack_songs else 1)      # charts per U-Net launch

    train_info = train_info32 = None
    if not a.no_training_step and a.weights == "f32" and a.pack_songs == 1 and a.audios_per_rank == 1:
        try:                                                  # an extra: its failure must not take the headline line with it
            train_info = training_leg(a, lib, model, dev, world, rank, grouped, sync_all, bf16=True)
            note("training step (bf16 GEMMs) done: %.1f samples/s" % train_info["value"])
            train_info32 = training_leg(a, lib, model, dev, world, rank, grouped, sync_all, bf16=False)
            note("training step (fp32) done: %.1f samples/s" % train_info32["value"])
        except Exception as e:                                # noqa: BLE001
            if train_info is None:
                train_info = {"error": repr(e)[:400]}
            else:
                train_info32 = {"error": repr(e)[:400]}
            note("training step failed: %r" % e)

    out = None
    if rank == 0:
        n_unet_steps = len(sampler.ddim_timesteps)
        charts = len(units) * a.steps
        out = {
            "metric": "charts_per_sec (3-min audio, %d DDIM steps, batch %d)" % (S, B),
            "value": charts / elapsed, "unit": "charts/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32" if a.weights == "f32" else "bf16 weights, f32 activations / MFMA inputs after widening / accumulation (reduced-precision mode)"),
            "dtype_detail": ("fp32 tensors, fp32 accumulation; conv / linear products on the f16 matrix cores with BOTH operands split into f16 hi + 2^11-scaled lo "
                             "halves (3 MFMAs per block): fp32-equivalent -- measured error vs float64 2.7e-7 relative at K = 1024 against 7.7e-7 for the "
                             "fp32-input MFMA chain (profiles/r4_h3_probe.txt).  Domain: the whole fp32 range -- both operands are block-scaled by exact powers "
                             "of two (weights per packed set, activations per wave and 16-channel chunk, following the data; include/mugd.h: mugd_version), "
                             "tests/test_ops.py holds conv / norm+conv / log-mel to the unit-scale fp32 tolerance at operand scales 1e-30 .. 1e30; every "
                             "parity test runs at the fp32 tolerances" if h3 else
                             "fp32 tensors, fp32-input MFMA (v_mfma_f32_32x32x2_f32), fp32 accumulation") if a.weights == "f32" else None,
            "library": lib.version(),
            "shader_clock_mhz_under_matrix_load": lib.dev_clock_probe(),      # boxes of the pool differ by up to 25 % in what they sustain
            "data": "synthetic",
            "config": {"workload": "configs[1]: %.0f s synthetic %.2f kHz audio -> %s z=%d, %d DDIM steps, batch %d, cfg_scale %g, "
                                   "mel + wave-encode (once per song, shared by the seeds) + DDIM + VAE decode + note grid"
                                   % (a.seconds, a.audio_sr / 1e3, "22.05 kHz (device polyphase resampler) ->" if a.audio_sr != sr else "",
                                      z, n_unet_steps, B, a.cfg_scale),
                       "parallelism": "dp%d: %d (audio, seed) units = %d audio(s) x %d seeds per rank, partitioned by mug/shard.py through the job driver "
                                      "(mug/job.py); no data-path collective, one bit-packed all_gather of the note grids at the end" % (world, len(units), a.audios_per_rank, B),
                       "songs_per_launch": a.pack_songs if a.audios_per_rank >= a.pack_songs else 1,
                       "weights": "seeded synthetic, shipped architecture (151 M params)"},
            "unet_sample_steps_per_s": Bl * n_unet_steps * (2 if a.cfg_scale != 1.0 else 1) / (ddim_ms[-1] * 1e-3) * world,
            "ddim_loop_ms": ddim_ms[-1], "notes_in_last_batch": notes,
        }
        if not a.no_roofline:
            # the U-Net program compiled for the timed run, replayed once eagerly with a HIP event pair per launch
            prof = unet.native().profile()
            k = {f: prof["conv_gemm"][f] + prof["conv_gemm_gated"][f] for f in ("ms", "flops", "launches")}
            # An event pair around a launch also times the gap to the previous kernel.  Calibrate it from two live measurements
            # of the same program: (sum of event-bracketed launch times, eager) - (time of one step of the timed
            # region), spread over the launches; rocprofv3's per-kernel averages (profiles/) agree with the corrected figure.
            eager_ms = sum(v["ms"] for v in prof.values())
            launches = sum(v["launches"] for v in prof.values())
            graph_ms = ddim_ms[-1] / n_unet_steps
            gap_ms = max(0.0, (eager_ms - graph_ms) / max(launches, 1))
            conv_ms = k["ms"] - k["launches"] * gap_ms
            achieved = k["flops"] / (conv_ms * 1e-3) / 1e12
            # HBM-side bytes per conv_gemm launch cannot be measured from inside this process: they come from separate rocprofv3
            # --pmc FETCH_SIZE / WRITE_SIZE passes over tests/gpu_unet_once.py (same program, same box class), whose summary is
            # committed as profiles/conv_traffic.json -- reported with its source, not as a live figure
            traffic, traffic_source = None, None
            tpath = os.path.join(ROOT, "profiles", "conv_traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    tj = json.load(f)
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = "profiles/conv_traffic.json (%s): separate rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes, FETCH_SIZE doubled " \
                                 "per MI355X_MICROARCH.md; not measured by this run" % tj.get("round", "round 1 tree")
            # host side of a step: wall clock around back-to-back enqueues of the same program (no events, no synchronisation inside)
            host_us, host_ops = unet.native().host_enqueue(5)
            out["host_enqueue"] = {"us_per_unet_eval": host_us, "launches_per_unet_eval": host_ops, "us_per_launch": host_us / max(host_ops, 1),
                                   "gpu_us_per_unet_eval": graph_ms * 1e3, "host_over_gpu": host_us / max(graph_ms * 1e3, 1e-9),
                                   "thread_affinity": affinity,
                                   "what": "mugd_net_host_enqueue: one thread enqueueing the U-Net program of the timed run 5 times back to back from an idle "
                                           "stream; conv launches are prepared once per compiled program (validation, kernel form, K-slices), a step only "
                                           "calls hipLaunchKernel"}
            peak = PEAK_H3_TFLOPS if h3 else PEAK_FP32_MFMA_TFLOPS
            out["roofline"] = {"kernel": "conv_gemm_kernel<..., TN = 32 | 16> (implicit-GEMM conv1d / linear, 32x32 and 32x16 tiles of one template, %s; "
                                         "all %d launches of one U-Net evaluation, HIP events around every launch on the library stream, "
                                         "minus the calibrated inter-launch gap)" % ("f16x3-split MFMA" if h3 else "fp32-input MFMA", k["launches"]),
                               "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                               "peak_detail": ("algorithmic 2MKN flops against the dense f16 MFMA peak / 3 (three MFMAs per fp32-equivalent product block); "
                                               "these launches are small dependent GEMMs bound by their per-tile latency chain and the VALU work of the operand "
                                               "transform / split, not by the matrix pipe (DESIGN.md 4)") if h3 else "fp32-input MFMA peak = vector fp32 peak",
                               "frac": achieved / peak, "frac_of_fp32_input_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
                               "traffic": traffic, "traffic_source": traffic_source,
                               "launches_per_unet_eval": k["launches"], "avg_launch_us": conv_ms * 1e3 / max(k["launches"], 1),
                               "avg_launch_us_event_bracketed": k["ms"] * 1e3 / max(k["launches"], 1),
                               "event_gap_us_per_launch": gap_ms * 1e3,
                               "achieved_event_bracketed": k["flops"] / (k["ms"] * 1e-3) / 1e12,
                               "algorithmic_gflop_per_launch": k["flops"] / 1e9 / max(k["launches"], 1),
                               "by_kernel_ms": {n: round(v["ms"], 4) for n, v in prof.items()}}
        note("roofline probe done")
        if world == 1 and not a.no_throughput_mode and a.pack_songs == 1 and a.audios_per_rank == 1:
            # throughput mode, reported NEXT TO the batch-4 headline: two songs x B seeds share one batch-2B launch (the DDIM loop is
            # launch-latency bound at batch 4).  Same pipeline through the same job driver, 2 timed passes.
            for npack, key in ((2, "throughput_mode"), (4, "throughput_mode_4_songs")):
                units2 = job.make_units(npack, B, prompts=prompts, seed0=1000)
                for au in range(1, npack):
                    if au not in pcm_in:
                        pcm_in[au] = torch.from_numpy(synth_audio(a.seconds, a.audio_sr, seed=au)).to(dev)

                def packed():
                    return job.run_job(model, timed, units2, mel_of, fy, steps=S, scale=a.cfg_scale, eta=0.0, batch=B, pack_songs=npack,
                                       max_audio_frame=SHIPPED["max_audio_frame"], z_length_cfg=z_cfg, gather=False)
                packed()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                reps = 2
                for _ in range(reps):
                    g2, st2 = packed()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / reps
                out[key] = {"what": "%d songs x %d seeds per batch-%d U-Net launch (mug/job.py pack_songs=%d), same pipeline" % (npack, B, npack * B, npack),
                            "value": npack * B / dt, "unit": "charts/s", "ms_per_launch_of_%d_charts" % (npack * B): dt * 1e3,
                            "ddim_loop_ms": ev[0].elapsed_time(ev[1]),
                            "unet_sample_steps_per_s": npack * B * n_unet_steps / (ev[0].elapsed_time(ev[1]) * 1e-3), "launches": st2["launches"]}
                note("throughput mode (%d songs per launch) done: %.1f charts/s" % (npack, npack * B / dt))
        if world == 1 and not a.no_throughput_mode and a.pack_songs == 1 and a.audios_per_rank == 1 and a.cfg_scale == 1.0:
            # the web UI's default click (webui.py:336-390: cfg scale 5 with an unconditional batch): U-Net batch 2 B, same pipeline, reported next to
            # the no-guidance headline (BASELINE configs[1] is quoted without guidance)
            def guided():
                return job.run_job(model, timed, units, mel_of, fy, steps=S, scale=5.0, eta=0.0, batch=B, pack_songs=1,
                                   max_audio_frame=SHIPPED["max_audio_frame"], z_length_cfg=z_cfg, gather=False)
            guided()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = 2
            for _ in range(reps):
                guided()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / reps
            out["cfg_scale_5"] = {"what": "the same step with classifier-free guidance (unconditional_guidance_scale = 5.0, webui.py's default): %d seeds -> U-Net batch %d" % (B, 2 * B),
                                  "value": len(units) / dt, "unit": "charts/s", "ms_per_step": dt * 1e3, "ddim_loop_ms": ev[0].elapsed_time(ev[1]),
                                  "unet_sample_steps_per_s": 2 * B * n_unet_steps / (ev[0].elapsed_time(ev[1]) * 1e-3)}
            note("cfg scale 5 leg done: %.1f charts/s" % (len(units) / dt))
        if world == 1 and not a.no_reduced_mode and a.weights == "f32" and a.pack_songs == 1 and a.audios_per_rank == 1:
            # reduced-precision mode, reported separately (never the headline: the reference is fp32): bf16 weight storage, fp32 everything
            # else; same pipeline, networks recompiled with re-packed weights
            lib.set_weight_precision(True)
            for m_ in model.modules():
                if hasattr(m_, "_fp"):
                    m_._fp = None                           # forces NativeModule.native() to hand the parameters over again (re-pack)
            one_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            reps = 2
            for _ in range(reps):
                one_step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / reps
            out["reduced_precision_mode"] = {"what": "conv / linear weights packed as bfloat16 (mugd_set_weight_precision), activations / accumulation fp32; "
                                                     "parity bound: tests/test_nets.py::test_reduced_precision_mode_bf16_weights",
                                             "dtype": "bf16 weights + f32", "value": len(units) / dt, "unit": "charts/s", "ms_per_step": dt * 1e3,
                                             "ddim_loop_ms": ev[0].elapsed_time(ev[1]),
                                             "unet_sample_steps_per_s": Bl * n_unet_steps / (ev[0].elapsed_time(ev[1]) * 1e-3)}
            lib.set_weight_precision(False)
            note("reduced-precision mode done: %.1f charts/s" % (len(units) / dt))
        if train_info is not None:
            out["training_step"] = train_info
        if train_info32 is not None:
            out["training_step_fp32"] = train_info32
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a, z, n_unet_steps)
            note("cpu baseline done")
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-run this very command line under torch.distributed.run, one rank per GPU
    of this node (rendezvous on 127.0.0.1, a free port).  The children see WORLD_SIZE and take the normal path; rank 0's JSON
    line goes to our stdout unchanged; our exit status is the launcher's."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_threads() // n)))
    note("starting %d ranks: %s" % (n, " ".join(cmd[1:9])))
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit(rc)


def launcher_selftest(a, rank, world):
    """The rank plumbing of main() -- process group, barrier, K 'steps' bracketed like the timed region, MAX over ranks, one gather,
    ONE JSON line from rank 0 -- on the gloo backend with a stub body, so the launcher path is testable without GPUs."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    dist.barrier()
    t0 = time.perf_counter()
    acc = torch.zeros(4)
    for i in range(a.steps):
        acc += torch.arange(4, dtype=torch.float32) * (rank + 1)
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    got = [torch.zeros(4) for _ in range(world)]
    dist.all_gather(got, acc)
    if rank == 0:
        print(json.dumps({"metric": "launcher_selftest", "n_gpus": world, "steps": a.steps, "elapsed_max_s": float(tt.item()),
                          "gathered": [g.tolist() for g in got]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


PEAK_BF16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16), 256 CUs @ 2.4 GHz
TRAIN_GFLOP_PER_SAMPLE = 3 * (22.37 + 50.5)      # forward + data gradients + weight gradients of the U-Net (22.37) and wave encoder (50.5), BASELINE.md section 2


def training_leg(a, lib, model, dev, world, rank, grouped, sync_all, bf16=True):
    """BASELINE configs[4] shape: one DDPM training step per rank on `--train-batch` synthetic samples (z = 512 latents, 32768-frame
    log-mel, random prompts and timesteps) = q_sample -> wave encoder -> prompt embedding -> U-Net -> smooth-L1 loss -> backward
    through all three networks (mug/train.py TrainPlan: native block forward / backward entry points, blocks keep their forward
    intermediates, no host synchronisation inside the step) -> bucketed all-reduce of the 1327 gradient tensors overlapped with the
    backward sweep (RCCL when N > 1) -> AdamW on every tensor (one launch).  bf16: the GEMMs on the bf16 matrix cores with fp32
    accumulation (configs[4]'s precision; everything else fp32); else the fp32 parity mode (forward / data-gradient GEMMs through conv_gemm's
    split-f16 arithmetic, weight gradients on the fp32-input MFMA).  2 warm-up + 3 timed steps,
    barrier + synchronize on both sides, MAX over ranks; then one more step with an event pair around every GEMM launch for the
    GEMM roofline.  Reported next to the headline, never as it."""
    import torch.distributed as dist
    from mug import train
    lib.train_set_precision(bf16)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    Bt, z = a.train_batch, SHIPPED["z_length"]
    g = torch.Generator().manual_seed(1234 + rank)
    x0, noise = torch.randn(Bt, 16, z, generator=g), torch.randn(Bt, 16, z, generator=g)
    t = torch.randint(0, 1000, (Bt,), generator=g)
    ids = torch.randint(0, sd["model.cond_stage_model.embedding.weight"].shape[0], (Bt, 21), generator=g)
    mel = torch.randn(Bt, SHIPPED["n_mels"], SHIPPED["max_audio_frame"], generator=g).abs()
    x0, noise, t, ids, mel = (v.to(dev) for v in (x0, noise, t, ids, mel))
    plan, opt = train.TrainPlan(lib, sd, SHIPPED["unet"], SHIPPED["wave"]), [None]

    def step(i):
        red = train.BucketedAllReduce(bucket_bytes=64 << 20, even_single=True) if grouped else None          # 64 MB buckets, reduced while the backward sweep runs
        loss, grads = plan.step(x0, noise, t, ids, mel, reducer=red)
        if opt[0] is None:
            opt[0] = train.AdamW(lib, {k: sd[k] for k in grads}, grads, lr=1e-6)           # the model's own tensors: updated in place
        opt[0].step(grads=grads)
        return loss, grads

    try:
        loss, grads = step(0)                        # first sight: weights packed one by one, scratch pool filled
        loss, grads = step(1)                        # steady state from here: one pack-table launch, one reduction-table launch per step
        sync_all()
        t0 = time.perf_counter()
        reps = 3
        for i in range(reps):
            loss, grads = step(2 + i)
        sync_all()
        dt = (time.perf_counter() - t0) / reps
        if grouped:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        lib.train_profile(True)                      # one more step with HIP events around every GEMM launch (library stream)
        step(2 + reps)
        prof = lib.train_profile(False)
    finally:
        lib.train_set_precision(False)
    nparam = sum(v.numel() for v in grads.values())
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
    gemm_ms = prof["conv"]["ms"] + prof["wgrad"]["ms"]
    gemm_fl = prof["conv"]["flops"] + prof["wgrad"]["flops"]
    kern = "tconv_bf16_kernel / twgrad_bf16_kernel (v_mfma_f32_32x32x16_bf16)" if bf16 else \
        "conv_gemm_kernel (forward, data gradients: 3 x v_mfma_f32_32x32x16_f16 per block on split operands) / wgrad_mfma_kernel (v_mfma_f32_32x32x2_f32)"
    gemm = {"kernel": kern + ": all %d GEMM launches of one step, HIP event pair around each on the library stream (bf16 mode: packed weights "
                             "come from the step bracket's cache and the split-K slices are summed by the step's one reduction launch -- both "
                             "outside the brackets)" % (prof["conv"]["launches"] + prof["wgrad"]["launches"]),
            "bound": "mfma", "achieved": gemm_fl / (gemm_ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
            "frac": gemm_fl / (gemm_ms * 1e-3) / 1e12 / peak,
            "gemm_ms_per_step": gemm_ms, "gemm_share_of_step": gemm_ms / (dt * 1e3),
            "forward_and_dgrad": {"tflops": prof["conv"]["flops"] / (prof["conv"]["ms"] * 1e-3) / 1e12, "ms": prof["conv"]["ms"], "launches": prof["conv"]["launches"]},
            "wgrad": {"tflops": prof["wgrad"]["flops"] / (prof["wgrad"]["ms"] * 1e-3) / 1e12, "ms": prof["wgrad"]["ms"], "launches": prof["wgrad"]["launches"]}}
    # What bounds the step is the memory system, not the matrix pipe: fp32 activations / gradients of 0.5 GB per tensor at the
    # wave encoder's first level stream through every pass.  `traffic` is NOT measured by this process: counter bytes per step from
    # profiles/train_traffic.json (separate rocprofv3 --pmc passes), scaled by the batch; `achieved` = those bytes over this run's step time.
    roof = dict(gemm, bound="mfma (no counter file: GEMM-only figure)", traffic=None)
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "train_traffic.json")) as f:
            tr = json.load(f)
        if bf16 and tr.get("mode") == "bf16":
            per_step = (tr["fetch_gb_x2"] + tr["write_gb"]) * 1e9 * Bt / tr["batch"]
            roof = {"kernel": "the whole step's launch list (tconv / twgrad / GroupNorm forward + backward are 85 % of its bytes)",
                    "bound": "hbm", "achieved": per_step / dt / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": per_step / dt / 1e9 / PEAK_HBM_GBS,
                    "traffic": per_step, "traffic_source": "profiles/train_traffic.json: %s" % tr["source"],
                    "traffic_range_gb": [(tr["fetch_gb_raw"] + tr["write_gb"]) * Bt / tr["batch"], (tr["fetch_gb_x2"] + tr["write_gb"]) * Bt / tr["batch"]],
                    # `frac` above uses the UPPER end of the range (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes); the lower end (raw counter) next to it
                    "frac_range": [(tr["fetch_gb_raw"] + tr["write_gb"]) * 1e9 * Bt / tr["batch"] / dt / 1e9 / PEAK_HBM_GBS, per_step / dt / 1e9 / PEAK_HBM_GBS],
                    "gemm_mfma": gemm}
    except (OSError, ValueError, KeyError):
        pass
    return {"what": "configs[4] shape: DDPM training step (q_sample, wave encoder, prompt embedding, U-Net, smooth-L1, backward through all three "
                    "networks, bucketed gradient all-reduce overlapped with the backward sweep, AdamW), per-GPU batch %d, z = %d, synthetic data; %s"
                    % (Bt, z, "conv / Linear GEMMs (forward, data and weight gradients) with bf16 MFMA inputs and fp32 accumulation, fp32 master weights, "
                              "activations, norms, softmax, S4 and reductions" if bf16 else "the fp32 parity mode: forward / data-gradient GEMMs on split-f16 operands (fp32-equivalent), weight gradients on the fp32-input MFMA"),
            "value": Bt * world / dt, "unit": "samples/s", "ms_per_step": dt * 1e3, "global_batch": Bt * world, "dtype": "bf16" if bf16 else "f32",
            "loss": float(loss), "gradient_tensors": len(grads), "trainable_parameters": int(nparam),
            "algorithmic_tflops_whole_step": TRAIN_GFLOP_PER_SAMPLE * Bt / dt / 1e3,
            "roofline": roof,
            "allreduce": ("RCCL, %.0f MB of fp32 gradients per step in 64 MB buckets, asynchronous, overlapped with the backward sweep"
                          % (nparam * 4 / 1e6)) if world > 1 else "none (1 rank)"}


def cpu_baseline(a, z, n_steps):
    """Runs `bench.py --cpu-baseline-worker` in a fresh process: OMP/torch thread count is fixed BEFORE torch
    creates its pools (a 256-thread pool makes the small-tensor oracle 100x slower), under a hard timeout."""
    import subprocess
    nthr = max(1, min(host_threads(), a.cpu_threads))
    env = dict(os.environ, OMP_NUM_THREADS=str(nthr), MKL_NUM_THREADS=str(nthr), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--batch", str(a.batch), "--seconds", str(a.seconds),
           "--audio-sr", str(a.audio_sr), "--cpu-threads", str(nthr), "--z", str(z), "--n-unet-steps", str(n_steps)]
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=sys.stderr, text=True, timeout=a.cpu_timeout)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and line:
            return json.loads(line[-1])
        return {"value": None, "unit": "charts/s", "cores": nthr, "kind": "port", "sample": "worker failed (rc %d)" % r.returncode}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "charts/s", "cores": nthr, "kind": "port", "sample": "worker exceeded %d s" % a.cpu_timeout}


def cpu_baseline_worker(a):
    """The oracle (oracle/: PyTorch-CPU fp32 restatement of the reference path, bit-identical to the real
    reference on the golden fixtures) on this host's cores, same seeded weights, bounded sample."""
    from oracle import host, nets
    from mug.model.paramtree import seed_all_parameters
    from mug.util import instantiate_from_config
    nthr = a.cpu_threads
    torch.set_num_threads(nthr)
    z, n_steps, B = a.z, a.n_unet_steps, a.batch
    model = instantiate_from_config(model_config()).eval()
    unet = model.model.unet_model
    seed_all_parameters(model, seed=0, s4_length_of=lambda k: unet.s4_length_of(k[len("model.unet_model."):], SHIPPED["z_length"]))
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    note("cpu baseline: weights ready (%d threads)" % nthr)
    g = torch.Generator().manual_seed(0)
    t0 = time.perf_counter()
    y = synth_audio(a.seconds, a.audio_sr, 0)
    t0 = time.perf_counter()
    if a.audio_sr != SHIPPED["sr"]:
        y = host.resample_poly(y, SHIPPED["sr"], a.audio_sr)
    mel = host.pad_or_trunc_mel(host.log_mel(y).astype(np.float32), z * 64)
    t_mel = time.perf_counter() - t0
    note("cpu baseline: resample + mel %.2fs" % t_mel)
    with torch.no_grad():
        t0 = time.perf_counter()
        w = nets.wave_encode(sd, nets.WAVE_DEFAULT, torch.from_numpy(mel)[None])
        t_wave = time.perf_counter() - t0
        note("cpu baseline: wave encoder %.2fs" % t_wave)
        w = [m.repeat(B, 1, 1) for m in w]
        x = torch.randn((B, 16, z), generator=g)
        c = torch.randn((B, 128, 21), generator=g)
        kc = {}
        nets.unet_forward(sd, nets.UNET_DEFAULT, x, torch.full((B,), 981), c, w, kernel_cache=kc)     # warm-up (bakes the S4 kernels)
        reps = 3
        t0 = time.perf_counter()
        for i in range(reps):
            nets.unet_forward(sd, nets.UNET_DEFAULT, x, torch.full((B,), 981 - 20 * i), c, w, kernel_cache=kc)
        t_unet = (time.perf_counter() - t0) / reps
        note("cpu baseline: U-Net eval %.3fs" % t_unet)
        t0 = time.perf_counter()
        nets.vae_decode(sd, nets.VAE_DEFAULT, x)
        t_dec = time.perf_counter() - t0
    per_batch = t_mel + t_wave + n_steps * t_unet + t_dec
    print(json.dumps({
        "value": B / per_batch, "unit": "charts/s", "cores": nthr, "kind": "port",
        "kind_detail": "oracle/ (PyTorch-CPU fp32 restatement, bit-identical to the reference on the golden fixtures); the GPU box has no "
                       "/root/reference.  Conservative: the S4 kernels are cached here while the reference regenerates them in every U-Net "
                       "call (s4.py:706-832); the unmodified reference timed in the authoring container: profiles/r2_reference_cpu_timing.json",
        "sample": "1 resample+mel (%.2fs) + 1 wave-encode B=1 (%.2fs) + %d U-Net evals B=%d z=%d (%.3fs each, S4 kernels cached) + 1 decode (%.2fs), "
                  "extrapolated to %d steps; PyTorch-CPU fp32, %d threads (host exposes %d)"
                  % (t_mel, t_wave, reps, B, z, t_unet, t_dec, n_steps, nthr, host_threads()),
        "unet_sample_steps_per_s": B / t_unet}), flush=True)


if __name__ == "__main__":
    main()
