#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): charts/s for 3-minute audio, 50 DDIM steps, batch 4, on N MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One STEP = one batch of `--batch` charts for one synthetic 3-minute 22.05 kHz song, i.e. one full pass of
the hot path with the PCM already resident in HBM:
    log-mel (HIP FFT + mel GEMM) -> wave encoder (once per song, shared by the seeds) -> prompt embedding
    -> 50-step DDIM loop over the U-Net (hipGraph replay) -> VAE decode -> thresholded 4K note grid.
Every rank works on its own (song, seeds) units; there is no data-path collective (weak scaling).
Weights are seeded synthetic values of the shipped architecture (no checkpoint exists offline), with the
reference's zero-initialised tensors randomised so no branch is a no-op.  Arithmetic is fp32 end to end
(fp32-input MFMA), like the reference.

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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the CPU baseline (torch scales badly past ~16 on these small tensors)")
    ap.add_argument("--cpu-timeout", type=int, default=150)
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--z", type=int, default=512, help=argparse.SUPPRESS)
    ap.add_argument("--n-unet-steps", type=int, default=50, help=argparse.SUPPRESS)
    a = ap.parse_args()
    faulthandler.dump_traceback_later(180, repeat=True, file=sys.stderr)      # a hung stage shows up in the log
    if a.cpu_baseline_worker:
        return cpu_baseline_worker(a)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == a.gpus, "launch with --nproc-per-node equal to --gpus"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libmugd has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)          # RCCL over xGMI; only barriers / final gather use it

    from mug._native import get_lib
    from mug.diffusion.ddim import DDIMSampler
    from mug.model.paramtree import seed_all_parameters
    from mug.util import instantiate_from_config, feature_dict_to_embedding_ids
    import yaml

    lib = get_lib()
    note("library loaded")
    model = instantiate_from_config(model_config()).eval()
    unet = model.model.unet_model
    z_cfg = SHIPPED["z_length"]
    seed_all_parameters(model, seed=0, s4_length_of=lambda k: unet.s4_length_of(k[len("model.unet_model."):], z_cfg))
    model = model.to(dev)
    sampler = DDIMSampler(model)
    note("model instantiated, seeded and moved to %s" % dev)

    B, S, sr, hop = a.batch, a.ddim_steps, SHIPPED["sr"], SHIPPED["n_fft"] // 4
    pcm_in = torch.from_numpy(synth_audio(a.seconds, a.audio_sr, seed=rank)).to(dev)   # resident before the timed region
    pcm = lib.resample_poly(pcm_in, sr, a.audio_sr)                                   # (length rule only; redone in every step)
    with open(FEATURE_YAML) as f:
        fy = yaml.safe_load(f)
    prompts = [{"sr": 4.0, "rank_status": "ranked"}, {"sr": 2.5, "ln_ratio": 0.4}, {"sr": 6.0, "ln": 1}, {"sr": 3.2}]
    ids = torch.tensor([feature_dict_to_embedding_ids(prompts[i % 4], fy) for i in range(B)], dtype=torch.float32, device=dev)
    uc_ids = torch.tensor([feature_dict_to_embedding_ids({}, fy)] * B, dtype=torch.float32, device=dev)
    ratio = SHIPPED["max_audio_frame"] // z_cfg
    frames = 1 + pcm.numel() // hop
    z = (int(frames / ratio / 32) + 1) * 32                     # webui.py:349-356
    model.z_length = z
    gens = [torch.Generator(device="cpu").manual_seed(1000 * rank + i) for i in range(B)]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ddim_ms = []

    def one_step():
        pcm = lib.resample_poly(pcm_in, sr, a.audio_sr) if a.audio_sr != sr else pcm_in     # 44.1 kHz -> 22.05 kHz, polyphase FIR
        mel = lib.log_mel(pcm, sr=sr, n_fft=SHIPPED["n_fft"], hop=hop, n_mels=SHIPPED["n_mels"])   # (128, frames), fp16-rounded
        t = mel.shape[1]
        tgt = z * ratio
        mel = torch.nn.functional.pad(mel, (0, tgt - t)) if t < tgt else mel[:, :tgt]               # webui.py:358-367
        w = model.model.wave_model(mel[None])                    # once per song; the B seeds share the maps
        c = model.model.cond_stage_model(ids)
        uc = model.model.cond_stage_model(uc_ids) if a.cfg_scale != 1.0 else None
        x_T = torch.stack([torch.randn((16, z), generator=g) for g in gens]).to(dev)
        ev0.record()
        lat, _ = sampler.sample(S=S, c=c, w=w, batch_size=B, eta=0.0, verbose=False, x_T=x_T,
                                unconditional_guidance_scale=a.cfg_scale, unconditional_conditioning=uc,
                                tqdm_class=lambda *aa, **kk: None)
        ev1.record()
        logits = model.model.decode(lat)
        grid = torch.cat([logits[:, 0:4] > 0, logits[:, 8:12] > 0], dim=1)       # convertor.py:212-216
        return grid, (ev0, ev1)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        one_step()
        torch.cuda.synchronize()
        note("warmup step %d done" % i)
    sync_all()
    t0 = time.perf_counter()
    notes = 0
    for _ in range(a.steps):
        grid, _ = one_step()
    if world > 1:                       # the job's only data movement between ranks: note grids (8 x T bits per chart) to every rank
        from mug import shard
        all_grids = shard.gather_grids(grid, world * B, device=dev)
        assert all_grids.shape[0] == world * B
    torch.cuda.synchronize()
    ddim_ms.append(ev0.elapsed_time(ev1))
    sync_all()
    elapsed = time.perf_counter() - t0
    note("timed region done: %.1f ms per step" % (elapsed / a.steps * 1e3))
    notes = int(grid.sum().item())
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    out = None
    if rank == 0:
        n_unet_steps = len(sampler.ddim_timesteps)
        charts = world * B * a.steps
        out = {
            "metric": "charts_per_sec (3-min audio, %d DDIM steps, batch %d)" % (S, B),
            "value": charts / elapsed, "unit": "charts/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: %.0f s synthetic %.2f kHz audio -> %s z=%d, %d DDIM steps, batch %d, cfg_scale %g, "
                                   "mel + wave-encode (once per song, shared by the seeds) + DDIM + VAE decode + note grid"
                                   % (a.seconds, a.audio_sr / 1e3, "22.05 kHz (device polyphase resampler) ->" if a.audio_sr != sr else "",
                                      z, n_unet_steps, B, a.cfg_scale),
                       "parallelism": "dp%d (independent (audio, seed) units per rank, no data-path collective)" % world,
                       "weights": "seeded synthetic, shipped architecture (151 M params)"},
            "unet_sample_steps_per_s": B * n_unet_steps * (2 if a.cfg_scale != 1.0 else 1) / (ddim_ms[-1] * 1e-3) * world,
            "ddim_loop_ms": ddim_ms[-1], "notes_in_last_batch": notes,
        }
        if not a.no_roofline:
            # the U-Net program compiled for the timed run, replayed once eagerly with a HIP event pair per launch
            prof = unet.native().profile()
            k = {f: prof["conv_gemm"][f] + prof["conv_gemm_gated"][f] for f in ("ms", "flops", "launches")}
            # An event pair around a launch also times the gap to the previous kernel.  Calibrate it from two live measurements
            # of the same program: (sum of event-bracketed launch times, eager) - (time of one graph-replayed step of the timed
            # region), spread over the launches; rocprofv3's per-kernel averages (profiles/) agree with the corrected figure.
            eager_ms = sum(v["ms"] for v in prof.values())
            launches = sum(v["launches"] for v in prof.values())
            graph_ms = ddim_ms[-1] / n_unet_steps
            gap_ms = max(0.0, (eager_ms - graph_ms) / max(launches, 1))
            conv_ms = k["ms"] - k["launches"] * gap_ms
            achieved = k["flops"] / (conv_ms * 1e-3) / 1e12
            traffic = None                      # HBM bytes per conv_gemm launch from a separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass
            tpath = os.path.join(ROOT, "profiles", "conv_traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch")
            out["roofline"] = {"kernel": "conv_gemm_kernel / conv_gemm16_kernel (fp32-MFMA implicit-GEMM conv1d / linear, 32x32 and 32x16 tiles; "
                                         "all %d launches of one U-Net evaluation, HIP events around every launch on the library stream, "
                                         "minus the calibrated inter-launch gap)" % k["launches"],
                               "bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                               "launches_per_unet_eval": k["launches"], "avg_launch_us": conv_ms * 1e3 / max(k["launches"], 1),
                               "avg_launch_us_event_bracketed": k["ms"] * 1e3 / max(k["launches"], 1),
                               "event_gap_us_per_launch": gap_ms * 1e3,
                               "achieved_event_bracketed": k["flops"] / (k["ms"] * 1e-3) / 1e12,
                               "algorithmic_gflop_per_launch": k["flops"] / 1e9 / max(k["launches"], 1),
                               "by_kernel_ms": {n: round(v["ms"], 4) for n, v in prof.items()}}
        note("roofline probe done")
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a, z, n_unet_steps)
            note("cpu baseline done")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


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
        "sample": "1 resample+mel (%.2fs) + 1 wave-encode B=1 (%.2fs) + %d U-Net evals B=%d z=%d (%.3fs each, S4 kernels cached) + 1 decode (%.2fs), "
                  "extrapolated to %d steps; PyTorch-CPU fp32, %d threads (host exposes %d)"
                  % (t_mel, t_wave, reps, B, z, t_unet, t_dec, n_steps, nthr, host_threads()),
        "unet_sample_steps_per_s": B / t_unet}), flush=True)


if __name__ == "__main__":
    main()
