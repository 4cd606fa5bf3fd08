"""GPU probe (development tool, run through gpurun): times the headline workload pieces and prints
the per-kernel-class event profile.  python tests/gpu_probe.py [--z 512] [--B 4] [--S 50] [--quick]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mug-diffusion_amd"))

import torch  # noqa: E402

from oracle import cases, nets, sampler, weights  # noqa: E402
from mug._native import get_lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--z", type=int, default=512)
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--S", type=int, default=50)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--skip-wave", action="store_true")
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true", help="U-Net forward + per-kernel-class profile + graph DDIM loop only")
    ap.add_argument("--streams", type=int, default=0, help="experiment: split the batch over N concurrent streams (N contexts)")
    a = ap.parse_args()
    case = cases.FULL
    z, B, S = a.z, a.B, a.S
    res = {"z": z, "B": B, "S": S}
    t0 = time.time()
    lib = get_lib()
    man = weights.load_manifest(os.path.join(cases.GOLDEN, case["manifest"]))
    sd = weights.set_s4_lengths(weights.make_state_dict(man, 0), case["unet"], z)
    print("weights built %.1fs" % (time.time() - t0), flush=True)
    unet = lib.unet(case["unet"]); unet.set_params(sd, "model.unet_model.")
    vae = lib.vae(case["vae"]); vae.set_params(sd, "model.first_stage_model.")
    wave = lib.wave(case["wave"]); wave.set_params(sd, "model.wave_model.")
    dev = lib.device

    def timed(fn, reps=a.reps):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        return min(ts), sorted(ts)[len(ts) // 2]

    x = cases.x_T(1, B, z).to(dev)
    t = torch.full((B,), 501, dtype=torch.long, device=dev)
    c = cases.context(case, 1, B).to(dev)
    w = [m.to(dev) for m in cases.audio_maps(case, 1, 1, z)]
    # ---- single U-Net evaluation (eager launches)
    mn, md = timed(lambda: unet.forward(x, t, c, w), reps=5)
    res["unet_forward_ms"] = md * 1e3
    print("unet forward B=%d z=%d: min %.3f ms  median %.3f ms" % (B, z, mn * 1e3, md * 1e3), flush=True)
    prof = unet.profile()
    res["unet_profile"] = prof
    tot = sum(v["ms"] for v in prof.values())
    for k, v in prof.items():
        tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 and v["flops"] > 0 else 0.0
        print("  %-16s %5d launches  %8.3f ms  %5.1f%%  %7.2f GFLOP  %6.2f TFLOP/s" % (
            k, v["launches"], v["ms"], 100 * v["ms"] / max(tot, 1e-9), v["flops"] / 1e9, tf))
    print("  total (event-bracketed, eager): %.3f ms" % tot, flush=True)
    # ---- DDIM loop
    steps = sampler.ddim_step_scalars(sd["alphas_cumprod"].numpy(), S, 0.0)
    ts_ = [s["t"] for s in steps]
    sched = [[s["a_t"], s["a_prev"], s["sigma"], s["sqrt_1m_at"]] for s in steps]
    if a.streams > 1:
        from mug._native import Lib
        n = a.streams
        assert B % n == 0
        ctxs = []
        for i in range(n):
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                li = Lib()
                ui = li.unet(case["unet"]); ui.set_params(sd, "model.unet_model.")
            ctxs.append((st, li, ui))
        bs = B // n

        def run_all():
            for i, (st, li, ui) in enumerate(ctxs):
                with torch.cuda.stream(st):
                    ui.ddim_sample(x[i * bs:(i + 1) * bs], c[i * bs:(i + 1) * bs], w, ts_, sched)
        mn, md = timed(run_all, reps=3)
        res["ddim_streams%d_ms" % n] = md * 1e3
        print("ddim %d streams x B=%d: %.2f ms total, %.3f ms/step, %.0f sample-steps/s" % (n, bs, md * 1e3, md * 1e3 / len(ts_), B * len(ts_) / md), flush=True)
    for graph in ((True,) if a.quick else (True, False)):
        lib.set_graph_mode(graph)
        mn, md = timed(lambda: unet.ddim_sample(x, c, w, ts_, sched), reps=3)
        key = "ddim_graph" if graph else "ddim_eager"
        res[key + "_ms"] = md * 1e3
        print("%s S=%d B=%d: %.2f ms total, %.3f ms/step, %.0f sample-steps/s" % (key, len(ts_), B, md * 1e3, md * 1e3 / len(ts_), B * len(ts_) / md), flush=True)
    lib.set_graph_mode(0)
    if a.quick:
        return
    uc = cases.context(case, 2, B).to(dev)
    mn, md = timed(lambda: unet.ddim_sample(x, c, w, ts_, sched, uc=uc, scale=5.0), reps=2)
    res["ddim_cfg_ms"] = md * 1e3
    print("ddim CFG S=%d B=%d (U-Net batch %d): %.2f ms, %.3f ms/step" % (len(ts_), B, 2 * B, md * 1e3, md * 1e3 / len(ts_)), flush=True)
    # ---- decode
    lat = cases.randn(3, 1, (B, 16, z)).to(dev)
    mn, md = timed(lambda: vae.decode(lat))
    res["vae_ms"] = md * 1e3
    print("vae decode B=%d: %.3f ms" % (B, md * 1e3), flush=True)
    print("  vae profile:", {k: round(v["ms"], 3) for k, v in vae.profile().items() if v["launches"]})
    if not a.skip_wave:
        mel = cases.mel_input(case, 4, 1, z * 64).to(dev)
        mn, md = timed(lambda: wave.encode(mel, only_last=4))
        res["wave_ms"] = md * 1e3
        print("wave encode B=1 frames=%d: %.3f ms" % (z * 64, md * 1e3), flush=True)
        print("  wave profile:", {k: round(v["ms"], 3) for k, v in wave.profile().items() if v["launches"]})
        pcm = torch.from_numpy(__import__("oracle.host", fromlist=["x"]).synth_audio(z * 64 * 128 / 22050.0 - 0.5)).to(dev)
        mn, md = timed(lambda: lib.log_mel(pcm))
        res["mel_ms"] = md * 1e3
        print("log-mel %d samples: %.3f ms" % (pcm.numel(), md * 1e3), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
