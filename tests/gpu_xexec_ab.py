"""GPU probe (development tool, run through gpurun): the XCD-resident executor (csrc/xexec.hip) against one launch per op, same box,
alternating.  python tests/gpu_xexec_ab.py [--B 8 16] [--S 50] [--reps 3] [--csv gpurun_out/x.csv]
Prints per (batch, mode): DDIM loop ms per step, sample-steps/s; for mode 0 also the event-bracketed per-class profile."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mug-diffusion_amd"))

import torch  # noqa: E402

from oracle import cases, sampler, weights  # noqa: E402
from mug._native import get_lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, nargs="+", default=[8, 16])
    ap.add_argument("--z", type=int, default=512)
    ap.add_argument("--S", type=int, default=50)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--rounds", type=int, default=2, help="A/B alternations per batch size")
    ap.add_argument("--profile", default=None, help="path prefix: dump per-op event timings and the executor's per-phase timeline for the first batch size")
    a = ap.parse_args()
    case, z, S = cases.FULL, a.z, a.S
    lib = get_lib()
    man = weights.load_manifest(os.path.join(cases.GOLDEN, case["manifest"]))
    sd = weights.set_s4_lengths(weights.make_state_dict(man, 0), case["unet"], z)
    dev = lib.device
    steps = sampler.ddim_step_scalars(sd["alphas_cumprod"].numpy(), S, 0.0)
    ts_ = [s["t"] for s in steps]
    sched = [[s["a_t"], s["a_prev"], s["sigma"], s["sqrt_1m_at"]] for s in steps]
    nets = {}
    for mode in (0, 1):
        lib.set_exec_mode(mode)
        n = lib.unet(case["unet"])
        n.set_params(sd, "model.unet_model.")
        nets[mode] = n
    for B in a.B:
        x = cases.x_T(1, B, z).to(dev)
        c = cases.context(case, 1, B).to(dev)
        w = [m.to(dev) for m in cases.audio_maps(case, 1, max(1, B // 4), z)]
        best = {0: 1e9, 1: 1e9}
        lat = {}
        for r in range(a.rounds):
            for mode in (0, 1):
                lib.set_exec_mode(mode)              # the mode is read when the (batch, length) program is compiled: first call per net
                n = nets[mode]
                lat[mode] = n.ddim_sample(x, c, w, ts_, sched)
                torch.cuda.synchronize()
                for _ in range(a.reps):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    n.ddim_sample(x, c, w, ts_, sched)
                    torch.cuda.synchronize()
                    best[mode] = min(best[mode], time.perf_counter() - t0)
                print("B=%d round %d mode %d: best so far %.3f ms/step" % (B, r, mode, best[mode] * 1e3 / S), flush=True)
        d = (lat[0] - lat[1]).abs().max().item()
        for mode in (0, 1):
            print("RESULT B=%d z=%d S=%d %s: %.2f ms per loop, %.3f ms/step, %.0f sample-steps/s" % (
                B, z, S, "executor  " if mode else "per-op    ", best[mode] * 1e3, best[mode] * 1e3 / S, B * S / best[mode]), flush=True)
        print("RESULT B=%d latent max|executor - per-op| = %.3e (range %.1f); speed-up %.3fx" % (B, d, lat[0].abs().max().item(), best[0] / best[1]), flush=True)
    if a.profile:
        # per-op event profile (mode 0 program and the executor-mode program run op by op) + the executor's per-phase clock stamps
        B = a.B[0]
        x = cases.x_T(1, B, z).to(dev)
        t = torch.full((B,), 501, dtype=torch.long, device=dev)
        c = cases.context(case, 1, B).to(dev)
        w = [m.to(dev) for m in cases.audio_maps(case, 1, max(1, B // 4), z)]
        for mode in (0, 1):
            lib.set_exec_mode(mode)
            os.environ["MUGD_PROFILE_CSV"] = a.profile + ".mode%d_per_op.csv" % mode
            os.environ["MUGD_XEXEC_CSV"] = a.profile + ".executor_phases.csv"
            nets[mode].forward(x, t, c, w)
            prof = nets[mode].profile()
            tot = sum(v["ms"] for v in prof.values())
            print("PROFILE mode %d program, op by op (event-bracketed): %.3f ms over %d launches" % (mode, tot, sum(v["launches"] for v in prof.values())), flush=True)
    lib.set_exec_mode(0)


if __name__ == "__main__":
    main()
