"""GPU probe (development tool, run through gpurun, usually under `rocprofv3 --kernel-trace`): the 50-step DDIM loop at
B = 4, z = 512 in ONE launch mode.   python tests/gpu_graph_vs_eager.py --mode graph|eager|graph_all [--reps 3]
Prints the wall time per step; the kernel trace of the run is reduced by tests/pp_kernel_gaps.py."""
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
    ap.add_argument("--mode", default="graph")
    ap.add_argument("--z", type=int, default=512)
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--S", type=int, default=50)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    case = cases.FULL
    lib = get_lib()
    man = weights.load_manifest(os.path.join(cases.GOLDEN, case["manifest"]))
    sd = weights.set_s4_lengths(weights.make_state_dict(man, 0), case["unet"], a.z)
    unet = lib.unet(case["unet"]); unet.set_params(sd, "model.unet_model.")
    dev = lib.device
    x = cases.x_T(1, a.B, a.z).to(dev)
    c = cases.context(case, 1, a.B).to(dev)
    w = [m.to(dev) for m in cases.audio_maps(case, 1, 1, a.z)]
    steps = sampler.ddim_step_scalars(sd["alphas_cumprod"].numpy(), a.S, 0.0)
    ts_ = [s["t"] for s in steps]
    sched = [[s["a_t"], s["a_prev"], s["sigma"], s["sqrt_1m_at"]] for s in steps]
    lib.set_graph_mode({"graph": 1, "eager": 0, "graph_all": 2}[a.mode])
    unet.ddim_sample(x, c, w, ts_, sched)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        unet.ddim_sample(x, c, w, ts_, sched)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print("ddim_%s S=%d B=%d z=%d: %.2f ms total, %.3f ms/step, %.0f sample-steps/s" % (a.mode, len(ts_), a.B, a.z, best * 1e3, best * 1e3 / len(ts_), a.B * len(ts_) / best), flush=True)


if __name__ == "__main__":
    main()
