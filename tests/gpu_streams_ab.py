"""Development tool (run through gpurun): do the fixed costs of the U-Net's dependent launch chain overlap ACROSS queues?
N host threads, each with its own library context (own stream) and its own U-Net instance, run the batch-b DDIM loop concurrently;
the figure of merit is aggregate sample-steps/s against one context at batch b and against one context at batch N*b.
python tests/gpu_streams_ab.py [--b 4] [--S 50] [--graph 0|2]"""
import argparse
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mug-diffusion_amd"))

from oracle import cases, sampler, weights  # noqa: E402
from mug._native import Lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--b", type=int, default=4)
    ap.add_argument("--S", type=int, default=50)
    ap.add_argument("--graph", type=int, default=0)
    ap.add_argument("--nmax", type=int, default=4)
    a = ap.parse_args()
    case, z = cases.FULL, 512
    man = weights.load_manifest(os.path.join(cases.GOLDEN, case["manifest"]))
    sd = weights.set_s4_lengths(weights.make_state_dict(man, 0), case["unet"], z)
    steps = sampler.ddim_step_scalars(sd["alphas_cumprod"].numpy(), a.S, 0.0)
    ts_ = [s["t"] for s in steps]
    sched = [[s["a_t"], s["a_prev"], s["sigma"], s["sqrt_1m_at"]] for s in steps]

    def make(bsz):
        li = Lib()
        li.set_graph_mode(a.graph)
        ui = li.unet(case["unet"]); ui.set_params(sd, "model.unet_model.")
        dev = li.device
        x = cases.x_T(1, bsz, z).to(dev)
        c = cases.context(case, 1, bsz).to(dev)
        w = [m.to(dev) for m in cases.audio_maps(case, 1, 1, z)]
        run = lambda: ui.ddim_sample(x, c, w, ts_, sched)
        run(); torch.cuda.synchronize()
        return li, ui, run

    def measure(runs, reps=3):
        best = 1e9
        for _ in range(reps):
            bar = threading.Barrier(len(runs) + 1)
            def body(r):
                bar.wait()
                r()
            th = [threading.Thread(target=body, args=(r,)) for r in runs]
            for t in th: t.start()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bar.wait()
            for t in th: t.join()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best

    ctxs = [make(a.b) for _ in range(a.nmax)]
    for n in range(1, a.nmax + 1):
        dt = measure([c[2] for c in ctxs[:n]])
        print("%d context(s) x batch %d (graph mode %d): %.2f ms per loop, %.0f sample-steps/s aggregate" % (n, a.b, a.graph, dt * 1e3, n * a.b * a.S / dt), flush=True)
    for n in (2, 4):
        if n <= a.nmax:
            big = make(n * a.b)
            dt = measure([big[2]])
            print("1 context x batch %d: %.2f ms per loop, %.0f sample-steps/s" % (n * a.b, dt * 1e3, n * a.b * a.S / dt), flush=True)
            del big


if __name__ == "__main__":
    main()
