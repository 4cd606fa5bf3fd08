"""GPU probe (development tool, run through gpurun): one whole-model training step (mug.train.training_step) of the SHIPPED
architecture on synthetic data -- loss, gradient sanity, wall time.  python tests/gpu_train_probe.py [--B 4] [--z 512] [--reps 2]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mug-diffusion_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import cases, weights  # noqa: E402
from mug import train  # noqa: E402
from mug._native import get_lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=4)
    ap.add_argument("--z", type=int, default=512)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--cprofile", action="store_true", help="cProfile of the host side of the last step")
    ap.add_argument("--adamw", action="store_true", help="include the optimiser step")
    ap.add_argument("--bf16", action="store_true", help="GEMMs on the bf16 matrix cores (mugd_train_set_precision)")
    ap.add_argument("--nosync", action="store_true", help="no host synchronisation between the steps (what bench.py's training leg times): one line for all of them")
    a = ap.parse_args()
    case = cases.TINY if a.tiny else cases.FULL
    lib = get_lib()
    lib.train_set_precision(a.bf16)
    man = weights.load_manifest(os.path.join(cases.GOLDEN, case["manifest"]))
    sd = weights.set_s4_lengths(weights.make_state_dict(man, 0), case["unet"], a.z)
    sd = {k: (v.to(lib.device) if v.dtype == torch.float32 else v) for k, v in sd.items()}
    rng = np.random.default_rng(0)
    B, z = a.B, a.z
    x0 = torch.from_numpy(rng.standard_normal((B, 16, z)).astype(np.float32))
    noise = torch.from_numpy(rng.standard_normal((B, 16, z)).astype(np.float32))
    t = torch.from_numpy(rng.integers(0, 1000, B))
    ids = torch.from_numpy(rng.integers(0, sd["model.cond_stage_model.embedding.weight"].shape[0], (B, case["n_ctx_tok"])))
    mel = torch.from_numpy(np.abs(rng.standard_normal((B, case["wave"]["n_freq"], z * case["audio_ratio"]))).astype(np.float32))
    x0, noise, t, ids, mel = (v.to(lib.device) for v in (x0, noise, t, ids, mel))        # resident inputs, like bench.py's training leg
    plan = train.TrainPlan(lib, sd, case["unet"], case["wave"])
    opt = None
    prof = None
    if a.nosync:
        for r in range(2):                       # warm-up: weights packed, pool filled
            loss, grads = plan.step(x0, noise, t, ids, mel)
            if a.adamw:
                if opt is None:
                    opt = train.AdamW(lib, {k: sd[k] for k in grads}, grads, lr=1e-6)
                opt.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(a.reps):
            loss, grads = plan.step(x0, noise, t, ids, mel)
            if a.adamw:
                opt.step()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%s %d steps without host synchronisation: %.2f ms per step (host enqueue %.2f ms per step), %.1f samples/s" % ("bf16" if a.bf16 else "fp32", a.reps, dt / a.reps * 1e3, t_host / a.reps * 1e3, B * a.reps / dt), flush=True)
        return
    for r in range(a.reps):
        torch.cuda.synchronize()
        if a.cprofile and r == a.reps - 1:
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        t0 = time.perf_counter()
        loss, grads = plan.step(x0, noise, t, ids, mel)
        if a.adamw:
            if opt is None:
                opt = train.AdamW(lib, {k: sd[k] for k in grads}, grads, lr=1e-6)
            opt.step()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if prof is not None:
            prof.disable()
        line = ("bf16 " if a.bf16 else "fp32 ") + "step %d: %.3f s wall, host enqueue %.3f s (%.1f samples/s)" % (r, dt, t_host, B / dt)
        if r == a.reps - 1 or r == 0:
            bad = [k for k, g in grads.items() if not torch.isfinite(g).all()]
            gn = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())))
            line += "  loss %.6f  %d gradient tensors (%d non-finite)  |g| %.4e" % (float(loss), len(grads), len(bad), gn)
        print(line, flush=True)
    if prof is not None:
        import pstats
        pstats.Stats(prof).sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
