"""GPU micro-benchmark (development tool, run through gpurun): the bf16 training GEMMs on the shapes that dominate a batch-32 training
step -- forward conv (tconv incl. its weight pack), and forward + backward (tconv x2 + twgrad) through mugd_train_conv.
python tests/gpu_tgemm_bench.py [--reps 20]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mug-diffusion_amd"))

import torch  # noqa: E402

from mug._native import get_lib  # noqa: E402

SHAPES = [  # name, B, Cin, Cout, T, taps, dil
    ("wave L0 conv3 128->128 T=32768", 32, 128, 128, 32768, 3, 1),
    ("wave L0 conv3 dil 8", 32, 128, 128, 32768, 3, 8),
    ("wave L2 conv3 128->128 T=8192", 32, 128, 128, 8192, 3, 1),
    ("wave L4 conv3 256->256 T=2048", 32, 256, 256, 2048, 3, 1),
    ("unet L0 res conv3 640->128 T=512", 32, 640, 128, 512, 3, 1),
    ("unet L1 res conv3 256->256 T=256", 32, 256, 256, 256, 3, 1),
    ("unet L3 res conv3 512->512 T=64", 32, 512, 512, 64, 3, 1),
    ("unet L3 res conv3 1536->512 T=64", 32, 1536, 512, 64, 3, 1),
    ("unet L3 linear 512->512 T=64", 32, 512, 512, 64, 1, 1),
    ("unet L3 GEGLU 512->4096 T=64", 32, 512, 4096, 64, 1, 1),
    ("unet L3 ff2 2048->512 T=64", 32, 2048, 512, 64, 1, 1),
    ("unet L1 linear 256->256 T=256", 32, 256, 256, 256, 1, 1),
    ("unet L1 GEGLU 256->2048 T=256", 32, 256, 2048, 256, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    lib = get_lib()
    lib.train_set_precision(True)
    dev = lib.device
    print("%-38s %10s %9s %9s   %10s %9s" % ("shape", "fwd us", "TF/s", "GB/s", "fwd+bwd us", "TF/s"))
    for name, B, Cin, Cout, T, taps, dil in SHAPES:
        if a.only and a.only not in name:
            continue
        w = torch.randn(Cout, Cin, taps, device=dev) * (Cin * taps) ** -0.5
        b = torch.randn(Cout, device=dev)
        x = torch.randn(B, Cin, T, device=dev)
        dy = torch.randn(B, Cout, T, device=dev)
        fl = 2.0 * Cout * Cin * taps * B * T
        byts = 4.0 * B * T * (Cin + Cout)
        with lib.on_stream():
            for mode in ("fwd", "both"):
                f = (lambda: lib.train_conv(w, b, x, None, dil=dil)) if mode == "fwd" else (lambda: lib.train_conv(w, b, x, dy, dil=dil))
                f(); f()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.reps):
                    f()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / a.reps
                if mode == "fwd":
                    s = "%-38s %10.1f %9.1f %9.0f" % (name, dt * 1e6, fl / dt / 1e12, byts / dt / 1e9)
                else:
                    s += "   %10.1f %9.1f" % (dt * 1e6, 3 * fl / dt / 1e12)
        print(s, flush=True)
        del w, b, x, dy
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
