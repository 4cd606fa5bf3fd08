"""Development tool (run through gpurun, usually under `rocprofv3 --pmc ...`): compiles the full-size U-Net program at the
benchmark shape (batch 4, z = 512) and evaluates it `--n` times -- a workload small enough for counter collection, whose
conv_gemm launches are exactly those of one DDIM step."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mug-diffusion_amd"))

import torch  # noqa: E402

from oracle import cases, weights  # noqa: E402
from mug._native import get_lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--z", type=int, default=512)
    ap.add_argument("--B", type=int, default=4)
    a = ap.parse_args()
    case = cases.FULL
    lib = get_lib()
    man = weights.load_manifest(os.path.join(cases.GOLDEN, case["manifest"]))
    sd = weights.set_s4_lengths(weights.make_state_dict(man, 0), case["unet"], a.z)
    unet = lib.unet(case["unet"])
    unet.set_params({k: v for k, v in sd.items() if k.startswith("model.unet_model.")}, "model.unet_model.")
    x = cases.x_T(1, a.B, a.z)
    t = torch.full((a.B,), 501, dtype=torch.long)
    c = cases.context(case, 1, a.B)
    w = cases.audio_maps(case, 1, 1, a.z)
    for _ in range(a.n):
        eps = unet.forward(x, t, c, w)
    torch.cuda.synchronize()
    print("ok", float(eps.abs().mean()))


if __name__ == "__main__":
    main()
