"""Development tool (run through gpurun with a -DMUGD_H3_COUNT variant: MUGD_LIB_PATH=tests/var/h3count/libmugd.so): how often the H3 domain
machinery of conv_gemm leaves its fast path on the shipped U-Net (z = 512, batch 4) -- parks, slow paths, rescales, bail-outs per evaluation."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mug-diffusion_amd"))

import torch  # noqa: E402

from oracle import cases, weights  # noqa: E402
from mug._native import get_lib  # noqa: E402

lib = get_lib()
case, z, B = cases.FULL, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 4
man = weights.load_manifest(os.path.join(cases.GOLDEN, case["manifest"]))
sd = weights.set_s4_lengths(weights.make_state_dict(man, 0), case["unet"], z)
unet = lib.unet(case["unet"])
unet.set_params({k: v for k, v in sd.items() if k.startswith("model.unet_model.")}, "model.unet_model.")
x, t = cases.x_T(1, B, z), torch.full((B,), 481, dtype=torch.long)
c, w = cases.context(case, 1, B), cases.audio_maps(case, 1, 1, z)
ev = (ctypes.c_ulonglong * 8)()
unet.forward(x, t, c, w)
lib.dll.mugd_dev_h3_counters(ev)
unet.forward(x, t, c, w)
lib.dll.mugd_dev_h3_counters(ev)
names = ["parks", "slow paths", "rescales", "tile redos (waves)", "all-zero chunks", "off-band high", "off-band low"]
print("one U-Net evaluation, z = %d, batch %d: " % (z, B) + ", ".join("%s %d" % (n, ev[i]) for i, n in enumerate(names)))
