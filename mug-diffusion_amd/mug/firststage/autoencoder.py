"""mug.firststage.autoencoder -- AutoencoderKL (mug/firststage/autoencoder.py:13-77).  Only `decode`
is on the sampling path and runs in libmugd; the encoder's parameters are registered so that the
reference's checkpoints load with identical keys, but `encode` (training / inpainting) is not built."""
import numpy as np          # noqa: F401
import shutil               # noqa: F401
import torch
import torch.nn as nn       # noqa: F401
import torch.nn.functional as F   # noqa: F401

from mug.model import specs
from mug.model.native_module import NativeModule
from mug.util import instantiate_from_config   # noqa: F401  (webui.py picks it up through `import *`)


class AutoencoderKL(NativeModule):
    def __init__(self, ddconfig, lossconfig=None, ckpt_path=None, remove_prefix=None, ignore_keys=None,
                 training_keys=None, monitor=None, kl_weight=0.0, scale=1.0, constant_var=None):
        super().__init__()
        dd = dict(ddconfig)
        dd.setdefault("num_groups", 32)
        dd["channel_mult"] = [int(m) for m in dd["channel_mult"]]
        self.scale = scale
        self.kl_weight = kl_weight
        if monitor is not None:
            self.monitor = monitor
        if constant_var is not None:
            raise NotImplementedError("constant_var is a training-time option")
        self._setup(specs.vae(dd), dd)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys, remove_prefix=remove_prefix)

    def init_from_ckpt(self, path, ignore_keys=None, remove_prefix=None):
        sd = torch.load(path, map_location="cpu")["state_dict"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in (ignore_keys or [])):
                del sd[k]
        if remove_prefix is not None:
            sd = {k.replace(remove_prefix, ""): v for k, v in sd.items() if k.startswith(remove_prefix)}
        missing, unexpected = self.load_state_dict(sd, strict=False)
        print(f"Restored from {path}, missing = {len(missing)}, unexpected = {len(unexpected)}")

    def _make_native(self, lib):
        return lib.vae(self._cfg, scale=self.scale)

    @torch.no_grad()
    def decode(self, z):
        return self.native().decode(z)

    def encode(self, x):
        raise NotImplementedError("AutoencoderKL.encode (training / inpainting) is outside the native sampling path")

    def forward(self, input, sample_posterior=True):
        raise NotImplementedError("AutoencoderKL.forward needs the encoder (training only)")
