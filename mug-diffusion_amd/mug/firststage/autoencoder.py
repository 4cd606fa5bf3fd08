"""mug.firststage.autoencoder -- AutoencoderKL (mug/firststage/autoencoder.py:13-77).  `decode` (sampling path) and
`encode` (inpainting / partial regeneration: SURVEY 8f rank 3) run in libmugd; `forward` / losses are training-only."""
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

    def native_encoder(self):
        """A second native network over the same parameters (names "encoder.*")."""
        lib = self.native().lib                      # registers / validates the parameters first
        if getattr(self, "_native_enc", None) is None or self._native_enc.lib is not lib or self._enc_fp != self._fp:
            self._native_enc = lib.vae(self._cfg, scale=self.scale, encoder=True)
            self._native_enc.set_params(dict(self._tensor_list()))
            self._enc_fp = self._fp
        return self._native_enc

    @torch.no_grad()
    def encode(self, x):
        """autoencoder.py:67-73: the posterior over latents of a note-grid tensor (no gradients: inference use only)."""
        return DiagonalGaussianDistribution(self.native_encoder().vae_encode(x), scale=self.scale)

    def forward(self, input, sample_posterior=True):
        raise NotImplementedError("AutoencoderKL.forward needs the encoder (training only)")


class DiagonalGaussianDistribution(object):
    """autoencoder.py:356-388 (the members inference code reads)."""

    def __init__(self, parameters, deterministic=False, scale=1.0, logvar=None):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        if logvar is not None:
            self.logvar = logvar * torch.ones_like(self.mean)
        self.logvar = torch.clamp(self.logvar, -10.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        self.scale = scale
        if self.deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self):
        return (self.mean + self.std * torch.randn(self.mean.shape).to(device=self.parameters.device)) * self.scale

    def mode(self):
        return self.mean * self.scale
