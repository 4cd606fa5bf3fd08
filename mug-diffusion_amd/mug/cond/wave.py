"""mug.cond.wave -- MelspectrogramScaleEncoder1D (mug/cond/wave.py:398-464) on libmugd."""
import numpy as np          # noqa: F401
import torch
import torch.nn as nn       # noqa: F401

from mug.model import specs
from mug.model.native_module import NativeModule


class MelspectrogramScaleEncoder1D(NativeModule):
    def __init__(self, *, n_freq, middle_channels, attention_resolutions, num_heads, num_groups, channel_mult,
                 num_res_blocks, use_checkpoint=True, dropout=0.0, **ignore_kwargs):
        super().__init__()
        cfg = dict(n_freq=n_freq, middle_channels=middle_channels,
                   attention_resolutions=[int(a) for a in attention_resolutions], num_heads=num_heads,
                   num_groups=num_groups, channel_mult=[int(m) for m in channel_mult], num_res_blocks=num_res_blocks)
        self.num_resolutions = len(cfg["channel_mult"])
        self.num_res_blocks = num_res_blocks
        self._setup(specs.wave(cfg), cfg)

    def _make_native(self, lib):
        return lib.wave(self._cfg)

    @torch.no_grad()
    def forward(self, x):
        """mel (B, n_freq, Ta) -> list with one feature map per level (the U-Net consumes the last four)."""
        return self.native().encode(x)

    def summary(self):
        pass
