"""mug.diffusion.unet -- UNetModel with the reference's constructor / forward signature
(mug/diffusion/unet.py:262-550) and state-dict keys; forward runs in libmugd (net.hip UNet)."""
import numpy as np          # noqa: F401  (webui.py star-imports np / torch / nn through this module)
import torch
import torch.nn as nn       # noqa: F401

from mug.model import s4 as s4host
from mug.model import specs
from mug.model.native_module import NativeModule
from mug.model.util import timestep_embedding, zero_module, conv_nd, linear, checkpoint  # noqa: F401


class UNetModel(NativeModule):
    def __init__(self, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 audio_channels, dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=1,
                 use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1,
                 use_scale_shift_norm=False, lstm_last=False, lstm_layer=False, s4_layer=False,
                 transformer_depth=1, context_dim=None):
        super().__init__()
        if dims != 1 or use_scale_shift_norm or lstm_layer or transformer_depth != 1 or num_head_channels != -1 or not conv_resample:
            raise NotImplementedError("only the configuration family of configs/mug/mug_diffusion.yaml is implemented natively")
        assert num_heads != -1, 'Either num_heads or num_head_channels has to be set'
        cfg = dict(in_channels=in_channels, model_channels=model_channels, out_channels=out_channels,
                   num_res_blocks=num_res_blocks, attention_resolutions=[int(a) for a in attention_resolutions],
                   channel_mult=[int(m) for m in channel_mult], num_heads=num_heads, context_dim=context_dim,
                   audio_channels=[int(a) for a in audio_channels], s4_layer=bool(s4_layer))
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, attention_resolutions
        self.channel_mult, self.num_heads, self.dropout = channel_mult, num_heads, dropout
        self.dtype = torch.float32          # the reference's use_fp16 only sets this attribute (unet.py:330)
        spec, self._s4_ds = specs.unet(cfg)
        self._setup(spec, cfg, s4host.s4_init)

    def _make_native(self, lib):
        return lib.unet(self._cfg)

    def s4_length_of(self, key, z):
        """Sequence length of the S4 layer owning `key` for latent length z."""
        for p, ds in self._s4_ds.items():
            if key.startswith(p + "."):
                return z // ds
        raise KeyError(key)

    @torch.no_grad()
    def prepare_length(self, z):
        """The reference grows each S4 kernel lazily inside forward (s4.py:726-730, mutating C and L);
        do the same here, before the native call, for every S4 layer at its level's length."""
        sd = dict(self._tensor_list())
        lbufs = [sd[p + ".s4_model.kernel.kernel.L"] for p in self._s4_ds]
        key = (int(z), hash(tuple((t.data_ptr(), t._version) for t in lbufs)))
        if getattr(self, "_len_ok", None) == key:          # lengths already checked for this z and these buffers: no device sync
            return
        for p, ds in self._s4_ds.items():
            k = p + ".s4_model.kernel.kernel."
            params = {n: sd[k + n] for n in s4host.S4_SUFFIXES}
            s4host.ensure_length_(params, z // ds)
        self._len_ok = (int(z), hash(tuple((t.data_ptr(), t._version) for t in lbufs)))

    @torch.no_grad()
    def forward(self, x, timesteps=None, context=None, *audios):
        if timesteps.dim() == 2:
            timesteps = timesteps[:, 0]
        self.prepare_length(x.shape[-1])
        return self.native().forward(x, timesteps, context, audios)

    def convert_to_fp16(self):
        pass

    def convert_to_fp32(self):
        pass

    def summary(self):
        pass
