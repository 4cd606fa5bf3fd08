"""TEST INFRASTRUCTURE (see oracle/__init__.py).

Shared, seed-reproducible test cases: model configs and synthetic inputs.  All
random inputs come from numpy Generators (bit-stable across machines), never
from torch's RNG, so the authoring container and the GPU box see identical data.
"""
import os

import numpy as np
import torch
import yaml

from . import host, nets

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), "tests", "golden")

FULL = dict(name="full", unet=nets.UNET_DEFAULT, wave=nets.WAVE_DEFAULT, vae=nets.VAE_DEFAULT,
            z_channels=16, n_ctx_tok=21, audio_ratio=64, manifest="manifest_full.json")

# A structurally complete miniature (every block kind, two levels, attention on the
# second level, S4 everywhere, odd head dim) small enough for the CPU emulator.
TINY = dict(
    name="tiny",
    unet=dict(in_channels=16, model_channels=32, out_channels=16, num_res_blocks=1,
              attention_resolutions=[2], channel_mult=[1, 2], num_heads=4, context_dim=32,
              audio_channels=[32, 64], s4_layer=True),
    wave=dict(n_freq=32, middle_channels=32, attention_resolutions=[4], num_res_blocks=2,
              num_heads=4, num_groups=32, channel_mult=[1, 1, 2]),
    vae=dict(x_channels=16, middle_channels=32, z_channels=16, num_groups=8,
             channel_mult=[1, 2], num_res_blocks=1),
    z_channels=16, n_ctx_tok=21, audio_ratio=2, manifest="manifest_tiny.json")


def feature_yaml():
    with open(os.path.join(GOLDEN, "mania_beatmap_features.yaml")) as f:
        return yaml.safe_load(f)


def rng(seed, *salt):
    return np.random.default_rng([int(seed)] + [int(s) for s in salt])


def randn(seed, salt, shape):
    return torch.from_numpy(rng(seed, salt).standard_normal(shape).astype(np.float32))


def x_T(seed, B, z, zc=16):
    """One generator per sample (seed + i), as SURVEY.md 8(d) prescribes."""
    return torch.stack([randn(seed + i, 1, (zc, z)) for i in range(B)])


def audio_maps(case, seed, B, z):
    """Synthetic stand-ins for the wave-encoder outputs consumed by the U-Net
    (the last len(channel_mult) maps: channels audio_channels[l] at length z/2^l)."""
    chans = case["unet"]["audio_channels"]
    return [0.5 * randn(seed, 10 + l, (B, c, z >> l)) for l, c in enumerate(chans)]


def context(case, seed, B):
    return randn(seed, 3, (B, case["unet"]["context_dim"], case["n_ctx_tok"]))


def mel_input(case, seed, B, frames):
    """log-mel-like non-negative input, fp16-rounded like the reference's (util.py:143)."""
    n = case["wave"]["n_freq"]
    m = np.abs(rng(seed, 5).standard_normal((B, n, frames))).astype(np.float16).astype(np.float32) * 2.0
    return torch.from_numpy(m.astype(np.float16).astype(np.float32))


def train_batch(case, seed, B, z, n_ids):
    """The training batch of the tests/golden/*_train_* fixtures (oracle/gen_golden.py: train_goldens): note-grid tensor
    (B, x_channels, 8 z), fp16-rounded log-mel (B, n_freq, z * audio_ratio), prompt ids (B, 21)."""
    up = 2 ** (len(case["vae"]["channel_mult"]) - 1)
    note_t = randn(seed, 1, (B, case["vae"]["x_channels"], z * up))
    mel = mel_input(case, seed, B, z * case["audio_ratio"])
    ids = torch.from_numpy(rng(seed, 2).integers(0, n_ids, (B, case["n_ctx_tok"])))
    return note_t, mel, ids


def train_sample_index(seed, i, n):
    """Flat positions kept of the i-th fully stored gradient when it has more than 8192 elements."""
    return np.sort(rng(seed, 9, i).choice(n, size=8192, replace=False))
