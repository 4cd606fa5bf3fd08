"""Small helpers re-exported for API compatibility with mug/model/util.py."""
import math

import torch
import torch.nn as nn


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """mug/model/util.py:156-176 (host version; the sampler computes it on device in k_misc.hip)."""
    if repeat_only:
        return timesteps[:, None].repeat(1, dim)
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(timesteps.device)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


def conv_nd(dims, *args, **kwargs):
    return {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}[dims](*args, **kwargs)


def linear(*args, **kwargs):
    return nn.Linear(*args, **kwargs)


def checkpoint(func, inputs, params, flag):
    """Inference-only package: activation checkpointing is a pass-through (util.py:104-119 under no_grad)."""
    return func(*inputs)


def extract_into_tensor(a, t, x_shape):
    b = t.shape[0]
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))
