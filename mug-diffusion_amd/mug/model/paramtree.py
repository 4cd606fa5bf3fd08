"""State-dict-compatible parameter containers.

The reference networks are deep trees of torch modules whose only contract with the outside
world (webui.py:52-58, diffusion.py:191-209) is their state-dict key set.  The native networks
do not need that module tree -- libmugd walks the keys itself -- so the Python side registers
parameters from a flat spec [(dotted.key, shape, kind)] into nested `Branch` containers, which
reproduces the reference's keys (1515 for the shipped config) without mirroring every class.
"""
import math
import zlib

import numpy as np
import torch
import torch.nn as nn


class Branch(nn.Module):
    """A bare container; children may be named '0', '1', ... like ModuleList/Sequential entries."""


def register_spec(root, spec):
    """spec: iterable of (key, shape, kind).  kinds: conv | linear | zero | norm_w | norm_b | bias |
    embed | rel | cemb | s4_* | buffer_i64."""
    for key, shape, kind in spec:
        parts = key.split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, Branch())
            mod = mod._modules[p]
        name = parts[-1]
        if kind == "buffer_i64":
            mod.register_buffer(name, torch.zeros(tuple(shape), dtype=torch.int64))
        else:
            mod.register_parameter(name, nn.Parameter(torch.empty(tuple(shape), dtype=torch.float32)))


def _fan_in(shape):
    return int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])


@torch.no_grad()
def init_spec(root, spec, s4_init=None):
    """PyTorch-default-like initialisation, with the reference's zero_module tensors set to zero
    (unet.py:82-84,187-189,492; attention.py:180) and HiPPO-LegS NPLR values for the S4 kernels."""
    sd = dict(root.named_parameters())
    sd.update(dict(root.named_buffers()))
    for key, shape, kind in spec:
        t = sd[key]
        if kind in ("conv", "linear"):
            bound = 1.0 / math.sqrt(_fan_in(shape))           # == kaiming_uniform_(a=sqrt(5))
            t.uniform_(-bound, bound)
        elif kind == "bias":
            wkey = key[:-4] + "weight"
            bound = 1.0 / math.sqrt(_fan_in(sd[wkey].shape)) if wkey in sd else 0.0
            t.uniform_(-bound, bound)
        elif kind in ("zero", "norm_b", "rel"):
            t.zero_()
        elif kind in ("norm_w", "cemb"):
            t.fill_(1.0)
        elif kind == "embed":
            t.normal_()
        elif kind == "buffer_i64":
            t.zero_()
        elif kind.startswith("s4_"):
            s4_init(key, kind, t)
        else:
            raise KeyError(kind)


@torch.no_grad()
def seed_all_parameters(module, seed=0, s4_length_of=None):
    """Deterministic synthetic weights for benchmarks / smoke tests: every float tensor, INCLUDING
    the zero-initialised ones (otherwise every residual branch is a no-op and the U-Net outputs
    exactly 0), is drawn from a numpy Generator keyed by (seed, crc32(key)).  S4 `C` is drawn
    complex-normal, the other S4 tensors keep their HiPPO initialisation, and each S4 length
    buffer `L` is set via s4_length_of(key) (the state a trained checkpoint is in)."""
    for key, t in list(module.named_parameters()) + list(module.named_buffers()):
        g = np.random.default_rng([int(seed), zlib.crc32(key.encode())])
        if t.dtype == torch.int64:
            if key.endswith(".kernel.kernel.L") and s4_length_of is not None:
                t.fill_(int(s4_length_of(key)))
            continue
        shape = tuple(t.shape)
        if ".kernel.kernel." in key:
            if key.endswith(".C"):
                a = g.normal(0.0, math.sqrt(0.5), shape)
            else:
                continue
        elif key.endswith("relative_position_embedding"):
            a = 0.3 * g.normal(size=shape)
        elif key.endswith("C_embedding"):
            a = 1.0 + 0.2 * g.normal(size=shape)
        elif key.endswith("embedding.weight") or key.endswith(".s4_model.D"):
            a = g.normal(size=shape)
        elif key.endswith(".weight") and len(shape) == 1:
            a = 1.0 + 0.1 * g.normal(size=shape)
        elif key.endswith(".bias"):
            a = 0.05 * g.normal(size=shape)
        elif key.endswith(".weight"):
            a = g.normal(0.0, 1.0 / math.sqrt(_fan_in(shape)), shape)
        else:
            continue                     # schedule buffers etc.
        t.copy_(torch.from_numpy(np.ascontiguousarray(a)).to(t.dtype))
    return module


def fingerprint(module):
    """Cheap change detector for lazily re-registering parameters with the native library."""
    fp = []
    for _, t in list(module.named_parameters()) + list(module.named_buffers()):
        fp.append((t.data_ptr(), t._version, t.device.index if t.is_cuda else -1))
    return hash(tuple(fp))
