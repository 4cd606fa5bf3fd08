"""TEST INFRASTRUCTURE (see oracle/__init__.py).

Deterministic weight factory.  No trained checkpoint exists in the reference
repo, so every parity check runs on seeded random weights that
can be regenerated anywhere from (manifest, seed): each tensor is drawn from a
numpy Generator keyed by (seed, crc32(name)).  Tensors the reference
zero-initialises (zero_module: unet.py:187-189,492,82-84; attention.py:180) get
non-zero values too, otherwise every residual branch is a no-op and parity
passes vacuously.

S4 parameters are drawn around the HiPPO-LegS NPLR initialisation
(mug/model/s4.py:379-436, restated in numpy) so the kernels are well behaved,
with `C` random and the length buffer `L` set to the layer's sequence length:
the state a trained checkpoint is in (SURVEY.md section 7 "S4 statefulness").
"""
import json
import zlib

import numpy as np
import torch

from . import nets, sampler


def hippo_legs_nplr(N=64):
    """(w, P, B) of the rank-1 NPLR form of HiPPO-LegS, keeping N//2 conjugate
    halves sorted by imaginary part -- s4.py:296-305 (transition 'legs'),
    :345-347 (rank_correction), :379-436 (nplr)."""
    q = np.arange(N, dtype=np.float64)
    col, row = np.meshgrid(q, q)
    r = 2 * q + 1
    M = -(np.where(row >= col, r, 0) - np.diag(q))
    T = np.sqrt(np.diag(2 * q + 1))
    A = T @ M @ np.linalg.inv(T)
    B = np.sqrt(2 * q + 1)
    P = np.sqrt(0.5 + q)
    AP = A + np.outer(P, P)
    w_re = np.mean(np.diagonal(AP))
    w_im, V = np.linalg.eigh(AP * -1j)
    w = w_re + 1j * w_im
    idx = np.argsort(w.imag)
    w = w[idx][: N // 2]
    V = V[:, idx][:, : N // 2]
    Vinv = V.conj().T
    return w, Vinv @ P, Vinv @ B


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def make_state_dict(manifest, seed=0):
    """manifest: list of [name, shape, dtype-string].  Returns name -> torch tensor."""
    sched = sampler.register_schedule()
    w0, P0, B0 = hippo_legs_nplr(64)
    sd = {}
    for name, shape, dtype in manifest:
        shape = tuple(shape)
        g = _rng(seed, name)
        if name in sched:
            a = sched[name]
        elif name.endswith(".kernel.kernel.L"):
            a = np.zeros((), dtype=np.int64)          # filled by set_s4_lengths
        elif name.endswith(".kernel.kernel.C"):
            a = g.normal(0.0, np.sqrt(0.5), shape)
        elif name.endswith(".kernel.kernel.log_dt"):
            a = g.uniform(np.log(1e-3), np.log(1e-1), shape)
        elif name.endswith(".kernel.kernel.B") or name.endswith(".kernel.kernel.P"):
            base = B0 if name.endswith(".B") else P0
            nh = shape[-2]
            assert nh == len(base), (name, shape)
            c = base[None, None, :] * (1.0 + 0.05 * g.normal(size=shape[:-1])) + 0.02 * g.normal(size=shape[:-1])
            a = np.stack([c.real, c.imag], axis=-1)
        elif name.endswith(".kernel.kernel.inv_w_real"):
            a = np.log(0.5) + 0.1 * g.normal(size=shape)
        elif name.endswith(".kernel.kernel.w_imag"):
            a = w0.imag[None, :] * (1.0 + 0.02 * g.normal(size=shape))
        elif name.endswith(".s4_model.D"):
            a = g.normal(size=shape)
        elif name.endswith("relative_position_embedding"):
            a = 0.3 * g.normal(size=shape)
        elif name.endswith("C_embedding"):
            a = 1.0 + 0.2 * g.normal(size=shape)
        elif name.endswith("embedding.weight"):
            a = g.normal(size=shape)
        elif name.endswith(".weight") and len(shape) == 1:
            a = 1.0 + 0.1 * g.normal(size=shape)       # GroupNorm / LayerNorm gain
        elif name.endswith(".bias"):
            a = 0.05 * g.normal(size=shape)
        elif name.endswith(".weight"):
            fan_in = int(np.prod(shape[1:]))
            a = g.normal(0.0, 1.0 / np.sqrt(fan_in), shape)
        else:
            raise KeyError("weight factory has no rule for %s %s" % (name, shape))
        td = getattr(torch, dtype)
        sd[name] = torch.from_numpy(np.ascontiguousarray(a)).to(td).reshape(shape)
    return sd


def set_s4_lengths(sd, unet_cfg, z, prefix="model.unet_model"):
    """Sets every S4 `L` buffer to the sequence length its layer runs at for
    latent length z (z, z/2, z/4, z/8 by level)."""
    inp, out = nets.unet_plan(unet_cfg)
    T = z
    for i, mod in enumerate(inp):
        if mod[0] == "down":
            T //= 2
        if mod[0] == "seq":
            for j, kind in enumerate(mod[1]):
                if kind == "s4":
                    sd["%s.input_blocks.%d.%d.s4_model.kernel.kernel.L" % (prefix, i, j)] = torch.tensor(T, dtype=torch.int64)
    for i, mod in enumerate(out):
        if mod[0] == "seq":
            for j, kind in enumerate(mod[1]):
                if kind == "s4":
                    sd["%s.output_blocks.%d.%d.s4_model.kernel.kernel.L" % (prefix, i, j)] = torch.tensor(T, dtype=torch.int64)
                if kind == "up":
                    T *= 2
    return sd


def load_manifest(path):
    with open(path) as f:
        return json.load(f)


def manifest_of(state_dict):
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in state_dict.items()]
