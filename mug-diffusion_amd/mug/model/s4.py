"""Host-side S4 (NPLR) weight handling for the native U-Net.

The per-step work of the reference's S4 layer (mug/model/s4.py:1471-1541: kernel generation +
long convolution + GELU + GLU) runs in libmugd (k_s4.hip).  What stays on the host is what the
reference also does outside the steady state:

* `hippo_legs_nplr` -- the HiPPO-LegS NPLR initialisation (s4.py:296-305, 345-347, 379-436), for
  constructing a fresh model;
* `setup_C_` -- SSKernelNPLR._setup_C (s4.py:557-584): the one-time C -> C~ = C (I - dA^L)
  transform and the length-doubling step C~ <- C~ (I + dA^L) the reference applies lazily inside
  `forward` whenever a longer sequence than the stored buffer `L` is requested (s4.py:726-730).
  Like the reference it MUTATES `C` and `L` in place.
"""
import math

import numpy as np
import torch

S4_SUFFIXES = ("C", "log_dt", "B", "P", "inv_w_real", "w_imag", "L")


def hippo_legs_nplr(N=64):
    """(w, P, B), complex128 numpy, keeping the N//2 conjugate halves sorted by imaginary part."""
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


_HIPPO = {}


@torch.no_grad()
def s4_init(key, kind, t, d_state=64, dt_min=0.001, dt_max=0.1):
    """Fresh-model values (SSKernel.__init__, s4.py:1259-1306; S4.__init__ :1433 for D)."""
    if d_state not in _HIPPO:
        _HIPPO[d_state] = hippo_legs_nplr(d_state)
    w, P, B = _HIPPO[d_state]
    H = t.shape[-3] if t.dim() == 4 else t.shape[0]
    if kind == "s4_C":
        t.copy_(torch.randn(t.shape) * math.sqrt(0.5))            # randn(cfloat): each component var 1/2
    elif kind == "s4_log_dt":
        t.copy_(torch.rand(t.shape) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min))
    elif kind in ("s4_B", "s4_P"):
        v = torch.from_numpy(B if kind == "s4_B" else P).to(torch.complex64)
        t.copy_(torch.view_as_real(v)[None, None].expand(1, H, -1, -1))
    elif kind == "s4_inv_w_real":
        t.copy_(torch.log(-torch.clamp(torch.from_numpy(w.real).float(), max=-1e-3))[None].expand(H, -1))
    elif kind == "s4_w_imag":
        t.copy_(torch.from_numpy(w.imag).float()[None].expand(H, -1))
    elif kind == "s4_D":
        t.normal_()
    else:
        raise KeyError(kind)


def _c(t):
    return torch.view_as_complex(t.contiguous())


def _conj_cat(x):
    return torch.cat([x, x.conj()], dim=-1)


@torch.no_grad()
def _discrete_A(params):
    """dA (H, 2N, 2N) of the bilinear-discretised NPLR system, built like _setup_linear +
    _step_state_linear + _setup_state (s4.py:833-923) on the conjugate-expanded state."""
    dt = torch.exp(params["log_dt"])
    w = -torch.exp(params["inv_w_real"]) + 1j * params["w_imag"]          # (H,N)
    P = _c(params["P"])                                                   # (1,H,N)
    Q = P.conj()
    D = (2.0 / dt.unsqueeze(-1) - w).reciprocal()                        # (H,N)
    r = 1.0 + 2.0 * torch.einsum("rhn,hn,shn->hrs", Q, D, P).real       # (H,1,1)
    R = torch.linalg.solve(r.to(Q.dtype), (Q * D).permute(1, 0, 2)).permute(1, 0, 2)   # (1,H,N)
    E = 2.0 / dt.unsqueeze(-1) + w
    Dx, Ex, Px, Qx, Rx = _conj_cat(D), _conj_cat(E), _conj_cat(P), _conj_cat(Q), _conj_cat(R)
    N2 = Dx.shape[-1]
    state = torch.eye(N2, dtype=Dx.dtype, device=Dx.device).unsqueeze(-2)        # (2N,1,2N)
    ns = Ex * state - torch.einsum("rhn,rhm,...hm->...hn", Px, Qx, state)
    ns = Dx * (ns - torch.einsum("rhn,rhm,...hm->...hn", Px, Rx, ns))
    return ns.permute(1, 2, 0)                                            # "n h m -> h m n"


def _power(L, A):
    """A^L by repeated squaring (s4.py:243-259)."""
    I = torch.eye(A.shape[-1], dtype=A.dtype, device=A.device)
    powers = [A]
    while True:
        if L % 2 == 1:
            I = powers[-1] @ I
        L //= 2
        if L == 0:
            break
        powers.append(powers[-1] @ powers[-1])
    return I


@torch.no_grad()
def setup_C_(params, L):
    """One call of SSKernelNPLR._setup_C(L) (s4.py:557-584).  params: dict with the seven kernel
    tensors ('C','B','P','inv_w_real','w_imag','log_dt','L'); mutates params['C'] and params['L']."""
    cur = int(params["L"].item())
    if cur == 0:
        double, Lp = False, int(L)
    elif L > cur:
        double, Lp = True, cur
    else:
        return False
    C = _c(params["C"])
    N = C.shape[-1]
    dA_L = _power(Lp, _discrete_A(params))
    C_ = _conj_cat(C)
    prod = torch.einsum("hmn,chn->chm", dA_L.transpose(-1, -2), C_)
    if double:
        prod = -prod
    C_ = (C_ - prod)[..., :N]
    params["C"].copy_(torch.view_as_real(C_))
    params["L"].fill_(2 * cur if double else cur + Lp)
    return True


@torch.no_grad()
def ensure_length_(params, L):
    """The reference's lazy growth loop (s4.py:726-730): afterwards params['L'] >= L."""
    changed = False
    while L > int(params["L"].item()):
        changed |= setup_C_(params, L)
    return changed
