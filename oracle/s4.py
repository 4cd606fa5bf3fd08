"""TEST INFRASTRUCTURE (see oracle/__init__.py).

Restatement of the S4 (NPLR) layer as the active config uses it:
S4(d_model=H, d_state=64 -> 32 stored conjugate pairs, channels=1, rank 1,
unidirectional, activation gelu, postact glu, transposed, l_max=None).

Follows mug/model/s4.py:706-832 (SSKernelNPLR.forward), :557-584 (_setup_C),
:586-604 (_omega), :690-704 (_w), :140-147 (cauchy_naive, the backend that is
importable without pykeops / the un-vendored CUDA extension -- SURVEY.md D10)
and :1471-1541 (S4.forward).
"""
import numpy as np
import torch
import torch.nn.functional as F

_r2c = torch.view_as_complex


def _omega_ref(L):
    """s4.py:595-599: omega = exp(-2 pi i / L) ** arange(L//2+1) in complex64,
    z = 2 (1 - omega) / (1 + omega)."""
    omega = torch.tensor(np.exp(-2j * np.pi / L), dtype=torch.cfloat)
    omega = omega ** torch.arange(0, L // 2 + 1)
    z = 2 * (1 - omega) / (1 + omega)
    return omega, z


def s4_kernel(sd, p, L, mode="reference", symmetric=False):
    """Convolution kernel k (H, L) of the SSM whose parameters live under
    `p` = '<...>.s4_model.kernel.kernel'.

    mode="reference": the reference's arithmetic (complex64, omega by repeated
    power, Cauchy sum over the 32 stored poles only, rank-1 Woodbury,
    * 2/(1+omega), irfft(n=L_internal), truncate to L).
    mode="exact": the same formula evaluated in complex128 with exact FFT nodes
    and the Nyquist-safe factorisation (what the HIP kernel implements); used to
    judge which fp32 evaluation is closer to the real-number answer.

    Requires the stored internal length buffer L_int >= L and > 0 (the state a
    trained checkpoint is in); the length-doubling path is s4_setup_C below.
    """
    Lint = int(sd[p + ".L"])
    if Lint <= 0 or Lint < L:
        raise ValueError("S4 kernel at %s: stored L=%d < requested %d (run s4_setup_C first)" % (p, Lint, L))
    dt = torch.exp(sd[p + ".log_dt"])                          # (H,)
    Bc = _r2c(sd[p + ".B"].contiguous())                       # (1,H,N)
    Cc = _r2c(sd[p + ".C"].contiguous())                       # (1,H,N)
    Pc = _r2c(sd[p + ".P"].contiguous())                       # (1,H,N)
    w = -torch.exp(sd[p + ".inv_w_real"]) + 1j * sd[p + ".w_imag"]  # (H,N)  s4.py:690-704 real_type='exp'
    if mode == "reference":
        Qc = Pc.conj()
        omega, z = _omega_ref(Lint)
        wdt = w * dt[:, None]
        Bs = torch.cat([Bc, Pc], dim=-3)                       # (2,H,N)
        Cs = torch.cat([Cc, Qc], dim=-3)                       # (2,H,N)
        v = Bs.unsqueeze(-3) * Cs.unsqueeze(-4)                # (2,2,H,N)
        # cauchy_naive s4.py:140-147
        r = (v.unsqueeze(-1) / (z.unsqueeze(-2) - wdt.unsqueeze(-1))).sum(dim=-2)   # (2,2,H,Lf)
        if symmetric:
            # cauchy_conj (pykeops, s4.py:55-77): 2 sum_n (z Re v - Re(v conj w)) / ((z - w)(z - conj w)), the sum over BOTH
            # conjugate halves; written out exactly as the Genred expression, in complex64
            zz, ww, vv = z.unsqueeze(-2), wdt.unsqueeze(-1), v.unsqueeze(-1)
            num = zz * vv.real - (vv * ww.conj()).real
            r = 2 * (num / ((zz - ww) * (zz - ww.conj()))).sum(dim=-2)
        r = r * dt[None, None, :, None]
        k_f = r[:-1, :-1] - r[:-1, -1:] * r[-1:, :-1] / (1 + r[-1:, -1:])
        k_f = k_f * 2 / (1 + omega)
        k = torch.fft.irfft(k_f, n=Lint)[..., :L]
        return k[0, 0]                                          # (H,L)
    # exact mode (complex128, Nyquist-safe)
    dt = dt.double()
    Bc, Cc, Pc, w = Bc.to(torch.cdouble), Cc.to(torch.cdouble), Pc.to(torch.cdouble), w.to(torch.cdouble)
    kk = torch.arange(0, Lint // 2 + 1, dtype=torch.float64)
    omega = torch.exp(-2j * np.pi * kk / Lint)
    u = 1 + omega
    a = 2 * (1 - omega)
    wdt = w * dt[:, None]
    den = a[None, None, :] - wdt[:, :, None] * u[None, None, :]          # (H,N,Lf)
    den_c = a[None, None, :] - wdt.conj()[:, :, None] * u[None, None, :]
    def S(x, y):
        xy = (x * y)[0][:, :, None] * dt[:, None, None]
        r = (xy / den).sum(dim=1)                                                # (H,Lf)
        return r + (xy.conj() / den_c).sum(dim=1) if symmetric else r
    s00, s01, s10, s11 = S(Bc, Cc), S(Bc, Pc.conj()), S(Pc, Cc), S(Pc, Pc.conj())
    k_f = 2 * (s00 - u[None] * s01 * s10 / (1 + u[None] * s11))
    k = torch.fft.irfft(k_f, n=Lint)[..., :L]
    return k.float()


def s4_forward(sd, p, u, kernel_cache=None, mode="reference", symmetric=False):
    """S4.forward (s4.py:1471-1541): FFT long-conv with the generated kernel,
    + D*u, exact-erf GELU, Conv1d(H->2H,k=1) + GLU over channels.  u: (B,H,L)."""
    L = u.shape[-1]
    key = (p, L, mode, symmetric)
    if kernel_cache is not None and key in kernel_cache:
        k = kernel_cache[key]
    else:
        k = s4_kernel(sd, p + ".kernel.kernel", L, mode, symmetric=symmetric)
        if kernel_cache is not None:
            kernel_cache[key] = k
    k_f = torch.fft.rfft(k, n=2 * L)
    u_f = torch.fft.rfft(u, n=2 * L)
    y = torch.fft.irfft(u_f * k_f[None], n=2 * L)[..., :L]
    y = y + u * sd[p + ".D"][0][None, :, None]
    y = F.gelu(y)
    y = F.conv1d(y, sd[p + ".output_linear.0.weight"], sd[p + ".output_linear.0.bias"])
    return F.glu(y, dim=-2)


def s4_direct_conv(k, u, D):
    """Causal direct convolution y[t] = sum_{s<=t} k[s] u[t-s] + D u[t] in float64
    (the real-number definition the FFT conv approximates)."""
    B, H, L = u.shape
    kd, ud = k.double(), u.double()
    y = torch.zeros_like(ud)
    for s in range(L):
        y[..., s:] += kd[None, :, s:s + 1] * ud[..., :L - s]
    return y + ud * D.double()[None, :, None]
