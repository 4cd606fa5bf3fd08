"""Training: DDPM loss pieces, every block type's forward + backward and the whole-model step through the C ABI,
against torch autograd on the same fp32 math (GroupNorm eps 1e-6, SiLU, conv1d, Linear, smooth_l1(beta=0.02) + 0.01 --
mug/diffusion/unet.py:212-239, mug/diffusion/diffusion.py:326-354), an AdamW step against torch.optim.AdamW, and the
data-parallel gradient all-reduce (2-rank gloo) against the full-batch gradient."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.nn.functional as F
import torch.multiprocessing as mp


def rnd(seed, *shape, scale=1.0):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape).astype(np.float32)) * scale


def resblock_params(Cin, Cout, Kemb, seed=0, skip=None):
    skip = (Cin != Cout) if skip is None else skip
    p = {"in_layers.0.weight": 1 + 0.1 * rnd(seed + 1, Cin), "in_layers.0.bias": 0.1 * rnd(seed + 2, Cin),
         "in_layers.2.weight": rnd(seed + 3, Cout, Cin, 3, scale=(3 * Cin) ** -0.5), "in_layers.2.bias": 0.1 * rnd(seed + 4, Cout),
         "emb_layers.1.weight": rnd(seed + 5, Cout, Kemb, scale=Kemb ** -0.5), "emb_layers.1.bias": 0.1 * rnd(seed + 6, Cout),
         "out_layers.0.weight": 1 + 0.1 * rnd(seed + 7, Cout), "out_layers.0.bias": 0.1 * rnd(seed + 8, Cout),
         "out_layers.3.weight": rnd(seed + 9, Cout, Cout, 3, scale=(3 * Cout) ** -0.5), "out_layers.3.bias": 0.1 * rnd(seed + 10, Cout)}
    if skip:
        p["skip_connection.weight"] = rnd(seed + 11, Cout, Cin, 1, scale=Cin ** -0.5)
        p["skip_connection.bias"] = 0.1 * rnd(seed + 12, Cout)
    return p


def resblock_torch(p, x, emb, groups):
    """TimestepResBlock._forward (unet.py:212-239) in torch functional ops."""
    h = F.conv1d(F.silu(F.group_norm(x, groups, p["in_layers.0.weight"], p["in_layers.0.bias"], eps=1e-6)),
                 p["in_layers.2.weight"], p["in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), p["emb_layers.1.weight"], p["emb_layers.1.bias"])
    h = h + e[..., None]
    h = F.conv1d(F.silu(F.group_norm(h, groups, p["out_layers.0.weight"], p["out_layers.0.bias"], eps=1e-6)),
                 p["out_layers.3.weight"], p["out_layers.3.bias"], padding=1)
    skip = F.conv1d(x, p["skip_connection.weight"], p["skip_connection.bias"]) if "skip_connection.weight" in p else x
    return skip + h


def close(a, b, tol, what):
    a, b = a.detach().cpu(), b.detach().cpu()
    d, s = (a - b).abs().max().item(), b.abs().max().item()
    assert a.shape == b.shape and d <= tol * max(1.0, s), "%s: max|diff| %.3e (ref max %.3e)" % (what, d, s)


@pytest.mark.parametrize("B,Cin,Cout,T,Kemb,groups", [(2, 32, 32, 40, 48, 8), (3, 32, 64, 64, 64, 16), (1, 64, 32, 20, 32, 32), (2, 128, 128, 96, 512, 32), (4, 256, 512, 64, 512, 32)])
def test_resblock_forward_backward_vs_autograd(lib, B, Cin, Cout, T, Kemb, groups):
    p = resblock_params(Cin, Cout, Kemb)
    x, emb, dy = rnd(20, B, Cin, T), rnd(21, B, Kemb), rnd(22, B, Cout, T)
    pt = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xt, et = x.clone().requires_grad_(True), emb.clone().requires_grad_(True)
    yt = resblock_torch(pt, xt, et, groups)
    yt.backward(dy)
    y, dx, demb, grads = lib.train_resblock(p, x, emb, dy, groups=groups)
    close(y, yt, 2e-5, "forward")
    close(dx, xt.grad, 5e-5, "dx")
    close(demb, et.grad, 5e-5, "demb")
    assert set(grads) == set(p)
    for k in p:
        close(grads[k], pt[k].grad, 1e-4, "grad of " + k)


@pytest.mark.parametrize("B,Cin,Cout,T,taps,dil,mode,gn", [(2, 32, 48, 40, 3, 1, 0, False), (2, 32, 32, 64, 3, 4, 0, False), (1, 48, 32, 50, 3, 8, 0, False),
                                                           (2, 32, 64, 36, 1, 1, 0, False), (2, 32, 48, 40, 3, 1, 1, False), (2, 48, 32, 26, 3, 1, 2, False),
                                                           (2, 64, 16, 40, 3, 1, 0, True), (2, 32, 32, 1024, 3, 2, 0, False), (2, 32, 32, 1024, 3, 1, 1, False),
                                                           (1, 32, 16, 2200, 3, 1, 0, True)])      # the last: GroupNorm rows long enough for the 4-granule batches
def test_conv_layer_forward_backward_vs_autograd(lib, B, Cin, Cout, T, taps, dil, mode, gn):
    """Plain / dilated conv (ResnetBlock, models.py:106-122), Downsample (pad right + stride 2, models.py:84-88), Upsample (nearest x2 +
    conv, models.py:66-70) and the GroupNorm + SiLU + conv head (unet.py:489-493): forward, dx, dw, db vs torch autograd."""
    w, b = rnd(50, Cout, Cin, taps, scale=(taps * Cin) ** -0.5), 0.1 * rnd(51, Cout)
    x = rnd(52, B, Cin, T)
    gw, gb = 1 + 0.1 * rnd(53, Cin), 0.1 * rnd(54, Cin)
    wt, bt, xt, gwt, gbt = [v.clone().requires_grad_(True) for v in (w, b, x, gw, gb)]
    a = F.silu(F.group_norm(xt, 8, gwt, gbt, eps=1e-6)) if gn else xt
    if mode == 0:
        yt = F.conv1d(a, wt, bt, padding=dil * (taps - 1) // 2, dilation=dil)
    elif mode == 1:
        yt = F.conv1d(F.pad(a, (0, 1)), wt, bt, stride=2)
    else:
        yt = F.conv1d(a.repeat_interleave(2, dim=-1), wt, bt, padding=1)
    dy = rnd(55, *yt.shape)
    yt.backward(dy)
    y, dx, dw, db, dg = lib.train_conv(w, b, x, dy, dil=dil, mode=mode, gn=(gw, gb) if gn else None, groups=8)
    close(y, yt, 2e-5, "forward")
    close(dx, xt.grad, 5e-5, "dx")
    close(dw, wt.grad, 1e-4, "dw")
    close(db, bt.grad, 1e-4, "db")
    if gn:
        close(dg[0], gwt.grad, 1e-4, "d gamma")
        close(dg[1], gbt.grad, 1e-4, "d beta")


def transformer_params(C, Cc, heads, pmax=64, seed=100):
    """A ContextualTransformer's tensors (mug/model/attention.py:154-199) with the reference's shapes; the zero-initialised ones
    (proj_out, relative_position_embedding) and the ones-initialised C_embedding are randomised so every gradient is exercised."""
    from mug._native import Lib
    shapes = {"norm.weight": (C,), "norm.bias": (C,), "proj_in.weight": (C, C, 1), "proj_in.bias": (C,), "proj_out.weight": (C, C, 1), "proj_out.bias": (C,)}
    b = "transformer_blocks.0."
    for i, kc in ((1, C), (2, Cc)):
        shapes[b + "norm%d.weight" % i] = (C,); shapes[b + "norm%d.bias" % i] = (C,)
        shapes[b + "attn%d.to_q.weight" % i] = (C, C); shapes[b + "attn%d.to_k.weight" % i] = (C, kc); shapes[b + "attn%d.to_v.weight" % i] = (C, kc)
        shapes[b + "attn%d.to_out.0.weight" % i] = (C, C); shapes[b + "attn%d.to_out.0.bias" % i] = (C,)
        shapes[b + "attn%d.relative_position_embedding" % i] = (2 * pmax + 1, heads); shapes[b + "attn%d.C_embedding" % i] = (2 * pmax + 1, heads)
    shapes[b + "norm3.weight"] = (C,); shapes[b + "norm3.bias"] = (C,)
    shapes[b + "ff.net.0.proj.weight"] = (8 * C, C); shapes[b + "ff.net.0.proj.bias"] = (8 * C,)
    shapes[b + "ff.net.2.weight"] = (C, 4 * C); shapes[b + "ff.net.2.bias"] = (C,)
    assert set(shapes) == set(Lib.TRANSFORMER_KEYS)
    p = {}
    for i, k in enumerate(Lib.TRANSFORMER_KEYS):
        sh = shapes[k]
        if k.endswith("norm.weight") or "norm1.weight" in k or "norm2.weight" in k or "norm3.weight" in k:
            p[k] = 1 + 0.1 * rnd(seed + i, *sh)
        elif k.endswith("C_embedding"):
            p[k] = 1 + 0.3 * rnd(seed + i, *sh)
        elif k.endswith("relative_position_embedding"):
            p[k] = 0.5 * rnd(seed + i, *sh)
        elif k.endswith(".bias"):
            p[k] = 0.1 * rnd(seed + i, *sh)
        else:
            p[k] = rnd(seed + i, *sh, scale=sh[1] ** -0.5)
    return p


@pytest.mark.parametrize("B,C,T,Cc,Tk,heads,groups,pmax", [(2, 32, 24, 16, 5, 2, 8, 4), (1, 64, 40, 0, 0, 4, 32, 8), (2, 64, 96, 32, 21, 4, 16, 64),
                                                           (2, 128, 64, 128, 21, 8, 32, 64)])
def test_transformer_forward_backward_vs_autograd(lib, B, C, T, Cc, Tk, heads, groups, pmax):
    """ContextualTransformer (GroupNorm -> proj_in -> [LN, rel-pos self-attention, LN, cross-attention over the prompt tokens, LN,
    GEGLU feed-forward] -> proj_out -> + x): native forward + backward against torch autograd through the oracle's restatement
    (oracle/nets.py:contextual_transformer, itself bit-equal to the reference on the goldens).  Cc = 0: attn2 is a second
    self-attention (the wave encoder's blocks)."""
    from oracle import nets
    p = transformer_params(C, Cc if Cc else C, heads, pmax)
    x, dy = rnd(40, B, C, T), rnd(41, B, C, T)
    ctx = rnd(42, B, Cc, Tk) if Cc else None
    pt = {"m." + k: v.clone().requires_grad_(True) for k, v in p.items()}
    xt = x.clone().requires_grad_(True)
    ct = None if ctx is None else ctx.clone().requires_grad_(True)
    real_gn, real_ca = nets.group_norm, nets.cross_attention
    nets.group_norm = lambda sd, pp, xx, g: real_gn(sd, pp, xx, groups)      # the oracle hard-codes 32 groups / pos_max 64: small shapes
    nets.cross_attention = lambda sd, pp, xx, cc, hh, pos_max=64: real_ca(sd, pp, xx, cc, hh, pos_max=pmax)
    try:
        yt = nets.contextual_transformer(pt, "m", xt, ct, heads)
    finally:
        nets.group_norm, nets.cross_attention = real_gn, real_ca
    yt.backward(dy)
    y, dx, dctx, grads = lib.train_transformer(p, x, ctx, dy, heads, groups=groups)
    close(y, yt, 5e-5, "forward")
    close(dx, xt.grad, 2e-4, "dx")
    if ctx is not None:
        close(dctx, ct.grad, 2e-4, "dcontext")
    for k in p:
        close(grads[k], pt["m." + k].grad, 3e-4, "grad of " + k)


@pytest.mark.parametrize("B,C,T,Cc,Tk,heads,groups,pmax", [(2, 64, 96, 32, 21, 4, 16, 64), (1, 64, 48, 0, 0, 4, 32, 8),
                                                           # the matrix-core attention backward at every head dim it is built for (32 / 48 / 64), one and
                                                           # two column tiles per wave (T = 64 ... 256), relative offsets beyond the clamp
                                                           (1, 128, 256, 0, 0, 4, 32, 64), (1, 192, 128, 32, 21, 4, 32, 16), (2, 128, 64, 0, 0, 2, 32, 64)])
def test_transformer_bf16_mode_vs_fp32_mode(lib, B, C, T, Cc, Tk, heads, groups, pmax):
    """The transformer block with its GEMMs on the bf16 matrix cores -- the Linears AND the key-side attention gradients (two batched
    matmuls with per-(batch row, head) operands: dk = dsim^T q, dv = A^T dO) and, where T is a multiple of 32, the attention backward's
    row kernel (S = q^T k, dA = dO^T v, dq = dsim k^T as MFMA tiles) -- against the fp32 mode (which equals autograd, above):
    every output within bf16 rounding of the operands (relative L2 error <= 2 %)."""
    p = transformer_params(C, Cc if Cc else C, heads, pmax)
    x, dy = rnd(40, B, C, T), rnd(41, B, C, T)
    ctx = rnd(42, B, Cc, Tk) if Cc else None
    y32, dx32, dc32, g32 = lib.train_transformer(p, x, ctx, dy, heads, groups=groups)
    lib.train_set_precision(True)
    try:
        y16, dx16, dc16, g16 = lib.train_transformer(p, x, ctx, dy, heads, groups=groups)
    finally:
        lib.train_set_precision(False)

    def rel(a, b):
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        return float((a - b).norm() / b.norm())
    worst = max([("forward", rel(y16, y32)), ("dx", rel(dx16, dx32))] + ([("dcontext", rel(dc16, dc32))] if ctx is not None else [])
                + [(k, rel(g16[k], g32[k])) for k in p], key=lambda kv: kv[1])
    print("bf16 transformer block vs fp32: worst relative L2 error %.2e (%s)" % (worst[1], worst[0]))
    assert worst[1] <= 2e-2, worst


def s4layer_params(H, N, L, seed=200):
    """An S4Layer's tensors with the reference's shapes (state-dict names relative to the layer) in a trained-checkpoint-like
    state: stored kernel length L, poles spread along the imaginary axis, dt in [1e-3, 1e-1]."""
    rng = np.random.default_rng(seed)
    p = {"norm.weight": 1 + 0.1 * rnd(seed + 1, H), "norm.bias": 0.1 * rnd(seed + 2, H),
         "s4_model.kernel.kernel.C": rnd(seed + 3, 1, H, N, 2, scale=0.7), "s4_model.kernel.kernel.B": rnd(seed + 4, 1, H, N, 2, scale=0.7),
         "s4_model.kernel.kernel.P": rnd(seed + 5, 1, H, N, 2, scale=0.3),
         "s4_model.kernel.kernel.inv_w_real": torch.from_numpy(np.log(0.5 + 0.2 * rng.random((H, N))).astype(np.float32)),
         "s4_model.kernel.kernel.w_imag": torch.from_numpy((np.pi * np.arange(N)[None] + 0.3 * rng.standard_normal((H, N))).astype(np.float32)),
         "s4_model.kernel.kernel.log_dt": torch.from_numpy(rng.uniform(np.log(1e-3), np.log(1e-1), H).astype(np.float32)),
         "s4_model.D": rnd(seed + 6, 1, H),
         "s4_model.output_linear.0.weight": rnd(seed + 7, 2 * H, H, 1, scale=H ** -0.5), "s4_model.output_linear.0.bias": 0.1 * rnd(seed + 8, 2 * H),
         "out_layer.weight": rnd(seed + 9, H, H, 3, scale=(3 * H) ** -0.5), "out_layer.bias": 0.1 * rnd(seed + 10, H)}
    p["s4_model.kernel.kernel.L"] = torch.tensor(L, dtype=torch.int64)
    return p


@pytest.mark.parametrize("B,Cin,Cout,T,Kemb,groups", [(2, 32, 32, 40, 48, 8), (3, 32, 64, 64, 64, 16), (2, 128, 128, 96, 512, 32), (1, 64, 32, 132, 32, 32)])
def test_resblock_bf16_activation_storage_changes_nothing(lib, monkeypatch, B, Cin, Cout, T, Kemb, groups):
    """bf16 mode stores a ResBlock's two normalised activations (the conv / weight-gradient operands) as bfloat16 -- the GEMMs round them at
    staging anyway: forward, input / embedding gradients and every parameter gradient are BIT-IDENTICAL to the fp32-stored arm
    (MUGD_TRAIN_ACT_FP32=1), with and without the time embedding, with the block's intermediates kept between the two calls or not."""
    p = resblock_params(Cin, Cout, Kemb)
    x, emb, dy = rnd(20, B, Cin, T), rnd(21, B, Kemb), rnd(22, B, Cout, T)

    def run(fp32_store, keep):
        monkeypatch.setenv("MUGD_TRAIN_ACT_FP32", "1" if fp32_store else "0")
        if keep:
            from mug._native import TrainState
            st = TrainState()
            y = lib.train_resblock(p, x, emb, None, groups=groups, state=st)[0]
            _, dx, demb, g = lib.train_resblock(p, x, emb, dy, groups=groups, state=st)
        else:
            y, dx, demb, g = lib.train_resblock(p, x, emb, dy, groups=groups)
        return [y, dx, demb] + [g[k] for k in sorted(g)]

    lib.train_set_precision(True)
    try:
        for keep in (False, True):
            a, b = run(False, keep), run(True, keep)
            for i, (u, v) in enumerate(zip(a, b)):
                assert torch.equal(u.cpu(), v.cpu()), (keep, i, float((u.cpu() - v.cpu()).abs().max()))
    finally:
        lib.train_set_precision(False)



@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("B,H,T,N,Lint,groups", [(2, 32, 32, 8, 32, 8), (2, 32, 48, 32, 64, 16), (1, 64, 128, 32, 128, 32),
                                                 # the Toeplitz-GEMM long convolution at its largest length (16 block diagonals) and over two
                                                 # batch tiles with a ragged second one
                                                 (3, 16, 512, 8, 512, 8), (34, 16, 64, 8, 64, 8)])
def test_s4layer_forward_backward_vs_autograd(lib, B, H, T, N, Lint, groups, symmetric):
    """S4Layer (unet.py:76-91): GroupNorm -> NPLR kernel (s4.py:706-832) -> causal long conv + D u -> GELU -> Conv1d(H->2H) + GLU ->
    conv3 -> + x.  Native forward + backward, INCLUDING the kernel generator's parameter gradients (C, B, P, inv_w_real, w_imag,
    log_dt), against torch autograd through the oracle's restatement in its float64 'exact' mode (the same Nyquist-safe formula the
    device evaluates; the complex64 'reference' mode differs from it by fp32 rounding only: tests/test_ops.py::test_s4_kernel).
    symmetric: the Cauchy sum over both conjugate halves (the reference's pykeops / CUDA-extension backends, s4.py:55-77;
    mugd_set_s4_symmetric) -- a checkpoint trained with those backends can be fine-tuned."""
    from oracle import s4 as s4o
    if symmetric and (B, H) != (2, 32):
        pytest.skip("one shape per Cauchy form is enough on the emulator")
    p = s4layer_params(H, N, Lint)
    x, dy = rnd(60, B, H, T), rnd(61, B, H, T)
    pt = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 else v) for k, v in p.items()}
    xt = x.clone().requires_grad_(True)
    u = F.group_norm(xt, groups, pt["norm.weight"], pt["norm.bias"], eps=1e-6)
    yt = xt + F.conv1d(s4o.s4_forward(pt, "s4_model", u, None, mode="exact", symmetric=symmetric), pt["out_layer.weight"], pt["out_layer.bias"], padding=1)
    yt.backward(dy)
    lib.set_s4_symmetric(symmetric)
    try:
        y, dx, grads = lib.train_s4layer(p, x, dy, groups=groups)
    finally:
        lib.set_s4_symmetric(False)
    close(y, yt, 1e-4, "forward")
    close(dx, xt.grad, 3e-4, "dx")
    for k in grads:
        close(grads[k], pt[k].grad, 5e-4, "grad of " + k)


@pytest.mark.gpu
def test_shipped_size_blocks_vs_autograd(gpu_lib):
    """The three block types at the SHIPPED model's deepest-level shapes (C = 512, T = 64, 8 heads of 64, 21 prompt tokens of 128
    channels, 32 S4 poles, batch 4) against torch autograd on the host: forward, input gradients and every parameter gradient."""
    from oracle import nets, s4 as s4o
    lib = gpu_lib
    B, Cm, T = 4, 512, 64
    # TimestepResBlock 1536 -> 512 (the first up-block after the skip concatenation)
    p = resblock_params(1536, Cm, 512)
    x, emb, dy = rnd(20, B, 1536, T), rnd(21, B, 512), rnd(22, B, Cm, T)
    pt = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xt, et = x.clone().requires_grad_(True), emb.clone().requires_grad_(True)
    yt = resblock_torch(pt, xt, et, 32)
    yt.backward(dy)
    y, dx, demb, grads = lib.train_resblock(p, x, emb, dy, groups=32)
    close(y, yt, 5e-5, "resblock forward")
    close(dx, xt.grad, 1e-4, "resblock dx")
    close(demb, et.grad, 1e-4, "resblock demb")
    for k in p:
        close(grads[k], pt[k].grad, 2e-4, "resblock grad of " + k)
    # ContextualTransformer with the prompt context
    p = transformer_params(Cm, 128, 8, 64)
    x, dy, ctx = rnd(40, B, Cm, T), rnd(41, B, Cm, T), rnd(42, B, 128, 21)
    pt = {"m." + k: v.clone().requires_grad_(True) for k, v in p.items()}
    xt, ct = x.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
    yt = nets.contextual_transformer(pt, "m", xt, ct, 8)
    yt.backward(dy)
    y, dx, dctx, grads = lib.train_transformer(p, x, ctx, dy, 8, groups=32)
    close(y, yt, 1e-4, "transformer forward")
    close(dx, xt.grad, 3e-4, "transformer dx")
    close(dctx, ct.grad, 3e-4, "transformer dcontext")
    for k in p:
        close(grads[k], pt["m." + k].grad, 5e-4, "transformer grad of " + k)
    # S4Layer
    p = s4layer_params(Cm, 32, T)
    x, dy = rnd(60, B, Cm, T), rnd(61, B, Cm, T)
    pt = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 else v) for k, v in p.items()}
    xt = x.clone().requires_grad_(True)
    u = F.group_norm(xt, 32, pt["norm.weight"], pt["norm.bias"], eps=1e-6)
    yt = xt + F.conv1d(s4o.s4_forward(pt, "s4_model", u, None, mode="exact"), pt["out_layer.weight"], pt["out_layer.bias"], padding=1)
    yt.backward(dy)
    y, dx, grads = lib.train_s4layer(p, x, dy, groups=32)
    close(y, yt, 1e-4, "s4 layer forward")
    close(dx, xt.grad, 3e-4, "s4 layer dx")
    for k in grads:
        close(grads[k], pt[k].grad, 1e-3, "s4 layer grad of " + k)


def test_training_entry_points_reject_bad_arguments(lib):
    """Error behaviour of the training ABI: bad geometry / missing tensors come back as MugdError (a negative status with a message),
    never as a crash; the context stays usable."""
    from mug._native import MugdError
    p = resblock_params(32, 32, 16)
    x, emb, dy = rnd(1, 2, 32, 16), rnd(2, 2, 16), rnd(3, 2, 32, 16)
    with pytest.raises(MugdError):
        lib.train_resblock(p, x, emb, dy, groups=5)                      # channels not divisible by the group count
    with pytest.raises((MugdError, KeyError)):
        lib.train_resblock({k: v for k, v in p.items() if k != "emb_layers.1.weight"}, x, emb, dy, groups=8)
    with pytest.raises(MugdError):
        lib.train_conv(rnd(4, 32, 32, 3), None, rnd(5, 2, 32, 15), rnd(6, 2, 32, 7), mode=1)      # Downsample needs an even length
    sp = s4layer_params(32, 8, 16)
    with pytest.raises(MugdError):
        lib.train_s4layer(sp, rnd(7, 1, 32, 32), rnd(8, 1, 32, 32), groups=8)                      # stored kernel length 16 < T = 32
    y, dx, demb, g = lib.train_resblock(p, x, emb, dy, groups=8)        # still works
    assert torch.isfinite(y).all() and torch.isfinite(dx).all()


def test_block_state_keeps_forward_intermediates(lib):
    """The `state` argument of the block entry points (include/mugd.h): forward-only call keeps the intermediates, the backward call
    with the same handle skips the forward and consumes them (bit-identical to the recomputing call); a consumed handle is rejected."""
    from mug._native import MugdError, TrainState
    p = resblock_params(32, 64, 16)
    x, emb, dy = rnd(1, 2, 32, 24), rnd(2, 2, 16), rnd(3, 2, 64, 24)
    y0, dx0, de0, g0 = lib.train_resblock(p, x, emb, dy, groups=8)
    st = TrainState()
    y1 = lib.train_resblock(p, x, emb, None, groups=8, state=st)[0]
    assert st.v.value != 0 and torch.equal(y0, y1)
    _, dx1, de1, g1 = lib.train_resblock(p, x, emb, dy, groups=8, state=st)
    assert st.v.value == 0 and torch.equal(dx0, dx1) and torch.equal(de0, de1) and all(torch.equal(g0[k], g1[k]) for k in g0)
    st.v.value = 12345
    with pytest.raises(MugdError):
        lib.train_resblock(p, x, emb, dy, groups=8, state=st)
    st2 = TrainState()
    lib.train_resblock(p, x, emb, None, groups=8, state=st2)
    lib.train_release_states()                                  # never-consumed handles are dropped
    with pytest.raises(MugdError):
        lib.train_resblock(p, x, emb, dy, groups=8, state=st2)


def test_q_sample_and_smooth_l1_loss(lib):
    """diffusion.py:326-354,386: x_t = sqrt(ac_t) x0 + sqrt(1 - ac_t) noise;  loss_b = mean smooth_l1(target, pred, beta=0.02) + 0.01."""
    B, Cc, T = 3, 16, 50
    betas = np.linspace(1e-4, 2e-2, 1000, dtype=np.float64)
    ac = torch.from_numpy(np.cumprod(1 - betas)).float()
    x0, noise = rnd(1, B, Cc, T), rnd(2, B, Cc, T)
    t = torch.tensor([0, 481, 999])
    xt = lib.train_q_sample(x0, noise, t, ac.sqrt(), (1 - ac).sqrt()).cpu()
    ref = ac.sqrt()[t][:, None, None] * x0 + (1 - ac).sqrt()[t][:, None, None] * noise
    assert torch.equal(xt, ref) or (xt - ref).abs().max().item() < 1e-6          # one fma per element: contraction may differ in the last bit
    pred = (noise + 0.03 * rnd(3, B, Cc, T)).requires_grad_(True)                # differences on both sides of beta
    lt = (F.smooth_l1_loss(noise, pred, beta=0.02, reduction="none") + 0.01).mean(dim=[1, 2])
    lt.mean().backward()
    loss, grad = lib.train_smooth_l1(pred.detach(), noise)
    close(loss, lt, 1e-6, "per-sample loss")
    close(grad, pred.grad, 1e-6, "loss gradient")


def test_adamw_step_matches_torch(lib):
    n = 1000
    p0, g1, g2 = rnd(5, n), rnd(6, n), rnd(7, n)
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
    dev = lib.device
    p, m, v = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for step, g in enumerate((g1, g2), start=1):
        pt.grad = g.clone()
        opt.step()
        lib.train_adamw(p, g.to(dev), m, v, step, lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05)
        close(p, pt, 2e-6, "parameters after step %d" % step)


def test_step_bracket_changes_nothing_over_two_steps(lib, monkeypatch):
    """TrainPlan.step inside the library's step bracket (packed bf16 weights from the cache -- refreshed by ONE table launch from the
    second step on -- and all split-K / bias-row reductions as ONE table launch) against the same two steps with every call packing
    and reducing on its own (MUGD_NO_STEP_BRACKET=1): loss and every gradient bit-identical, with an AdamW update between the steps so
    that the second step's cache refresh has new values to pick up."""
    from oracle import cases
    from mug import train
    case, z, B = cases.TINY, 32, 2
    g, sd0, seed, batch = _load_train_fixture(case, z, B)
    t, noise = torch.from_numpy(g["t"]), torch.from_numpy(g["noise"])
    dev = lib.device

    def two_steps(no_bracket):
        monkeypatch.setenv("MUGD_NO_STEP_BRACKET", "1" if no_bracket else "0")
        sd = {k: (v.clone().to(dev) if torch.is_tensor(v) else v) for k, v in sd0.items()}
        x0 = train.encode_x0(lib, sd, case["vae"], batch["note"])
        plan = train.TrainPlan(lib, sd, case["unet"], case["wave"])
        assert plan.bracket
        ids, mel = batch["feature"].long().to(dev), batch["audio"].to(dev)
        loss1, grads = plan.step(x0, noise.to(dev), t.to(dev), ids, mel)
        opt = train.AdamW(lib, sd, grads, lr=1e-3)
        opt.step()
        loss2, grads = plan.step(x0, noise.to(dev), t.to(dev), ids, mel)
        return float(loss1), float(loss2), {k: v.clone().cpu() for k, v in grads.items()}

    lib.train_set_precision(True)
    try:
        a1, a2, ga = two_steps(False)
        b1, b2, gb = two_steps(True)
    finally:
        lib.train_set_precision(False)
    assert a1 == b1 and a2 == b2 and a1 != a2, (a1, b1, a2, b2)
    assert set(ga) == set(gb)
    bad = [k for k in ga if not torch.equal(ga[k], gb[k])]
    assert not bad, bad[:5]


def test_step_bracket_entry_points_tolerate_any_call_order(lib):
    """mugd_train_step_begin / _flush / _end in odd orders (end or flush without begin, begin twice, an empty bracket) succeed and leave the
    block entry points working -- a conv layer's results inside a bracket opened twice equal the unbracketed ones, and a bracket that is
    never closed is closed by the next begin (its queued reductions run then)."""
    lib.train_step_end()
    lib.train_step_flush()
    lib.train_step_begin()
    lib.train_step_begin()
    lib.train_step_end()
    # resident tensors: a weight used inside a bracket must outlive it (include/mugd.h) -- no per-call staging copies
    w, b, x, dy = (v.to(lib.device) for v in (rnd(1, 48, 32, 3, scale=0.1), rnd(2, 48), rnd(3, 2, 32, 40), rnd(4, 2, 48, 40)))
    lib.train_set_precision(True)
    try:
        want = lib.train_conv(w, b, x, dy)
        lib.train_step_begin()
        got = lib.train_conv(w, b, x, dy)                 # weight / bias gradients are only queued here ...
        lib.train_step_begin()                           # ... and reduced by the next begin (an abandoned step's leftovers)
        lib.train_step_end()
        for u, v in zip(want[:4], got[:4]):
            assert torch.equal(u.cpu(), v.cpu())
    finally:
        lib.train_set_precision(False)


def test_fused_adamw_over_a_tensor_list_matches_torch(lib):
    """mug.train.AdamW: the whole parameter list in ONE launch (mugd_train_adamw_chunks: tensors cut into runs of <= 4096 elements) against
    torch.optim.AdamW, three steps, tensors below / at / above the run length and of odd sizes."""
    from mug import train
    dev = lib.device
    shapes = [(7,), (4096,), (4097,), (3, 5000), (64, 16, 3)]
    ps = [rnd(300 + i, *sh) for i, sh in enumerate(shapes)]
    pts = [p.clone().requires_grad_(True) for p in ps]
    opt_t = torch.optim.AdamW(pts, lr=2e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.02)
    params = {"p%d" % i: p.clone().to(dev) for i, p in enumerate(ps)}
    grads = {k: torch.zeros_like(v) for k, v in params.items()}
    opt = train.AdamW(lib, params, grads, lr=2e-3, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.02)
    for step in range(3):
        for i, pt in enumerate(pts):
            g = rnd(400 + 10 * step + i, *shapes[i])
            pt.grad = g.clone()
            grads["p%d" % i].copy_(g.to(dev))
        opt_t.step()
        opt.step()
        for i, pt in enumerate(pts):
            close(params["p%d" % i], pt, 2e-6, "tensor %d after step %d" % (i, step + 1))


def test_whole_model_training_step_vs_autograd(lib):
    """configs[4] in miniature: DDPM.p_losses (diffusion.py:356-414) on the structurally complete `tiny` model -- q_sample, prompt
    embedding, wave encoder, U-Net (ResBlocks, transformers with cross-attention, S4 layers, down / upsampling, audio and skip
    concatenations), smooth-L1 loss -- through mug.train.training_step (native block forward / backward entry points, block-level
    checkpointing), against torch autograd through the oracle's restatement of the same networks: the loss and the gradient of
    EVERY trainable tensor of the three networks."""
    from oracle import cases, nets, weights
    from mug import train
    case = cases.TINY
    man = weights.load_manifest(os.path.join(cases.GOLDEN, case["manifest"]))
    B, z = 2, 32
    sd = weights.set_s4_lengths(weights.make_state_dict(man, 3), case["unet"], z)
    x0, noise = rnd(70, B, 16, z), rnd(71, B, 16, z)
    t = torch.tensor([17, 803])
    ids = cases.prompt_ids(case, 5, B) if hasattr(cases, "prompt_ids") else torch.from_numpy(np.random.default_rng(5).integers(0, sd["model.cond_stage_model.embedding.weight"].shape[0], (B, case["n_ctx_tok"])))
    mel = rnd(72, B, case["wave"]["n_freq"], z * case["audio_ratio"]).abs()
    trainable = [k for k, v in sd.items() if v.dtype == torch.float32 and k.startswith("model.") and not k.startswith("model.first_stage_model")]
    st = {k: (v.clone().requires_grad_(True) if k in trainable else v) for k, v in sd.items()}
    xt = st["sqrt_alphas_cumprod"][t][:, None, None] * x0 + st["sqrt_one_minus_alphas_cumprod"][t][:, None, None] * noise
    ctx = nets.cond_embed(st, ids)
    w = nets.wave_encode(st, case["wave"], mel)
    pred = nets.unet_forward(st, case["unet"], xt, t, ctx, w)
    lt = (F.smooth_l1_loss(noise, pred, beta=0.02, reduction="none") + 0.01).mean(dim=[1, 2]).mean()
    lt.backward()
    loss, grads = train.training_step(lib, sd, case["unet"], case["wave"], x0, noise, t, ids, mel)
    loss_r, grads_r = train.training_step(lib, sd, case["unet"], case["wave"], x0, noise, t, ids, mel, recompute=True)
    assert float(loss) == float(loss_r) and set(grads) == set(grads_r) and all(torch.equal(grads[k], grads_r[k]) for k in grads), \
        "kept-intermediates and recompute (checkpointing) sweeps must give identical gradients"
    lt = lt.detach()
    assert abs(float(loss) - float(lt)) <= 1e-5 * max(1.0, abs(float(lt))), (float(loss), float(lt))
    missing = [k for k in trainable if st[k].grad is not None and k not in grads]
    assert not missing, "no gradient produced for %d tensors, e.g. %s" % (len(missing), missing[:5])
    worst = (0.0, None)
    for k in trainable:
        if st[k].grad is None:
            assert k not in grads or float(grads[k].abs().max()) == 0.0, k         # tensors the loss does not reach (wave levels the U-Net does not read)
            continue
        a, b = grads[k].detach().cpu().reshape(st[k].shape), st[k].grad
        d, sc = (a - b).abs().max().item(), b.abs().max().item()
        rel = d / max(sc, 1e-6)
        worst = max(worst, (rel, k))
        assert d <= 2e-3 * max(sc, 1e-4), "gradient of %s: max|diff| %.3e (ref max %.3e)" % (k, d, sc)
    print("worst relative gradient error %.2e at %s over %d tensors" % (worst[0], worst[1], len(trainable)))


def _tiny_training_inputs(B, z):
    from oracle import cases, weights
    case = cases.TINY
    man = weights.load_manifest(os.path.join(cases.GOLDEN, case["manifest"]))
    sd = weights.set_s4_lengths(weights.make_state_dict(man, 3), case["unet"], z)
    x0, noise = rnd(70, B, 16, z), rnd(71, B, 16, z)
    t = torch.tensor([17, 803, 402, 999][:B])
    ids = torch.from_numpy(np.random.default_rng(5).integers(0, sd["model.cond_stage_model.embedding.weight"].shape[0], (B, case["n_ctx_tok"])))
    mel = rnd(72, B, case["wave"]["n_freq"], z * case["audio_ratio"]).abs()
    return case, sd, x0, noise, t, ids, mel


def _step_worker(rank, world, port, q):
    import torch.distributed as dist
    from conftest import emu_lib
    from mug import train
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = emu_lib()
        case, sd, x0, noise, t, ids, mel = _tiny_training_inputs(2, 32)
        sl = slice(rank, rank + 1)
        red = train.BucketedAllReduce(bucket_bytes=64 << 10)         # ... vs small buckets overlapped with the backward sweep
        loss, grads = train.training_step(lib, sd, case["unet"], case["wave"], x0[sl], noise[sl], t[sl], ids[sl], mel[sl], reducer=red)
        assert red.n_buckets > 3
        loss2, grads2 = train.training_step(lib, sd, case["unet"], case["wave"], x0[sl], noise[sl], t[sl], ids[sl], mel[sl])
        train.allreduce_gradients(grads2, average=True)
        assert set(grads) == set(grads2) and all(torch.equal(grads[k], grads2[k]) for k in grads), "bucketed / overlapped reduction differs"
        # the same in bf16 mode: weight-gradient split-K slices are still queued in the library's reduction table when a bucket fills --
        # the reducer flushes the table before it reads the bucket's tensors
        lib.train_set_precision(True)
        try:
            red16 = train.BucketedAllReduce(bucket_bytes=64 << 10)
            _, g16 = train.training_step(lib, sd, case["unet"], case["wave"], x0[sl], noise[sl], t[sl], ids[sl], mel[sl], reducer=red16)
            _, g16b = train.training_step(lib, sd, case["unet"], case["wave"], x0[sl], noise[sl], t[sl], ids[sl], mel[sl])
            train.allreduce_gradients(g16b, average=True)
            assert red16.n_buckets > 3 and all(torch.equal(g16[k], g16b[k]) for k in g16), "bf16 mode: bucketed / overlapped reduction differs"
        finally:
            lib.train_set_precision(False)
        q.put((rank, float(loss), {k: v.cpu().numpy() for k, v in grads.items()}))
    finally:
        dist.destroy_process_group()


def test_whole_model_data_parallel_step_equals_full_batch():
    """configs[4]'s shape in miniature: 2 ranks, one sample each, whole-model training_step on the emulated build, ONE bucketed
    all_reduce over all 395 gradient tensors (gloo here, RCCL on the GPU box) = the single-rank full-batch gradients; then one AdamW
    step on every tensor."""
    from conftest import emu_lib
    from mug import train
    lib = emu_lib()
    case, sd, x0, noise, t, ids, mel = _tiny_training_inputs(2, 32)
    loss, want = train.training_step(lib, sd, case["unet"], case["wave"], x0, noise, t, ids, mel)
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_step_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = [q.get(timeout=900) for _ in procs]
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    assert abs(sum(g[1] for g in got) / world - float(loss)) < 1e-5
    for rank, _, grads in got:
        assert set(grads) == set(want)
        for k in want:
            close(torch.from_numpy(grads[k]), want[k], 2e-5, "rank %d averaged grad of %s" % (rank, k))
    # optimiser: one AdamW step over every tensor leaves finite, changed parameters
    params = {k: sd[k].clone().to(lib.device) for k in want}
    before = {k: v.clone() for k, v in params.items()}
    train.adamw_step(lib, params, want, {}, 1, lr=1e-3)
    changed = sum(int(not torch.equal(params[k], before[k])) for k in params)
    assert all(torch.isfinite(v).all() for v in params.values()) and changed > 0.9 * len(params)
    for k in list(params)[::97]:                                  # a sample of tensors against torch.optim.AdamW
        pt = before[k].cpu().clone().requires_grad_(True)
        opt = torch.optim.AdamW([pt], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
        pt.grad = want[k].cpu().clone()
        opt.step()
        close(params[k], pt, 2e-6, "AdamW step of " + k)


def test_fit_reduces_the_loss_on_a_fixed_batch(lib):
    """mug.train.fit: a few AdamW steps of the whole-model training step on one fixed synthetic batch of the tiny model -- the loss
    goes down (every gradient has the right sign and the optimiser applies it), the parameters stay finite."""
    from mug import train
    case, sd, *_ = _tiny_training_inputs(2, 32)
    sd = {k: v.clone() for k, v in sd.items()}
    torch.manual_seed(0)
    losses = train.fit(lib, sd, case["unet"], case["wave"], steps=4, batch=2, z=32, lr=2e-3, seed=1, fixed_batch=True,
                       audio_ratio=case["audio_ratio"], ntok=case["n_ctx_tok"])
    print("losses", ["%.4f" % v for v in losses])
    assert all(np.isfinite(losses)) and losses[-1] < 0.9 * losses[0], losses
    assert all(torch.isfinite(v).all() for v in sd.values() if v.dtype == torch.float32)


def test_fit_three_steps_match_autograd_plus_torch_adamw(lib):
    """mug.train.fit against the same optimisation done by torch: three AdamW steps on one fixed batch of the tiny model, gradients
    from autograd through the oracle's restatement, torch.optim.AdamW with the same hyper-parameters -- EVERY trainable tensor after
    step 3.  A gradient tensor that moves between steps while the optimiser's table still points at step 1's (the round-3 bug: the four
    time_embed tensors and the prompt-embedding table) shows up here as a tensor that only saw weight decay."""
    from oracle import nets
    from mug import train
    case, sd, *_ = _tiny_training_inputs(2, 32)
    B, z, steps, lr, seed = 2, 32, 3, 1e-3, 1
    sd_fit = {k: v.clone() for k, v in sd.items()}
    train.fit(lib, sd_fit, case["unet"], case["wave"], steps=steps, batch=B, z=z, lr=lr, seed=seed, fixed_batch=True,
              audio_ratio=case["audio_ratio"], ntok=case["n_ctx_tok"])
    # ---- the same three steps with torch (fit's batch and (t, noise) draws, reproduced)
    n_ids = sd["model.cond_stage_model.embedding.weight"].shape[0]
    x0, mel, ids = train.synthetic_batch(B, z, case["unet"], case["wave"], n_ids, case["n_ctx_tok"], case["audio_ratio"], seed * 7919, "cpu")
    g = torch.Generator().manual_seed(seed * 1000)
    t = torch.randint(0, sd["sqrt_alphas_cumprod"].shape[0], (B,), generator=g)
    noise = torch.randn(B, case["unet"]["in_channels"], z, generator=g)
    trainable = [k for k, v in sd.items() if v.dtype == torch.float32 and k.startswith("model.") and not k.startswith("model.first_stage_model")]
    st = {k: (v.clone().requires_grad_(True) if k in trainable else v) for k, v in sd.items()}
    opt, gmin = None, {}
    for _ in range(steps):
        xt = st["sqrt_alphas_cumprod"][t][:, None, None] * x0 + st["sqrt_one_minus_alphas_cumprod"][t][:, None, None] * noise
        pred = nets.unet_forward(st, case["unet"], xt, t, nets.cond_embed(st, ids), nets.wave_encode(st, case["wave"], mel))
        loss = (F.smooth_l1_loss(noise, pred, beta=0.02, reduction="none") + 0.01).mean(dim=[1, 2]).mean()
        for k in trainable:
            st[k].grad = None
        loss.backward()
        if opt is None:                   # fit optimises exactly the tensors the step produced a gradient for
            reached = [k for k in trainable if st[k].grad is not None]
            opt = torch.optim.AdamW([st[k] for k in reached], lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
        for k in reached:
            gmin[k] = st[k].grad.abs() if k not in gmin else torch.minimum(gmin[k], st[k].grad.abs())
        opt.step()
    moved = 0
    for k in reached:
        a, b, o = sd_fit[k].detach().cpu(), st[k].detach(), sd[k]
        # Adam divides by sqrt(v) + 1e-8: where the gradient is rounding noise (a conv bias in front of a GroupNorm is analytically
        # gradient-free) the update's sign is arbitrary on both sides -- compare the elements with a real gradient in all three steps
        alive = gmin[k] > 1e-5
        if not alive.any():
            continue
        # Adam's first steps move every element by ~lr whatever the gradient's size: compare the UPDATE, not the value
        du, dr = (a - o), (b - o)
        tol = 0.05 * steps * lr + 8 * 1.2e-7 * o.abs()          # + a few ulps of the parameter itself (S4's w_imag is O(100))
        assert bool(((du - dr).abs() <= tol)[alive].all()), "%s: update differs from torch's by %.3e (lr %.0e)" % (k, (du - dr).abs().max().item(), lr)
        moved += int(dr.abs().max().item() > 0.5 * lr)
    for k in ("model.unet_model.time_embed.0.weight", "model.unet_model.time_embed.2.bias", "model.cond_stage_model.embedding.weight"):
        du, dr = (sd_fit[k].cpu() - sd[k]), (st[k].detach() - sd[k])
        live = dr.abs() > 0.5 * lr                      # elements the gradient reaches (embedding rows of ids that occur)
        assert live.any() and (du[live] - dr[live]).abs().max().item() <= 0.05 * steps * lr, k
    assert moved > 0.9 * len(reached)


def test_replacing_a_parameter_between_bracketed_steps(lib):
    """The step bracket's cache is keyed by tensor address (include/mugd.h): after a parameter is REPLACED (not updated in place) and the
    plan invalidated, the next bracketed step must use the new tensor -- and nothing may read the old one, which is freed here.  Same
    for a second plan on another state dict after the first plan is gone."""
    import gc
    from mug import train
    case, sd0, x0, noise, t, ids, mel = _tiny_training_inputs(2, 32)
    dev = lib.device
    lib.train_set_precision(True)
    try:
        sd = {k: v.clone().to(dev) for k, v in sd0.items()}
        plan = train.TrainPlan(lib, sd, case["unet"], case["wave"])
        assert plan.bracket and lib._bracket_owner is plan
        args = [v.to(dev) for v in (x0, noise, t, ids, mel)]
        loss_a, _ = plan.step(*args)
        k = "model.unet_model.input_blocks.2.0.in_layers.2.weight"
        sd[k] = (sd[k] * 1.5).contiguous()               # a new tensor; the old one is dropped
        plan.invalidate()
        gc.collect()
        loss_b, grads_b = plan.step(*args)
        grads_b = {q: v.clone() for q, v in grads_b.items()}
        loss_b2, _ = plan.step(*args)                     # the cached packs of the NEW tensor
        assert float(loss_b) == float(loss_b2) and float(loss_b) != float(loss_a)
        # reference: a fresh plan over an equal state dict
        sd_ref = {q: v.clone() for q, v in sd.items()}
        del plan
        gc.collect()
        plan2 = train.TrainPlan(lib, sd_ref, case["unet"], case["wave"])
        assert lib._bracket_owner is plan2
        loss_c, grads_c = plan2.step(*args)
        assert float(loss_c) == float(loss_b)
        assert all(torch.equal(grads_b[q].cpu(), grads_c[q].cpu()) for q in grads_c)
        # fp32 mode after a bf16 step: the bracket re-reads nothing (the bf16 cache is dropped, not refreshed)
        lib.train_set_precision(False)
        del sd, sd_ref
        loss_d, _ = plan2.step(*args)
        assert np.isfinite(float(loss_d))
    finally:
        lib.train_set_precision(False)
        lib.train_step_reset()


# ------------------------------------------------------------------ data parallel: gradient all-reduce over 2 gloo ranks
def _worker(rank, world, port, q):
    import torch.distributed as dist
    from conftest import emu_lib
    from mug import train
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lib = emu_lib()
        B, Cin, Cout, T, Kemb, groups = 4, 32, 32, 24, 32, 8
        p = resblock_params(Cin, Cout, Kemb)
        x, emb, target = rnd(30, B, Cin, T), rnd(31, B, Kemb), rnd(32, B, Cout, T)
        lo, hi = rank * B // world, (rank + 1) * B // world
        loss, grads = train.resblock_loss_and_grads(lib, p, x[lo:hi], emb[lo:hi], target[lo:hi], groups)
        train.allreduce_gradients(grads, average=True)
        q.put((rank, float(loss), {k: v.cpu().numpy() for k, v in grads.items()}))
    finally:
        dist.destroy_process_group()


def test_data_parallel_gradients_equal_full_batch():
    """configs[4] in miniature: each of 2 ranks computes loss + gradients of the block on its half of the batch (native kernels on
    the emulated build), one bucketed all_reduce averages them (gloo here, RCCL on the GPU box): equal to the full-batch gradient."""
    from conftest import emu_lib
    from mug import train
    lib = emu_lib()
    B, Cin, Cout, T, Kemb, groups = 4, 32, 32, 24, 32, 8
    p = resblock_params(Cin, Cout, Kemb)
    x, emb, target = rnd(30, B, Cin, T), rnd(31, B, Kemb), rnd(32, B, Cout, T)
    _, want = train.resblock_loss_and_grads(lib, p, x, emb, target, groups)
    # and the full-batch gradient itself is autograd's
    pt = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    lt = (F.smooth_l1_loss(target, resblock_torch(pt, x, emb, groups), beta=0.02, reduction="none") + 0.01).mean(dim=[1, 2]).mean()
    lt.backward()
    for k in p:
        close(want[k], pt[k].grad, 1e-4, "full-batch grad of " + k)
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = [q.get(timeout=600) for _ in procs]
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0
    for rank, loss, grads in got:
        for k in p:
            close(torch.from_numpy(grads[k]), want[k], 1e-5, "rank %d averaged grad of %s" % (rank, k))


# ------------------------------------------------------------------------------------------------------------------------------
# the reference's own training step (oracle/gen_golden.py --train-only: DDPM.forward -> p_losses -> backward of the REAL
# reference): tests/golden/*_train_*.npz
# ------------------------------------------------------------------------------------------------------------------------------
def block_type(name):
    """Which kind of block a state-dict tensor belongs to (for the per-block-type error report)."""
    rules = (("time_embed", "time_embed"), ("emb_layers", "resblock.emb"), ("cond_stage_model", "prompt embedding"),
             ("relative_position_embedding", "attention tables"), ("C_embedding", "attention tables"),
             ("transformer_blocks.0.attn", "attention linears"), ("transformer_blocks.0.ff", "feed-forward"),
             ("transformer_blocks.0.norm", "layer norms"), ("proj_in", "transformer proj"), ("proj_out", "transformer proj"),
             ("s4_model.kernel", "s4 kernel generator"), ("s4_model", "s4 D / output_linear"), ("out_layer", "s4 out conv"),
             ("in_layers", "resblock convs / norms"), ("out_layers", "resblock convs / norms"), ("skip_connection", "resblock convs / norms"),
             ("nin_shortcut", "resnet block"), ("block.", "resnet block"), ("downsample", "resampling convs"), (".conv.", "resampling convs"),
             ("conv_in", "plain convs"), ("input_blocks.0.0", "plain convs"), (".out.", "out head"), (".norm.", "group norms"))
    for pat, kind in rules:
        if pat in name:
            return kind
    return "other"


def _load_train_fixture(case, z, B):
    from oracle import cases, weights
    g = np.load(os.path.join(cases.GOLDEN, "%s_train_z%d_B%d.npz" % (case["name"], z, B)))
    man = weights.load_manifest(os.path.join(cases.GOLDEN, case["manifest"]))
    sd = weights.set_s4_lengths(weights.make_state_dict(man, seed=0), case["unet"], z)
    seed = int(g["seed"])
    note_t, mel, ids = cases.train_batch(case, seed, B, z, sd["model.cond_stage_model.embedding.weight"].shape[0])
    return g, sd, seed, {"note": note_t, "audio": mel, "feature": ids}


def _check_against_train_fixture(case, g, seed, loss, grads, tol_norm, tol_elem):
    """loss, the L2 norm of EVERY gradient tensor and the stored gradients (all of them for the tiny model; a spread over every
    block type, sampled above 8192 elements, for the shipped one) against the real reference's."""
    from oracle import cases
    assert abs(float(loss) - float(g["loss"])) <= 2e-5 * max(1.0, abs(float(g["loss"]))), (float(loss), float(g["loss"]))
    names = [str(n) for n in g["names"]]
    assert sorted(grads) == names, "gradient set differs from the reference's: %s" % sorted(set(grads) ^ set(names))[:6]
    worst = {}
    gmax = float(np.max(g["absmax"]))
    ZERO_GRAD = 1e-6 * gmax
    for n, ref_norm, ref_max in zip(names, g["norms"], g["absmax"]):
        got = float(grads[n].double().norm())
        rel = abs(got - ref_norm) / max(ref_norm, 1e-12)
        if ref_max < ZERO_GRAD:                     # analytically zero (a bias in front of a GroupNorm): rounding noise on both sides
            assert got <= 1e-4 * gmax * max(1.0, grads[n].numel() ** 0.5), "gradient of %s should vanish: |g| %.3e" % (n, got)
            continue
        bt = block_type(n)
        worst[bt] = max(worst.get(bt, (0.0, "")), (rel, n))
        assert rel <= tol_norm, "|grad| of %s: %.6e, reference %.6e" % (n, got, ref_norm)
    print("worst relative error of a gradient NORM per block type (vs the reference's DDPM.p_losses backward):")
    for bt, (rel, n) in sorted(worst.items()):
        print("   %-26s %.2e   %s" % (bt, rel, n))
    worst = {}
    for i, n in enumerate(str(x) for x in g["full_names"]):
        ref = torch.from_numpy(g["g%d" % i]).reshape(-1)
        got = grads[n].detach().cpu().reshape(-1)
        if got.numel() > 8192:
            got = got[torch.from_numpy(cases.train_sample_index(seed, i, got.numel()))]
        d, sc = (got - ref).abs().max().item(), ref.abs().max().item()
        if sc < ZERO_GRAD:
            continue
        bt = block_type(n)
        worst[bt] = max(worst.get(bt, (0.0, "")), (d / sc, n))
        assert d <= tol_elem * sc, "gradient of %s: max|diff| %.3e (reference max %.3e)" % (n, d, sc)
    print("worst element error / max|reference| per block type over the %d stored gradients:" % len(g["full_names"]))
    for bt, (rel, n) in sorted(worst.items()):
        print("   %-26s %.2e   %s" % (bt, rel, n))


def test_tiny_training_step_vs_reference_golden(lib):
    """The whole step -- frozen VAE encode -> mode(), q_sample, prompt embedding, wave encoder, U-Net, smooth-L1, backward -- on the
    tiny model against the REAL reference's DDPM.forward / p_losses / backward (fixture from oracle/gen_golden.py --train-only)."""
    from oracle import cases
    from mug import train
    case, z, B = cases.TINY, 32, 2
    g, sd, seed, batch = _load_train_fixture(case, z, B)
    loss, grads = train.training_step_from_batch(lib, sd, case["unet"], case["wave"], case["vae"], batch, torch.from_numpy(g["t"]), torch.from_numpy(g["noise"]))
    x0 = train.encode_x0(lib, sd, case["vae"], batch["note"]).cpu()
    assert (x0 - torch.from_numpy(g["x_start"])).abs().max().item() <= 1e-4 * max(1.0, float(np.abs(g["x_start"]).max()))
    _check_against_train_fixture(case, g, seed, loss, grads, tol_norm=1e-4, tol_elem=3e-4)


@pytest.mark.gpu
def test_shipped_training_step_vs_reference_golden_and_autograd(gpu_lib):
    """configs[4]'s model: the SHIPPED architecture's whole training step (151 M parameters, 1327 gradient tensors) against
    (a) the real reference's DDPM.forward / p_losses / backward fixture (loss, every gradient's norm, 31 gradients across all block
    types) and (b) torch autograd through the oracle's restatement for EVERY tensor, element by element."""
    from oracle import cases, nets
    from mug import train
    lib = gpu_lib
    case, z, B = cases.FULL, 96, 2
    g, sd, seed, batch = _load_train_fixture(case, z, B)
    t, noise = torch.from_numpy(g["t"]), torch.from_numpy(g["noise"])
    loss, grads = train.training_step_from_batch(lib, sd, case["unet"], case["wave"], case["vae"], batch, t, noise)
    _check_against_train_fixture(case, g, seed, loss, grads, tol_norm=1e-4, tol_elem=3e-4)
    # (b) autograd through the oracle, all tensors
    trainable = [str(n) for n in g["names"]]
    st = {k: (v.clone().requires_grad_(True) if k in trainable else v) for k, v in sd.items()}
    x0 = torch.from_numpy(g["x_start"])
    xt = st["sqrt_alphas_cumprod"][t][:, None, None] * x0 + st["sqrt_one_minus_alphas_cumprod"][t][:, None, None] * noise
    pred = nets.unet_forward(st, case["unet"], xt, t, nets.cond_embed(st, batch["feature"]), nets.wave_encode(st, case["wave"], batch["audio"]))
    lt = (F.smooth_l1_loss(noise, pred, beta=0.02, reduction="none") + 0.01).mean(dim=[1, 2]).mean()
    lt.backward()
    assert abs(float(lt) - float(g["loss"])) == 0.0 or abs(float(lt) - float(g["loss"])) < 1e-6
    worst = {}
    gmax = float(np.max(g["absmax"]))
    for k in trainable:
        a, b = grads[k].detach().cpu().reshape(st[k].shape), st[k].grad
        d, sc = (a - b).abs().max().item(), b.abs().max().item()
        if sc < 1e-6 * gmax:
            continue
        bt = block_type(k)
        worst[bt] = max(worst.get(bt, (0.0, "")), (d / sc, k))
        assert d <= 3e-4 * sc, "gradient of %s: max|diff| %.3e (autograd max %.3e)" % (k, d, sc)
    print("worst element error / max|autograd| per block type over all %d tensors:" % len(trainable))
    for bt, (rel, n) in sorted(worst.items()):
        print("   %-26s %.2e   %s" % (bt, rel, n))


@pytest.mark.gpu
def test_shipped_training_step_at_the_timed_geometry_vs_autograd(gpu_lib):
    """The geometry bench.py times -- shipped model, z = 512, 32768-frame mel -- at batch 4 (the batch-32 step is the same launch list
    with 8x the rows per launch): loss and the gradient of ALL 1327 trainable tensors against autograd through the oracle's restatement
    (identical to the reference on the z = 96 fixture).  Covers what the z = 96 tests cannot: the multi-tile walks of the long layers
    (T = 32768 ... 512), S4 at L = 512, 2048-column GEMMs, the 0.5 GB-tensor layers of the wave encoder.  fp32 mode at the fp32
    tolerances of the z = 96 test; then the bf16 step against the fp32 step at the stated bf16 tolerance."""
    from oracle import cases, nets, weights
    from mug import train
    case, z, B = cases.FULL, 512, 4
    man = weights.load_manifest(os.path.join(cases.GOLDEN, case["manifest"]))
    sd = weights.set_s4_lengths(weights.make_state_dict(man, 7), case["unet"], z)
    n_ids = sd["model.cond_stage_model.embedding.weight"].shape[0]
    x0, noise = rnd(170, B, 16, z), rnd(171, B, 16, z)
    t = torch.tensor([17, 803, 402, 999])
    ids = torch.from_numpy(np.random.default_rng(5).integers(0, n_ids, (B, case["n_ctx_tok"])))
    mel = cases.mel_input(case, 172, B, z * case["audio_ratio"])
    trainable = [k for k, v in sd.items() if v.dtype == torch.float32 and k.startswith("model.") and not k.startswith("model.first_stage_model")]
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, threads))
    try:
        st = {k: (v.clone().requires_grad_(True) if k in trainable else v) for k, v in sd.items()}
        xt = st["sqrt_alphas_cumprod"][t][:, None, None] * x0 + st["sqrt_one_minus_alphas_cumprod"][t][:, None, None] * noise
        pred = nets.unet_forward(st, case["unet"], xt, t, nets.cond_embed(st, ids), nets.wave_encode(st, case["wave"], mel))
        lt = (F.smooth_l1_loss(noise, pred, beta=0.02, reduction="none") + 0.01).mean(dim=[1, 2]).mean()
        lt.backward()
    finally:
        torch.set_num_threads(threads)
    dev = gpu_lib.device
    sdd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sd.items()}
    args = [v.to(dev) for v in (x0, noise, t, ids, mel)]
    loss, grads = train.training_step(gpu_lib, sdd, case["unet"], case["wave"], *args)
    assert abs(float(loss) - float(lt)) <= 1e-5 * max(1.0, abs(float(lt))), (float(loss), float(lt))
    gmax = max(float(st[k].grad.abs().max()) for k in trainable if st[k].grad is not None)
    worst, n_cmp = {}, 0
    for k in trainable:
        if st[k].grad is None:
            continue
        a, b = grads[k].detach().cpu().reshape(st[k].shape), st[k].grad
        d, sc = (a - b).abs().max().item(), b.abs().max().item()
        if sc < 1e-6 * gmax:
            continue                                     # analytically vanishing gradients (a conv bias in front of a GroupNorm): noise on both sides
        n_cmp += 1
        bt = block_type(k)
        worst[bt] = max(worst.get(bt, (0.0, "")), (d / sc, k))
        assert d <= 3e-4 * sc, "gradient of %s: max|diff| %.3e (autograd max %.3e)" % (k, d, sc)
    assert n_cmp > 1200
    print("z = 512, B = 4: worst element error / max|autograd| per block type over %d tensors:" % n_cmp)
    for bt, (rel, n) in sorted(worst.items()):
        print("   %-26s %.2e   %s" % (bt, rel, n))
    g32 = {k: v.detach().clone() for k, v in grads.items()}
    gpu_lib.train_set_precision(True)
    try:
        loss16, g16 = train.training_step(gpu_lib, sdd, case["unet"], case["wave"], *args)
    finally:
        gpu_lib.train_set_precision(False)
    _report_bf16(float(loss16), float(loss), g16, g32)


# ------------------------------------------------------------------------------------------------------------------------------
# bf16 training GEMMs (k_tgemm.hip; mugd_train_set_precision(1)): BASELINE configs[4]'s precision
# ------------------------------------------------------------------------------------------------------------------------------
def _r16(t):
    return t.detach().to(torch.bfloat16).to(torch.float32)


class _ConvBF16(torch.autograd.Function):
    """conv1d whose GEMM operands are rounded to bfloat16 (round to nearest even) with fp32 accumulation, forward AND backward:
    y = conv(r(a), r(w)) + b;  da = conv^T(r(dy), r(w));  dw = corr(r(dy), r(a));  db = sum dy -- what tconv / twgrad compute."""

    @staticmethod
    def forward(ctx, a, w, b, stride, padding, dilation):
        ctx.save_for_backward(a, w)
        ctx.geo = (stride, padding, dilation)
        return F.conv1d(_r16(a), _r16(w), b, stride=stride, padding=padding, dilation=dilation)

    @staticmethod
    def backward(ctx, dy):
        a, w = ctx.saved_tensors
        stride, padding, dilation = ctx.geo
        da = torch.nn.grad.conv1d_input(a.shape, _r16(w), _r16(dy), stride=stride, padding=padding, dilation=dilation)
        dw = torch.nn.grad.conv1d_weight(_r16(a), w.shape, _r16(dy), stride=stride, padding=padding, dilation=dilation)
        return da, dw, dy.sum(dim=(0, 2)), None, None, None


@pytest.mark.parametrize("B,Cin,Cout,T,taps,dil,mode,gn", [(2, 32, 48, 40, 3, 1, 0, False), (2, 32, 32, 64, 3, 4, 0, False), (1, 48, 32, 50, 3, 8, 0, False),
                                                           (2, 32, 64, 36, 1, 1, 0, False), (2, 32, 48, 40, 3, 1, 1, False), (2, 48, 32, 26, 3, 1, 2, False),
                                                           (2, 64, 16, 40, 3, 1, 0, True), (2, 32, 160, 200, 3, 2, 0, False), (1, 144, 32, 132, 1, 1, 0, False),
                                                           (2, 32, 32, 130, 3, 1, 1, False), (3, 64, 128, 512, 1, 1, 0, False), (2, 128, 128, 256, 3, 2, 0, False)])
def test_conv_layer_bf16_gemms_vs_rounded_operand_reference(lib, B, Cin, Cout, T, taps, dil, mode, gn):
    """The bf16 training GEMMs (tconv forward / data gradient, twgrad) on every conv geometry of the model -- plain, dilated, Downsample,
    Upsample, 1x1, the GroupNorm + SiLU head; row counts off the 128-row block, channel counts off the 64-channel stage, lengths off the
    64-sample tile -- against a torch reference with the SAME arithmetic: operands rounded to bfloat16, fp32 accumulation.  Only the
    summation order differs, so the tolerance is an fp32 one: an indexing error cannot hide behind bf16 noise."""
    w, b = rnd(50, Cout, Cin, taps, scale=(taps * Cin) ** -0.5), 0.1 * rnd(51, Cout)
    x = rnd(52, B, Cin, T)
    gw, gb = 1 + 0.1 * rnd(53, Cin), 0.1 * rnd(54, Cin)
    wt, bt, xt, gwt, gbt = [v.clone().requires_grad_(True) for v in (w, b, x, gw, gb)]
    a = F.silu(F.group_norm(xt, 8, gwt, gbt, eps=1e-6)) if gn else xt
    if mode == 0:
        yt = _ConvBF16.apply(a, wt, bt, 1, dil * (taps - 1) // 2, dil)
    elif mode == 1:
        yt = _ConvBF16.apply(F.pad(a, (0, 1)), wt, bt, 2, 0, 1)
    else:
        yt = _ConvBF16.apply(a.repeat_interleave(2, dim=-1), wt, bt, 1, 1, 1)
    dy = rnd(55, *yt.shape)
    yt.backward(dy)
    lib.train_set_precision(True)
    try:
        y, dx, dw, db, dg = lib.train_conv(w, b, x, dy, dil=dil, mode=mode, gn=(gw, gb) if gn else None, groups=8)
    finally:
        lib.train_set_precision(False)
    close(y, yt, 3e-5, "forward")
    if mode == 2:
        # the reference rounds the UPSAMPLED gradient pairs' sum after the conv; the kernel sums pairs of fp32 results: same values
        pass
    close(dx, xt.grad, 1e-4, "dx")
    close(dw, wt.grad, 2e-4, "dw")
    close(db, bt.grad, 1e-4, "db")
    if gn:
        close(dg[0], gwt.grad, 2e-4, "d gamma")
        close(dg[1], gbt.grad, 2e-4, "d beta")


# Stated tolerance of the bf16 mode (bf16 MFMA inputs: 8 mantissa bits per operand, fp32 accumulation) against the fp32 path:
# (measured on the tiny model: 4.4 % globally, 4-8 % per tensor, uniform over depth -- the smooth-L1 loss with beta = 0.02 turns a
# forward perturbation d of the prediction into a gradient perturbation d / 0.02 wherever |pred - target| < beta, so the gradient
# noise is set by the loss, not by accumulated rounding; the GEMM kernels themselves are pinned to fp32 tolerances by
# test_conv_layer_bf16_gemms_vs_rounded_operand_reference)
BF16_GLOBAL_L2 = 8e-2     # ||g16 - g32|| over ALL parameters <= 8 % of ||g32||
BF16_REL_L2 = 0.15        # per gradient tensor: ||g16 - g32|| <= 15 % of ||g32|| ...
BF16_MAX_ELEM = 0.35      # ... and no element further off than 35 % of the tensor's largest |g32|


def _bf16_err(a, b):
    a, b = a.detach().cpu().double().reshape(-1), b.detach().cpu().double().reshape(-1)
    return (a - b).norm().item() / b.norm().item(), (a - b).abs().max().item() / b.abs().max().item()


def test_whole_model_training_step_bf16_vs_fp32(lib):
    """The whole tiny-model step with the GEMMs on the bf16 matrix cores against the fp32 step (which equals the reference's, above):
    loss within 1e-2 relative, every gradient tensor within the stated bf16 tolerance (relative L2 error <= 6 %, no element off by more
    than 25 % of the tensor's maximum); the worst tensor per block type is printed."""
    from oracle import cases
    from mug import train
    case, z, B = cases.TINY, 32, 2
    g, sd, seed, batch = _load_train_fixture(case, z, B)
    t, noise = torch.from_numpy(g["t"]), torch.from_numpy(g["noise"])
    loss32, g32 = train.training_step_from_batch(lib, sd, case["unet"], case["wave"], case["vae"], batch, t, noise)
    lib.train_set_precision(True)
    try:
        loss16, g16 = train.training_step_from_batch(lib, sd, case["unet"], case["wave"], case["vae"], batch, t, noise)
    finally:
        lib.train_set_precision(False)
    assert abs(float(loss16) - float(loss32)) <= 1e-2 * abs(float(loss32)), (float(loss16), float(loss32))
    _report_bf16(loss16, loss32, g16, g32)


def _report_bf16(loss16, loss32, g16, g32):
    gmax = max(float(v.abs().max()) for v in g32.values())
    worst, bad = {}, []
    num = den = 0.0
    for k in g32:
        num += float((g16[k].double().cpu() - g32[k].double().cpu()).pow(2).sum())
        den += float(g32[k].double().pow(2).sum())
        if float(g32[k].abs().max()) < 1e-6 * gmax:
            continue
        rel, mx = _bf16_err(g16[k], g32[k])
        bt = block_type(k)
        worst[bt] = max(worst.get(bt, (0.0, 0.0, "")), (rel, mx, k))
        if rel > BF16_REL_L2 or mx > BF16_MAX_ELEM:
            bad.append((k, rel, mx))
    print("bf16 GEMMs vs fp32: loss %.6f vs %.6f, global relative L2 error of the gradient %.3e; worst tensor per block type, relative L2 (max-element / max):"
          % (float(loss16), float(loss32), (num / den) ** 0.5))
    for bt, (rel, mx, n) in sorted(worst.items()):
        print("   %-26s %.2e  (%.2e)   %s" % (bt, rel, mx, n))
    assert (num / den) ** 0.5 <= BF16_GLOBAL_L2, "global gradient error %.3e" % (num / den) ** 0.5
    assert not bad, "outside the stated bf16 tolerance: %s" % bad[:5]


@pytest.mark.gpu
def test_shipped_training_step_bf16_vs_fp32(gpu_lib):
    """configs[4]'s model in configs[4]'s precision: the shipped architecture's whole step with bf16 MFMA inputs against its fp32 step
    (pinned to the reference above), stated bf16 tolerance, worst tensor per block type printed."""
    from oracle import cases
    from mug import train
    lib = gpu_lib
    case, z, B = cases.FULL, 96, 2
    g, sd, seed, batch = _load_train_fixture(case, z, B)
    t, noise = torch.from_numpy(g["t"]), torch.from_numpy(g["noise"])
    loss32, g32 = train.training_step_from_batch(lib, sd, case["unet"], case["wave"], case["vae"], batch, t, noise)
    lib.train_set_precision(True)
    try:
        loss16, g16 = train.training_step_from_batch(lib, sd, case["unet"], case["wave"], case["vae"], batch, t, noise)
    finally:
        lib.train_set_precision(False)
    assert abs(float(loss16) - float(loss32)) <= 1e-2 * abs(float(loss32)), (float(loss16), float(loss32))
    _report_bf16(loss16, loss32, g16, g32)
