"""Operator parity: each HIP kernel (through the C ABI) against the plain PyTorch-CPU fp32
op it replaces, on seeded inputs.  Runs on the emulated build here and on the MI355X under
`-m gpu`.  Tolerances are fp32-reassociation level: the attention / S4 / norm kernels compute in fp32, conv_gemm on split-f16 operands with
fp32 accumulation (H3: fp32-equivalent at every operand scale, see the operand-scale tests below)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nets, s4 as s4o, weights


def rnd(seed, *shape, scale=1.0):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape).astype(np.float32)) * scale


def close(got, ref, atol, rtol=1e-5, what=""):
    got = got.detach().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = ~(err <= tol)
    if bad.any():
        i = int(torch.where(bad.flatten())[0][0])
        raise AssertionError("%s: %d/%d off, max err %.3e; first bad at %s (ref %.5f got %.5f)" % (
            what, int(bad.sum()), bad.numel(), float(err[~torch.isnan(err)].max()) if (~torch.isnan(err)).any() else float("nan"),
            tuple(int(v) for v in np.unravel_index(i, tuple(err.shape))), ref.flatten()[i].item(), got.flatten()[i].item()))


@pytest.mark.parametrize("B,C,T,groups,silu", [(2, 64, 48, 32, 1), (1, 96, 100, 32, 0), (2, 32, 33, 8, 1), (1, 32, 16, 32, 1)])
def test_group_norm(lib, B, C, T, groups, silu):
    x, g, b = rnd(1, B, C, T, scale=2.0) + 0.5, 1 + 0.1 * rnd(2, C), 0.1 * rnd(3, C)
    ref = F.group_norm(x, groups, g, b, eps=1e-6)
    if silu:
        ref = F.silu(ref)
    close(lib.op_group_norm(x, g, b, groups, silu), ref, 2e-5, what="group_norm")


@pytest.mark.parametrize("B,C,T", [(2, 64, 40), (1, 48, 21), (1, 256, 7)])
def test_layer_norm(lib, B, C, T):
    x, g, b = rnd(4, B, C, T, scale=3.0), 1 + 0.1 * rnd(5, C), 0.1 * rnd(6, C)
    ref = F.layer_norm(x.transpose(1, 2), (C,), g, b, eps=1e-5).transpose(1, 2)
    close(lib.op_layer_norm(x, g, b), ref, 2e-5, what="layer_norm")


CONV_CASES = [
    # B, C, T, M, taps, dil, stride, pad, ups
    (2, 32, 40, 32, 3, 1, 1, 1, 0),      # plain k3
    (1, 48, 33, 64, 1, 1, 1, 0, 0),      # 1x1, ragged T
    (2, 16, 64, 48, 3, 1, 1, 1, 0),      # single K-chunk (only one wave has work), M not multiple of 32
    (1, 32, 50, 16, 3, 2, 1, 2, 0),      # dilation 2, M=16 (padded tile)
    (1, 32, 70, 32, 3, 8, 1, 8, 0),      # dilation 8 (wave encoder)
    (1, 32, 64, 32, 3, 4, 1, 4, 0),      # dilation 4
    (2, 32, 48, 32, 3, 1, 2, 0, 0),      # Downsample: pad right 1, stride 2
    (1, 32, 37, 32, 3, 1, 2, 0, 0),      # Downsample, odd length
    (2, 32, 24, 32, 3, 1, 1, 1, 1),      # Upsample: nearest x2 then k3
    (1, 144, 12, 96, 3, 1, 1, 1, 0),     # short sequence (z=96 level 3), K split unevenly over 4 waves
]


@pytest.mark.parametrize("B,C,T,M,taps,dil,stride,pad,ups", CONV_CASES)
def test_conv1d(lib, B, C, T, M, taps, dil, stride, pad, ups):
    x, w, b = rnd(7, B, C, T), rnd(8, M, C, taps, scale=1.0 / math.sqrt(C * taps)), 0.1 * rnd(9, M)
    xin = x.repeat_interleave(2, dim=-1) if ups else x
    if stride == 2:
        xin = F.pad(xin, (0, 1))
    ref = F.conv1d(xin, w, b, stride, pad, dil)
    resid = rnd(10, *ref.shape)
    got = lib.op_conv1d(x, w, b, resid, dil=dil, stride=stride, pad=pad, upsample=bool(ups), Tout=ref.shape[-1])
    close(got, ref + resid, 2e-5, what="conv1d")


NORM_CONV_CASES = [
    # B, C, T, M, taps, dil, norm, groups, silu, wk
    (2, 64, 48, 64, 3, 1, 1, 32, 1, 0),      # GroupNorm+SiLU -> k3 (ResBlock half), fast window path (T % 4 == 0)
    (2, 64, 48, 64, 3, 1, 1, 32, 1, 1),      # same, one wave owns all of K
    (2, 64, 48, 64, 3, 1, 1, 32, 1, 2),
    (1, 128, 40, 32, 3, 1, 1, 32, 1, 4),     # ragged last tile (40 = 32 + 8)
    (1, 128, 40, 32, 3, 1, 1, 32, 1, 8),
    (2, 64, 48, 64, 3, 1, 1, 32, 2, 4),      # silu = 2: SiLU on v_exp_f32 / v_rcp_f32
    (1, 96, 36, 48, 1, 1, 1, 32, 0, 0),      # GroupNorm -> 1x1 (transformer proj_in)
    (1, 32, 33, 32, 3, 1, 1, 8, 1, 0),       # T % 4 != 0: generic window path
    (1, 32, 64, 32, 3, 2, 1, 8, 1, 0),       # dilated (wave encoder / VAE ResnetBlock)
    (1, 32, 64, 32, 3, 4, 1, 8, 1, 0),
    (1, 32, 72, 32, 3, 8, 1, 8, 1, 2),
    (2, 64, 40, 96, 1, 1, 2, 0, 0, 0),       # LayerNorm -> Linear (q/k/v, GEGLU projection)
    (1, 48, 21, 32, 1, 1, 2, 0, 0, 0),       # LayerNorm, generic path
    (1, 256, 8, 64, 1, 1, 2, 0, 0, 4),       # LayerNorm, short sequence
]


@pytest.mark.parametrize("B,C,T,M,taps,dil,norm,groups,silu,wk", NORM_CONV_CASES)
def test_norm_conv1d_fused(lib, B, C, T, M, taps, dil, norm, groups, silu, wk):
    x = rnd(30, B, C, T, scale=2.0) + 0.3
    g, b = 1 + 0.1 * rnd(31, C), 0.1 * rnd(32, C)
    w, bias = rnd(33, M, C, taps, scale=1.0 / math.sqrt(C * taps)), 0.1 * rnd(34, M)
    if norm == 1:
        n = F.group_norm(x, groups, g, b, eps=1e-6)
        if silu:
            n = F.silu(n)
    else:
        n = F.layer_norm(x.transpose(1, 2), (C,), g, b, eps=1e-5).transpose(1, 2)
    pad = dil * (taps - 1) // 2
    ref = F.conv1d(n, w, bias, 1, pad, dil)
    got = lib.op_norm_conv1d(x, g, b, w, bias, dil=dil, pad=pad, norm=norm, groups=groups, silu=silu, wk=wk)
    close(got, ref, 3e-5, what="norm+conv1d")


@pytest.mark.parametrize("tn", [16, 32])
@pytest.mark.parametrize("B,C,T,M,taps,dil,norm,groups,silu,wk", [c for c in NORM_CONV_CASES if c[2] % 4 == 0 and c[5] == 1])
def test_norm_conv1d_tile_widths(lib, tn, B, C, T, M, taps, dil, norm, groups, silu, wk):
    """Both tile widths of the conv_gemm template (32 x 32 and 32 x 16 tiles: csrc/conv_body.h, ConvGeo) give the same conv."""
    lib.set_conv_tiling(0, tn)
    try:
        test_norm_conv1d_fused(lib, B, C, T, M, taps, dil, norm, groups, silu, wk)
    finally:
        lib.set_conv_tiling(0, 0)


@pytest.mark.parametrize("tn", [16, 32])
@pytest.mark.parametrize("B,C,T,M,taps,dil,stride,pad,ups", [c for c in CONV_CASES if c[2] % 4 == 0 and c[5] == 1 and c[6] == 1 and not c[8]])
def test_conv1d_tile_widths(lib, tn, B, C, T, M, taps, dil, stride, pad, ups):
    lib.set_conv_tiling(0, tn)
    try:
        test_conv1d(lib, B, C, T, M, taps, dil, stride, pad, ups)
        for wk in (1, 2, 4, 8):
            lib.set_conv_tiling(wk, tn)
            test_conv1d(lib, B, C, T, M, taps, dil, stride, pad, ups)
    finally:
        lib.set_conv_tiling(0, 0)


WIDE_CASES = [
    # B, C, T, M, taps: the M-split ("wide") form of conv_gemm (conv_body.h: MS) -- 2 / 4 / 8 row tiles per workgroup, ragged last row group,
    # M off the 32-row grid, ragged last column tile, K not a multiple of the group (partial last phase), K shorter than one phase
    (2, 64, 64, 64, 3), (1, 48, 40, 96, 3), (2, 32, 36, 80, 1), (1, 160, 72, 160, 3), (1, 16, 64, 272, 1), (3, 208, 32, 128, 1), (1, 144, 64, 320, 3),
]


@pytest.mark.parametrize("mode", ["1", "2"])
@pytest.mark.parametrize("B,C,T,M,taps", WIDE_CASES)
def test_conv1d_wide_tiles(lib, monkeypatch, B, C, T, M, taps, mode):
    """MUGD_CONV_WIDE=1 forces the M-split form wherever it exists: the waves of a workgroup own different row tiles and share every staged
    window (= 2: its M-split x K-split variant, two K-slices per workgroup).  Plain conv + bias + residual, GroupNorm(+SiLU) / LayerNorm
    operand transforms, gated epilogues -- same results."""
    monkeypatch.setenv("MUGD_CONV_WIDE", mode)
    pad = (taps - 1) // 2
    x, w, b = rnd(7, B, C, T), rnd(8, M, C, taps, scale=1.0 / math.sqrt(C * taps)), 0.1 * rnd(9, M)
    ref = F.conv1d(x, w, b, 1, pad)
    resid = rnd(10, *ref.shape)
    lib.set_conv_tiling(0, 32)
    try:
        close(lib.op_conv1d(x, w, b, resid, dil=1, stride=1, pad=pad, Tout=T), ref + resid, 2e-5, what="wide conv1d")
        if T % 4 == 0:
            test_norm_conv1d_fused(lib, B, C, T, M, taps, 1, 1, 16, 1, 0)
            if taps == 1:
                test_norm_conv1d_fused(lib, B, C, T, M, 1, 1, 2, 0, 0, 0)
    finally:
        lib.set_conv_tiling(0, 0)


def test_conv1d_shape_the_host_rule_sends_to_the_wide_form(lib, monkeypatch):
    """launch_conv_gemm picks the M-split form by itself for tall-M / short-K launches with >= 160 workgroups (k_conv.hip): such a shape,
    default settings, against the reference convolution -- and the K-split form of the same launch (MUGD_CONV_WIDE=0) agrees."""
    B, C, T, M = 8, 32, 320, 512              # 16 row tiles -> 2 groups of 8, 10 column tiles, batch 8: 160 workgroups; K = 2 chunks
    x, w, b = rnd(31, B, C, T), rnd(32, M, C, 1, scale=1.0 / math.sqrt(C)), 0.1 * rnd(33, M)
    ref = F.conv1d(x, w, b)
    lib.set_conv_tiling(0, 32)
    try:
        got = lib.op_conv1d(x, w, b)
        close(got, ref, 2e-5, what="rule-selected form")
        monkeypatch.setenv("MUGD_CONV_WIDE", "0")
        close(lib.op_conv1d(x, w, b), got.detach().cpu(), 2e-6, what="K-split form of the same launch")
    finally:
        lib.set_conv_tiling(0, 0)


@pytest.mark.parametrize("mode", ["1", "2"])
@pytest.mark.parametrize("epi", [1, 2])
def test_conv1d_gated_wide(lib, monkeypatch, epi, mode):
    monkeypatch.setenv("MUGD_CONV_WIDE", mode)
    lib.set_conv_tiling(0, 32)
    try:
        test_conv1d_gated(lib, epi)
    finally:
        lib.set_conv_tiling(0, 0)


@pytest.mark.parametrize("tn", [16, 32])
@pytest.mark.parametrize("epi", [1, 2])
def test_conv1d_gated_tile_widths(lib, tn, epi):
    lib.set_conv_tiling(0, tn)
    try:
        test_conv1d_gated(lib, epi)
    finally:
        lib.set_conv_tiling(0, 0)


@pytest.mark.parametrize("epi", [1, 2])
def test_conv1d_gated(lib, epi):
    B, C, T, M = 2, 32, 40, 128
    x, w, b = rnd(11, B, C, T), rnd(12, M, C, 1, scale=1.0 / math.sqrt(C)), 0.1 * rnd(13, M)
    y = F.conv1d(x, w, b)
    a, g = y.chunk(2, dim=1)
    ref = a * torch.sigmoid(g) if epi == 1 else a * F.gelu(g)
    close(lib.op_conv1d(x, w, b, epi=epi), ref, 2e-5, what="gated conv")


@pytest.mark.parametrize("epi", [1, 2])
def test_fast_erf_and_sigmoid_match_the_library_forms(lib, epi):
    """The gated epilogues' branch-free erf-GELU / sigmoid (csrc/common.h: erf_fast, gelu_gate, sigmoid_gate) on a dense sweep of gate values,
    isolated from the GEMM: value rows = 1 (zero weights, unit bias), gate rows = one input channel each (unit weight), so the output IS
    gelu(x) resp. sigmoid(x) up to the exact products 1 * x.  Against float64: the accuracy of torch's / libm's fp32 forms (a few 1e-7 of
    max(1, |x|)), on both sides of the polynomial switch at |x / sqrt(2)| = 1 and out to where erf saturates."""
    C, T = 16, 2048
    xs = torch.linspace(-9.0, 9.0, C * T, dtype=torch.float64).reshape(T, C).t().contiguous()      # channel c, sample t
    x = xs.float()[None]                                                                             # (1, C, T)
    M = 64                                                                                           # rows 0..31 value, 32..63 gate
    w = torch.zeros(M, C, 1)
    for m in range(32):
        w[32 + m, m % C, 0] = 1.0
    b = torch.zeros(M)
    b[:32] = 1.0
    got = lib.op_conv1d(x, w, b, epi=epi).detach().cpu().double()[0]                                # (32, T): row m = gate channel m % 16
    g = x[0].double()
    ref = (torch.sigmoid(g) if epi == 1 else 0.5 * g * (1.0 + torch.erf(g / math.sqrt(2.0))))
    for m in range(32):
        err = (got[m] - ref[m % C]).abs()
        tol = 4e-7 * torch.clamp(g[m % C].abs(), min=1.0)
        assert (err <= tol).all(), (epi, m, float((err / tol).max()))


# ---------------------------------------------------------------------------------------------------------------------
# The DOMAIN of conv_gemm's split-f16 arithmetic (csrc/conv_body.h: "The DOMAIN of H3").  The reference is fp32 with a +-3e38 range
# (/root/reference/mug/diffusion/unet.py:27-33: convert_module_to_f16 is a no-op); f16 halves alone overflow above 65504 and lose their
# low half below ~2^-12 (round 4: all-NaN tiles at |x| ~ 1e5, relative error 1e-3 at 1e-8).  Both operands are therefore carried as
# block floating point (weights: one power of two per packed set; normalised activations: one per tensor from the affine bound; raw
# activations: 2^8 while a wave's K-slice stays inside the band, else a per-wave power of two that follows the data chunk by chunk), and
# these tests hold the kernels to the SAME fp32 tolerance at every operand scale, against float64.
# ---------------------------------------------------------------------------------------------------------------------
SCALES = [1e-8, 1e-6, 1.0, 1e5, 1e7]


def conv_ref64(x, w, b=None, stride=1, pad=0, dil=1):
    return F.conv1d(x.double(), w.double(), None if b is None else b.double(), stride, pad, dil)


def close_scaled(got, ref64, scale, atol=2e-5, rtol=1e-5, what=""):
    """fp32 tolerance of the unit-scale tests, relative to the product of the operand scales."""
    got = got.detach().cpu().double() / scale
    ref = ref64 / scale
    assert torch.isfinite(got).all(), "%s: %d non-finite outputs" % (what, int((~torch.isfinite(got)).sum()))
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bad.any(), "%s: %d/%d off, max err %.3e (scaled units)" % (what, int(bad.sum()), bad.numel(), float(err.max()))


@pytest.mark.parametrize("sw", [1e-6, 1.0, 1e5])
@pytest.mark.parametrize("sx", SCALES)
@pytest.mark.parametrize("B,C,T,M,taps,dil,stride,pad,ups", [CONV_CASES[0], CONV_CASES[1], CONV_CASES[6], CONV_CASES[9]])
def test_conv1d_operand_scales(lib, sx, sw, B, C, T, M, taps, dil, stride, pad, ups):
    x, w = rnd(7, B, C, T) * sx, rnd(8, M, C, taps, scale=1.0 / math.sqrt(C * taps)) * sw
    b = 0.1 * rnd(9, M) * (sx * sw)
    xin = F.pad(x, (0, 1)) if stride == 2 else x
    ref = conv_ref64(xin, w, b, stride, pad, dil)
    got = lib.op_conv1d(x, w, b, None, dil=dil, stride=stride, pad=pad, upsample=bool(ups), Tout=ref.shape[-1])
    close_scaled(got, ref, sx * sw, what="conv1d sx=%g sw=%g" % (sx, sw))


@pytest.mark.parametrize("sx,sw", [(1e-30, 1.0), (1e-20, 1e-10), (1e-12, 1.0), (1e20, 1.0), (1e30, 1e-3), (1e-3, 1e30), (1e25, 1e10)])
def test_conv1d_operand_scales_far_out(lib, sx, sw):
    """the ends of the fp32 range (the training step's data-gradient convs run on gradients of 1e-7 .. 1e-12: train.hip run_dgrad)"""
    B, C, T, M, taps = 2, 64, 48, 48, 3
    x, w = rnd(13, B, C, T) * sx, rnd(14, M, C, taps, scale=1.0 / math.sqrt(C * taps)) * sw
    ref = conv_ref64(x, w, None, 1, 1, 1)
    close_scaled(lib.op_conv1d(x, w, pad=1), ref, sx * sw, what="conv1d sx=%g sw=%g" % (sx, sw))


@pytest.mark.parametrize("tn,wk", [(16, 0), (32, 1), (32, 2), (32, 8)])
@pytest.mark.parametrize("sx", [1e-8, 1e7])
def test_conv1d_operand_scales_tilings(lib, sx, tn, wk):
    """every K-split and both tile widths carry their own per-wave scale"""
    B, C, T, M, taps = 2, 128, 64, 96, 3
    x, w = rnd(17, B, C, T) * sx, rnd(18, M, C, taps, scale=1.0 / math.sqrt(C * taps))
    ref = conv_ref64(x, w, None, 1, 1, 1)
    lib.set_conv_tiling(wk, tn)
    try:
        close_scaled(lib.op_conv1d(x, w, pad=1), ref, sx, what="conv1d tn=%d wk=%d sx=%g" % (tn, wk, sx))
    finally:
        lib.set_conv_tiling(0, 0)


@pytest.mark.parametrize("wk", [0, 1, 2])
def test_conv1d_channel_blocks_of_different_scale(lib, wk):
    """Channel chunks 14 orders of magnitude apart inside one reduction: the per-wave scale moves DOWN for the large chunk and back UP
    for the small ones, the accumulators follow exactly -- output rows that read only the small channels keep fp32 accuracy relative to
    THEIR magnitude (a single scale per launch, or per tile, would not give that), and the rows that read everything match as well."""
    B, C, T, M = 1, 64, 64, 64
    x = rnd(21, B, C, T)
    cs = torch.tensor([1e-6] * 16 + [1e8] * 16 + [1e-3] * 16 + [1.0] * 16)
    x = x * cs[None, :, None]
    w = rnd(22, M, C, 1, scale=1.0 / math.sqrt(C))
    w[:16, 16:32] = 0.0                      # rows 0..15 never read the 1e8 channels
    w[:8, 48:] = 0.0                         # rows 0..7 read the 1e-6 and 1e-3 channels only
    ref = conv_ref64(x, w)
    lib.set_conv_tiling(wk, 32)
    try:
        got = lib.op_conv1d(x, w).detach().cpu().double()
    finally:
        lib.set_conv_tiling(0, 0)
    assert torch.isfinite(got).all()
    for rows, mag in ((slice(0, 8), 1e-3), (slice(8, 16), 1.0), (slice(16, 64), 1e8)):
        err = (got[:, rows] - ref[:, rows]).abs().max().item()
        assert err <= 2e-5 * mag, "rows %s: max err %.3e at magnitude %g" % (rows, err, mag)


def test_conv1d_small_chunk_inside_an_in_band_slice(lib):
    """The stated limit of the fast mode for raw operands (include/mugd.h): while a wave's K-slice stays inside the band at the fixed scale 2^8
    (largest sample in [2^-6, 2^7)) no chunk is looked at on its own, so a chunk 2^30 below the slice's largest sample is carried with an
    ABSOLUTE error of 2^-36 / 2^8 per sample instead of a relative one: rows that read only those channels come out within ~1e-12 absolute (not
    2e-5 of their 1e-9 magnitude), every other row at the fp32 tolerance.  (A slice that LEAVES the band is redone chunk by chunk: the tests above.)"""
    B, C, T, M = 1, 64, 64, 64
    x = rnd(61, B, C, T)
    x[:, 16:32] *= 2.0 ** -30
    w = rnd(62, M, C, 1, scale=1.0 / math.sqrt(C))
    w[:16, :16] = 0.0
    w[:16, 32:] = 0.0                        # rows 0..15 read the tiny channels only
    ref = conv_ref64(x, w)
    lib.set_conv_tiling(1, 32)
    try:
        got = lib.op_conv1d(x, w).detach().cpu().double()
    finally:
        lib.set_conv_tiling(0, 0)
    assert (got[:, :16] - ref[:, :16]).abs().max().item() <= 1e-12
    assert (got[:, 16:] - ref[:, 16:]).abs().max().item() <= 2e-5


@pytest.mark.parametrize("taps", [1, 3])
@pytest.mark.parametrize("wk", [1, 2, 4])
def test_conv1d_scale_jump_in_the_middle_of_a_long_reduction(lib, wk, taps):
    """A slice that leaves the band at the fixed scale makes the wave redo its tile in the careful mode, whose scale follows the data chunk by chunk
    (conv_body.h: conv_tile, "redo"): a 16-chunk reduction (rotated K order: >= 8 chunks per wave at wk = 1, 2) whose chunks 5
    and 11 are 2^40 larger / smaller than the others, several batch rows and column tiles so that the rotation starts them at different chunks."""
    B, C, T, M = 3, 256, 96, 64
    x = rnd(51, B, C, T)
    x[:, 80:96] *= 2.0 ** 40
    x[:, 176:192] *= 2.0 ** -40
    w = rnd(52, M, C, taps, scale=1.0 / math.sqrt(C * taps))
    w[:32, 80:96] = 0.0                      # rows 0..31 never read the huge chunk: they must come out at THEIR magnitude
    ref = conv_ref64(x, w, None, 1, (taps - 1) // 2)
    lib.set_conv_tiling(wk, 32)
    try:
        got = lib.op_conv1d(x, w, pad=(taps - 1) // 2).detach().cpu().double()
    finally:
        lib.set_conv_tiling(0, 0)
    assert torch.isfinite(got).all()
    for rows, mag in ((slice(0, 32), 1.0), (slice(32, 64), 2.0 ** 40)):
        err = (got[:, rows] - ref[:, rows]).abs().max().item()
        assert err <= 2e-5 * mag, "rows %s: max err %.3e at magnitude %g" % (rows, err, mag)


@pytest.mark.parametrize("mode", ["1", "2"])
@pytest.mark.parametrize("sx", [1e-8, 1e5, 1e7])
def test_conv1d_operand_scales_wide(lib, monkeypatch, sx, mode):
    """the M-split forms share their windows between waves: the window's scale travels with it"""
    monkeypatch.setenv("MUGD_CONV_WIDE", mode)
    B, C, T, M, taps = 2, 160, 72, 160, 3
    x, w = rnd(27, B, C, T) * sx, rnd(28, M, C, taps, scale=1.0 / math.sqrt(C * taps)) * 1e5
    x[:, 32:48] *= 1e6                       # one chunk far above the others: scales differ between the windows of a phase
    ref = conv_ref64(x, w, None, 1, 1, 1)
    lib.set_conv_tiling(0, 32)
    try:
        close_scaled(lib.op_conv1d(x, w, pad=1), ref, sx * 1e5 * 1e6, what="wide conv1d sx=%g" % sx)
    finally:
        lib.set_conv_tiling(0, 0)


@pytest.mark.parametrize("norm,groups,silu", [(1, 32, 1), (1, 32, 2), (2, 0, 0)])
@pytest.mark.parametrize("sg,sw", [(1e-6, 1e5), (1e5, 1e-6), (1e7, 1.0), (1.0, 1e7), (1e-8, 1.0)])
def test_norm_conv1d_operand_scales(lib, sg, sw, norm, groups, silu):
    """GroupNorm / LayerNorm affine parameters and the weights at the ends of the range (the normalised tensor is |gamma| sqrt(n) at
    most, so gamma IS its scale); the input's own scale is normalised away, so it rides along at 1e5."""
    B, C, T, M, taps = 2, 64, 48, 64, 3 if norm == 1 else 1
    x = (rnd(30, B, C, T, scale=2.0) + 0.3) * 1e5
    g, b = (1 + 0.1 * rnd(31, C)) * sg, 0.1 * rnd(32, C) * sg
    w = rnd(33, M, C, taps, scale=1.0 / math.sqrt(C * taps)) * sw
    xd = x.double()
    if norm == 1:
        n = F.group_norm(xd, groups, g.double(), b.double(), eps=1e-6)
        if silu:
            n = F.silu(n)
    else:
        n = F.layer_norm(xd.transpose(1, 2), (C,), g.double(), b.double(), eps=1e-5).transpose(1, 2)
    pad = (taps - 1) // 2
    ref = F.conv1d(n, w.double(), None, 1, pad)
    got = lib.op_norm_conv1d(x, g, b, w, None, dil=1, pad=pad, norm=norm, groups=groups, silu=silu)
    close_scaled(got, ref, sg * sw, atol=3e-5, what="norm+conv sg=%g sw=%g" % (sg, sw))


def test_conv1d_non_finite_operands_propagate(lib):
    """inf / NaN in an operand reach exactly the outputs they reach in fp32 (as non-finite values); every other output stays exact"""
    B, C, T, M = 1, 32, 64, 32
    x, w = rnd(41, B, C, T), rnd(42, M, C, 3, scale=1.0 / math.sqrt(C * 3))
    x[0, 5, 20] = float("inf")
    x[0, 17, 40] = float("nan")
    ref = F.conv1d(x, w, None, 1, 1)
    got = lib.op_conv1d(x, w, pad=1).detach().cpu()
    assert torch.equal(torch.isfinite(got), torch.isfinite(ref))
    fin = torch.isfinite(ref)
    assert (got[fin] - ref[fin]).abs().max().item() <= 2e-5


def test_conv1d_inf_next_to_a_large_finite_sample(lib):
    """An inf inside a chunk must not stop the chunk's FINITE samples from being rescaled (round 6; ADVICE r5: the chunk maximum was inf, no
    scale was derived from it, and a 1e3 next to it overflowed the f16 halves -- 192 non-finite outputs where fp32 has 96): the non-finite
    outputs are exactly fp32's, the outputs that read the large finite sample but not the inf are exact."""
    B, C, T, M = 1, 32, 64, 32
    x, w = rnd(41, B, C, T), rnd(42, M, C, 3, scale=1.0 / math.sqrt(C * 3))
    x[0, 5, 20] = float("inf")
    x[0, 6, 27] = 1e3
    x[0, 9, 50] = -3e4
    ref = F.conv1d(x, w, None, 1, 1)
    for wk in (0, 1, 8):
        lib.set_conv_tiling(wk, 32)
        try:
            got = lib.op_conv1d(x, w, pad=1).detach().cpu()
        finally:
            lib.set_conv_tiling(0, 0)
        assert torch.equal(torch.isfinite(got), torch.isfinite(ref)), (wk, int((~torch.isfinite(got)).sum()), int((~torch.isfinite(ref)).sum()))
        fin = torch.isfinite(ref)
        assert (got[fin] - ref[fin]).abs().max().item() <= 2e-5 * 3e4


@pytest.mark.parametrize("mode", ["0", "1", "2"])
@pytest.mark.parametrize("order", ["big_first", "small_first"])
@pytest.mark.parametrize("big,small", [(1e30, 1e-30), (1e36, 1e-36)])
def test_conv1d_chunks_sixty_orders_apart_every_tile_form(lib, monkeypatch, mode, order, big, small):
    """One 16-channel chunk at 1e30 (1e36) and one at 1e-30 (1e-36) inside one reduction, the rest O(1), in either order, through the K-split
    form and BOTH M-split forms (MUGD_CONV_WIDE = 1 | 2): finite everywhere and the float64 result at the fp32 tolerance of the large terms.
    Round 6: a consumer of a shared window adopted whatever scale the (lagging) parking wave had picked -- the accumulators
    jumped by 2^200 and every output of the M-split forms came out non-finite; now a rise is bounded where the accumulators live
    (conv_body.h: h3_rise_ok) and a window that cannot be reached is dropped through the all-zero window."""
    monkeypatch.setenv("MUGD_CONV_WIDE", mode)
    B, C, T, M = 2, 160, 96, 128
    x = rnd(51, B, C, T)
    w = rnd(52, M, C, 3, scale=1.0 / math.sqrt(C * 3))
    cs = torch.ones(C)
    a, b = (16, 32) if order == "big_first" else (32, 16)
    cs[a:a + 16] = big
    cs[b:b + 16] = small
    x = x * cs[None, :, None]
    ref = conv_ref64(x, w, None, 1, 1, 1)
    lib.set_conv_tiling(0, 32)
    try:
        got = lib.op_conv1d(x, w, pad=1).detach().cpu().double()
    finally:
        lib.set_conv_tiling(0, 0)
    assert torch.isfinite(got).all(), "%d of %d outputs non-finite" % (int((~torch.isfinite(got)).sum()), got.numel())
    assert ((got - ref).abs() / big).max().item() <= 2e-5
    # (what a wave that walks BOTH chunks can promise is an error relative to the largest operand of its walk -- the stated contract of
    # include/mugd.h -- not relative to each output row's own magnitude: rows with zero weights on the large chunk are not held to more here)


@pytest.mark.parametrize("dil,wk", [(2, 1), (2, 2), (1, 1)])
def test_conv1d_tiny_chunk_behind_an_in_band_first_chunk(lib, dil, wk):
    """One wave walks an O(1) chunk (in band at the starting scale: no rescale) and then a chunk at 1e-36, in the data-following mode (dilated
    kernels always; dil = 1: via a first chunk at 1e3, which sends the plain kernel's wave to its careful pass): the rise for the tiny chunk
    is bounded by the scale the accumulators ALREADY hold products at (round 6: that scale was not recorded when the first chunk needed no
    rescale, and the accumulators were lifted by 2^117 -> inf)."""
    B, C, T, M = 1, 48, 64, 32
    x = rnd(71, B, C, T)
    cs = torch.ones(C)
    cs[16:32] = 1e-36
    if dil == 1:
        cs[:16] = 1e3
    x = x * cs[None, :, None]
    w = rnd(72, M, C, 3, scale=1.0 / math.sqrt(C * 3))
    ref = conv_ref64(x, w, None, 1, dil, dil)
    lib.set_conv_tiling(wk, 32)
    try:
        got = lib.op_conv1d(x, w, dil=dil, pad=dil).detach().cpu().double()
    finally:
        lib.set_conv_tiling(0, 0)
    assert torch.isfinite(got).all(), int((~torch.isfinite(got)).sum())
    assert (got - ref).abs().max().item() <= 2e-5 * (1e3 if dil == 1 else 1.0)


def ref_attention(q, k, v, rel, cemb, heads):
    """mug/model/attention.py:98-124 on channel-major tensors."""
    B, C, Tq = q.shape
    Tk = k.shape[2]
    d = C // heads
    qh = q.view(B, heads, d, Tq).transpose(2, 3)
    kh = k.view(B, heads, d, Tk).transpose(2, 3)
    vh = v.view(B, heads, d, Tk).transpose(2, 3)
    pm = (rel.shape[0] - 1) // 2
    idx = (torch.arange(Tk)[None, :] - torch.arange(Tq)[:, None]).clamp(-pm, pm) + pm
    sim = (qh @ kh.transpose(-1, -2) + rel[idx].permute(2, 0, 1)[None]) * d ** -0.5
    attn = sim.softmax(-1) * cemb[idx].permute(2, 0, 1)[None]
    return (attn @ vh).transpose(2, 3).reshape(B, C, Tq)


@pytest.mark.parametrize("B,heads,d,Tq,Tk", [(2, 4, 16, 40, 40), (1, 8, 48, 64, 64), (1, 2, 64, 33, 21), (1, 4, 32, 12, 12),
                                             (1, 2, 32, 100, 100)])
def test_attention(lib, B, heads, d, Tq, Tk):
    C = heads * d
    q, k, v = rnd(14, B, C, Tq), rnd(15, B, C, Tk), rnd(16, B, C, Tk)
    rel, cemb = 0.5 * rnd(17, 129, heads), 1 + 0.3 * rnd(18, 129, heads)
    close(lib.op_attention(q, k, v, rel, cemb, heads), ref_attention(q, k, v, rel, cemb, heads), 3e-5, what="attention")


def s4_params(H, L, seed=0):
    man = [["s.kernel.kernel.%s" % n, sh, dt] for n, sh, dt in [
        ("C", [1, H, 32, 2], "float32"), ("log_dt", [H], "float32"), ("B", [1, H, 32, 2], "float32"),
        ("P", [1, H, 32, 2], "float32"), ("inv_w_real", [H, 32], "float32"), ("w_imag", [H, 32], "float32"),
        ("L", [], "int64")]]
    sd = weights.make_state_dict(man, seed)
    sd["s.kernel.kernel.L"] = torch.tensor(L, dtype=torch.int64)
    return sd


@pytest.mark.parametrize("H,Lint,L", [(32, 32, 32), (32, 64, 48), (64, 96, 96)])
def test_s4_kernel(lib, H, Lint, L):
    sd = s4_params(H, Lint)
    p = "s.kernel.kernel"
    got = lib.op_s4_kernel(sd[p + ".C"], sd[p + ".B"], sd[p + ".P"], sd[p + ".inv_w_real"], sd[p + ".w_imag"],
                           sd[p + ".log_dt"], Lint, L).cpu()
    exact = s4o.s4_kernel(sd, p, L, "exact")
    ref = s4o.s4_kernel(sd, p, L, "reference")
    e_hip, e_ref = (got - exact).abs().max().item(), (ref - exact).abs().max().item()
    # the fp32 HIP evaluation must be as close to the real-number kernel as the reference's own complex64 one
    assert e_hip < max(4 * e_ref, 2e-5), (e_hip, e_ref)
    close(got, ref, 3e-5, what="s4 kernel vs reference arithmetic")


@pytest.mark.parametrize("H,Lint,L", [(32, 32, 32), (64, 96, 96)])
def test_s4_kernel_symmetric_cauchy(lib, H, Lint, L):
    """mugd_set_s4_symmetric(1): the Cauchy sum over both conjugate halves of the poles -- the reference's pykeops backend
    (cauchy_conj, s4.py:55-77: 2 sum (z Re v - Re(v conj w)) / ((z - w)(z - conj w))) written out in torch inside the oracle's
    reference-arithmetic path, and the fp64 evaluation of the same sum; the default (cauchy_naive) must differ from it."""
    sd = s4_params(H, Lint)
    p = "s.kernel.kernel"
    args = (sd[p + ".C"], sd[p + ".B"], sd[p + ".P"], sd[p + ".inv_w_real"], sd[p + ".w_imag"], sd[p + ".log_dt"], Lint, L)
    naive = lib.op_s4_kernel(*args).cpu()
    lib.set_s4_symmetric(True)
    try:
        got = lib.op_s4_kernel(*args).cpu()
    finally:
        lib.set_s4_symmetric(False)
    exact = s4o.s4_kernel(sd, p, L, "exact", symmetric=True)
    ref = s4o.s4_kernel(sd, p, L, "reference", symmetric=True)
    e_hip, e_ref = (got - exact).abs().max().item(), (ref - exact).abs().max().item()
    assert e_hip < max(4 * e_ref, 2e-5), (e_hip, e_ref)
    close(got, ref, 3e-5, what="symmetric s4 kernel vs the pykeops formula")
    assert (got - naive).abs().max().item() > 1e-3 * naive.abs().max().item()      # the two backends are NOT the same function
    assert torch.equal(lib.op_s4_kernel(*args).cpu(), naive)                        # switch restored


@pytest.mark.parametrize("B,H,L", [(2, 32, 40), (1, 48, 300), (5, 8, 64), (2, 8, 128), (3, 4, 256), (1, 4, 512), (2, 4, 12), (1, 4, 1),
                                   (1, 3, 816), (1, 2, 1632), (1, 2, 2048), (1, 2, 2100)])
def test_s4_conv(lib, B, H, L):
    u, k, D = rnd(20, B, H, L), rnd(21, H, L, scale=0.2), rnd(22, H)
    ref = F.gelu(s4o.s4_direct_conv(k, u, D).float())
    close(lib.op_s4_conv(u, k, D), ref, 3e-5, what="s4 conv")
    # and the reference's FFT formulation agrees with the direct one it stands for
    k_f, u_f = torch.fft.rfft(k, n=2 * L), torch.fft.rfft(u, n=2 * L)
    fft = torch.fft.irfft(u_f * k_f[None], n=2 * L)[..., :L] + u * D[None, :, None]
    close(lib.op_s4_conv(u, k, D), F.gelu(fft), 5e-5, what="s4 conv vs FFT form")


@pytest.mark.parametrize("B,H,L,groups", [(3, 64, 64, 32), (2, 32, 128, 8), (1, 64, 256, 32), (5, 32, 512, 32), (2, 32, 40, 8),
                                          (2, 8, 12, 4), (1, 4, 204, 2), (1, 4, 1632, 2), (1, 4, 2100, 2)])
def test_gn_s4_conv(lib, B, H, L, groups):
    """GroupNorm fused into the S4 convolution kernel (in-kernel statistics for L <= 2048, stats kernel + affine beyond)."""
    u, k, D = rnd(23, B, H, L, scale=1.5) + 0.4, rnd(24, H, L, scale=0.2), rnd(25, H)
    g, b = 1 + 0.1 * rnd(26, H), 0.1 * rnd(27, H)
    n = F.group_norm(u, groups, g, b, eps=1e-6)
    ref = F.gelu(s4o.s4_direct_conv(k, n, D).float())
    close(lib.op_gn_s4_conv(u, k, D, g, b, groups), ref, 3e-5, what="GroupNorm + s4 conv")


def test_timestep_embedding(lib):
    t = torch.tensor([1, 21, 501, 981], dtype=torch.long)
    ref = nets.timestep_embedding(t, 128)
    close(lib.op_timestep_embedding(t, 128), ref, 2e-4, rtol=0, what="timestep embedding")


@pytest.mark.parametrize("seconds", [0.37, 1.0])
def test_log_mel(lib, seconds):
    """HIP STFT->mel->log1p->fp16 against the numpy restatement of librosa's algorithm (oracle/host.py)."""
    from oracle import host
    y = host.synth_audio(seconds, seed=3)
    ref = torch.from_numpy(host.log_mel(y).astype(np.float32))
    got = lib.log_mel(torch.from_numpy(y)).cpu()
    assert got.shape == ref.shape == (128, 1 + len(y) // 128)
    # values are fp16-rounded: allow one fp16 ulp (2^-10 relative) on the rare rounding-boundary cases
    err = (got - ref).abs()
    ulp = torch.maximum(ref.abs(), torch.tensor(2.0 ** -14)) * 2.0 ** -10
    assert (err <= ulp * 1.01).all(), float((err / ulp).max())
    assert (err > 0).float().mean().item() < 0.02


@pytest.mark.parametrize("n_fft,hop,nsamp", [(512, 128, 8159), (512, 128, 300), (1024, 256, 5000), (64, 16, 33)])
def test_log_mel_reflect_padding(lib, n_fft, hop, nsamp):
    """librosa <= 0.9 framing (stft(pad_mode='reflect'); the reference leaves librosa unpinned: requirements.txt:8, mug/util.py:138-143) behind
    mugd_set_mel_pad_mode / log_mel(pad_mode=): against oracle/host.py's log_mel(pad_mode='reflect') at the usual one-fp16-ulp criterion --
    every frame, the edge frames in particular; songs barely longer than half a window (300 > 256, 33 > 32 samples: a frame reflects at BOTH
    ends); the per-call override leaves the context's default ('constant') as it was; too-short audio is rejected like librosa does."""
    from oracle import host
    y = (host.synth_audio(nsamp / 22050.0 + 0.01, seed=11)[:nsamp] + 0.3).astype(np.float32)      # (+ DC: the padded edges matter)
    n_mels = 128 if n_fft >= 512 else 16
    for mode in ("reflect", "constant"):
        ref = torch.from_numpy(host.log_mel(y, n_fft=n_fft, hop=hop, n_mels=n_mels, pad_mode=mode).astype(np.float32))
        got = (lib.log_mel(torch.from_numpy(y), n_fft=n_fft, hop=hop, n_mels=n_mels, pad_mode="reflect") if mode == "reflect"
               else lib.log_mel(torch.from_numpy(y), n_fft=n_fft, hop=hop, n_mels=n_mels)).cpu()
        assert got.shape == ref.shape
        err = (got - ref).abs()
        ulp = torch.maximum(ref.abs(), torch.tensor(2.0 ** -14)) * 2.0 ** -10
        assert (err <= ulp * 1.01).all(), (mode, float((err / ulp).max()))
    edge = n_fft // (2 * hop) + 1
    a = host.log_mel(y, n_fft=n_fft, hop=hop, n_mels=n_mels, pad_mode="reflect").astype(np.float32)
    b = host.log_mel(y, n_fft=n_fft, hop=hop, n_mels=n_mels, pad_mode="constant").astype(np.float32)
    assert np.abs(a[:, :edge] - b[:, :edge]).max() > 1e-3              # the two framings really differ at the edges ...
    if a.shape[1] > 2 * edge + 2:
        assert np.array_equal(a[:, edge:-edge], b[:, edge:-edge])     # ... and nowhere else
    lib.set_mel_pad_mode("reflect")
    try:
        got = lib.log_mel(torch.from_numpy(y), n_fft=n_fft, hop=hop, n_mels=n_mels).cpu().numpy()      # the context-wide switch
        assert np.abs(got - a).max() <= np.abs(a).max() * 2.0 ** -10 * 1.01
        with pytest.raises(Exception):
            lib.log_mel(torch.from_numpy(y[: n_fft // 2]), n_fft=n_fft, hop=hop, n_mels=n_mels)
    finally:
        lib.set_mel_pad_mode("constant")


@pytest.mark.parametrize("kind", ["dc", "square30", "sine_fullscale", "quiet"])
@pytest.mark.parametrize("n_fft,hop", [(512, 128), (1024, 256)])
def test_log_mel_loud_and_quiet_audio(lib, kind, n_fft, hop):
    """The mel filterbank GEMM reads the UN-normalised power spectrum: a constant 1.0 PCM has |X[0]|^2 = (n_fft / 2)^2 = 65536 at n_fft = 512
    -- one above the f16 range (round 4: 21 632 of 22 144 mel cells NaN) -- a full-scale sine (n_fft / 4)^2 per bin, a -120 dB signal
    1e-12 of that.  Same one-ulp-of-fp16 criterion as test_log_mel, no non-finite cell anywhere."""
    from oracle import host
    n = 22050 // 2
    t = np.arange(n, dtype=np.float64) / 22050.0
    y = {"dc": np.ones(n), "square30": np.sign(np.sin(2 * np.pi * 30.0 * t) + 1e-9), "sine_fullscale": np.sin(2 * np.pi * 440.0 * t),
         "quiet": 1e-6 * np.sin(2 * np.pi * 440.0 * t)}[kind].astype(np.float32)
    ref = torch.from_numpy(host.log_mel(y, n_fft=n_fft, hop=hop).astype(np.float32))
    got = lib.log_mel(torch.from_numpy(y), n_fft=n_fft, hop=hop).cpu()
    assert torch.isfinite(got).all(), int((~torch.isfinite(got)).sum())
    err = (got - ref).abs()
    ulp = torch.maximum(ref.abs(), torch.tensor(2.0 ** -14)) * 2.0 ** -10
    assert (err <= ulp * 1.01).all(), float((err / ulp).max())


@pytest.mark.parametrize("n_fft,hop,n_mels", [(64, 16, 16), (256, 64, 64), (1024, 256, 128)])
def test_log_mel_other_transform_sizes(lib, n_fft, hop, n_mels):
    """The paired-frame FFT at the other power-of-two sizes the entry point accepts (N < 512: a lane's butterfly slots are
    partly empty; N = 1024: two slot groups per stage and the wide instantiation); an odd frame count leaves the last pair half empty."""
    from oracle import host
    y = host.synth_audio(0.21, seed=5)[:4 * hop * 9]          # 37 frames
    ref = torch.from_numpy(host.log_mel(y, n_fft=n_fft, hop=hop, n_mels=n_mels).astype(np.float32))
    got = lib.log_mel(torch.from_numpy(y), n_fft=n_fft, hop=hop, n_mels=n_mels).cpu()
    assert got.shape == ref.shape == (n_mels, 1 + len(y) // hop)
    err = (got - ref).abs()
    ulp = torch.maximum(ref.abs(), torch.tensor(2.0 ** -14)) * 2.0 ** -10
    assert (err <= ulp * 1.01).all(), float((err / ulp).max())
    assert (err > 0).float().mean().item() < 0.02
