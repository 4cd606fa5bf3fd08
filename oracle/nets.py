"""TEST INFRASTRUCTURE (see oracle/__init__.py).

Functional PyTorch-CPU fp32 restatement of the reference networks on the
sampling hot path.  No nn.Module: every function takes the flat reference
state-dict `sd` (name -> tensor) and a key prefix, so the same weights feed the
oracle, the reference and the HIP library.

Citations are relative to /root/reference.
"""
import math

import torch
import torch.nn.functional as F

from . import s4 as s4o

# ----------------------------------------------------------------------------
# primitive layers
# ----------------------------------------------------------------------------


def group_norm(sd, p, x, groups):
    """mug/model/models.py:10-13  Normalize = GroupNorm(groups, eps=1e-6, affine)."""
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def conv1d(sd, p, x, stride=1, padding=0, dilation=1):
    return F.conv1d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride, padding, dilation)


def linear(sd, p, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def layer_norm(sd, p, x):
    """mug/model/attention.py:136-138  nn.LayerNorm(dim) (eps=1e-5) over the last dim."""
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps=1e-5)


def downsample(sd, p, x):
    """mug/model/models.py:84-88: pad right by one zero, conv k=3 stride 2 pad 0."""
    return conv1d(sd, p + ".conv", F.pad(x, (0, 1)), stride=2)


def upsample(sd, p, x):
    """mug/model/models.py:66-70: nearest x2 then conv k=3 pad 1."""
    return conv1d(sd, p + ".conv", x.repeat_interleave(2, dim=-1), padding=1)


def timestep_embedding(t, dim):
    """mug/model/util.py:156-176 ([cos | sin], half = dim//2, max_period 1e4)."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


# ----------------------------------------------------------------------------
# blocks
# ----------------------------------------------------------------------------


def timestep_resblock(sd, p, x, emb):
    """mug/diffusion/unet.py:212-239 (use_scale_shift_norm=False, no up/down)."""
    h = conv1d(sd, p + ".in_layers.2", F.silu(group_norm(sd, p + ".in_layers.0", x, 32)), padding=1)
    h = h + linear(sd, p + ".emb_layers.1", F.silu(emb))[..., None]
    h = conv1d(sd, p + ".out_layers.3", F.silu(group_norm(sd, p + ".out_layers.0", h, 32)), padding=1)
    if (p + ".skip_connection.weight") in sd:
        x = conv1d(sd, p + ".skip_connection", x)
    return x + h


def resnet_block(sd, p, x, groups, dilations=(1, 1)):
    """mug/model/models.py:142-159 (temb_channels=0 on this path, nin_shortcut 1x1)."""
    h = conv1d(sd, p + ".conv1", F.silu(group_norm(sd, p + ".norm1", x, groups)),
               padding=dilations[0], dilation=dilations[0])
    h = conv1d(sd, p + ".conv2", F.silu(group_norm(sd, p + ".norm2", h, groups)),
               padding=dilations[1], dilation=dilations[1])
    if (p + ".nin_shortcut.weight") in sd:
        x = conv1d(sd, p + ".nin_shortcut", x)
    return x + h


def cross_attention(sd, p, x, context, heads, pos_max=64):
    """mug/model/attention.py:91-126.  x (B,T,C), context (B,Tk,Cc) or None.

    sim = (q k^T + Rel[idx]) * d^-1/2 ; P = softmax(sim) * Cemb[idx] (not
    renormalised) ; idx = clamp(j - i, -64, 64) + 64."""
    ctx = x if context is None else context
    q = linear(sd, p + ".to_q", x)
    k = linear(sd, p + ".to_k", ctx)
    v = linear(sd, p + ".to_v", ctx)
    B, T, C = q.shape
    Tk = k.shape[1]
    d = C // heads
    q = q.view(B, T, heads, d).permute(0, 2, 1, 3)
    k = k.view(B, Tk, heads, d).permute(0, 2, 1, 3)
    v = v.view(B, Tk, heads, d).permute(0, 2, 1, 3)
    idx = (torch.arange(Tk)[None, :] - torch.arange(T)[:, None]).clamp(-pos_max, pos_max) + pos_max
    rel = sd[p + ".relative_position_embedding"][idx].permute(2, 0, 1)  # (h, T, Tk)
    cem = sd[p + ".C_embedding"][idx].permute(2, 0, 1)
    sim = (q @ k.transpose(-1, -2) + rel[None]) * (d ** -0.5)
    attn = sim.softmax(dim=-1) * cem[None]
    out = (attn @ v).permute(0, 2, 1, 3).reshape(B, T, C)
    return linear(sd, p + ".to_out.0", out)


def geglu_ff(sd, p, x):
    """mug/model/attention.py:38-65  GEGLU(dim, 4dim) -> Linear(4dim, dim); exact-erf gelu."""
    a, gate = linear(sd, p + ".net.0.proj", x).chunk(2, dim=-1)
    return linear(sd, p + ".net.2", a * F.gelu(gate))


def contextual_transformer(sd, p, x, context, heads):
    """mug/model/attention.py:186-199 + BasicTransformerBlock._forward :148-152 (depth 1).

    x (B,C,T); context (B,Cc,Tk) or None (then attn2 is a second self-attention)."""
    h = conv1d(sd, p + ".proj_in", group_norm(sd, p + ".norm", x, 32))
    h = h.transpose(1, 2)
    ctx = None if context is None else context.transpose(1, 2)
    b = p + ".transformer_blocks.0"
    h = cross_attention(sd, b + ".attn1", layer_norm(sd, b + ".norm1", h), None, heads) + h
    h = cross_attention(sd, b + ".attn2", layer_norm(sd, b + ".norm2", h), ctx, heads) + h
    h = geglu_ff(sd, b + ".ff", layer_norm(sd, b + ".norm3", h)) + h
    h = conv1d(sd, p + ".proj_out", h.transpose(1, 2))
    return h + x


def s4_layer(sd, p, x, kernel_cache=None):
    """mug/diffusion/unet.py:86-91 -> mug/model/s4.py:1471-1541 (S4.forward)."""
    u = group_norm(sd, p + ".norm", x, 32)
    y = s4o.s4_forward(sd, p + ".s4_model", u, kernel_cache)
    return x + conv1d(sd, p + ".out_layer", y, padding=1)


# ----------------------------------------------------------------------------
# U-Net (mug/diffusion/unet.py:262-550)
# ----------------------------------------------------------------------------

UNET_DEFAULT = dict(in_channels=16, model_channels=128, out_channels=16, num_res_blocks=2,
                    attention_resolutions=[8, 4, 2], channel_mult=[1, 2, 3, 4], num_heads=8,
                    context_dim=128, audio_channels=[256, 512, 512, 512], s4_layer=True)


def unet_plan(cfg):
    """Replays the constructor loops (unet.py:341-487) and returns the module
    list as tuples, index-compatible with the reference's state-dict keys."""
    mc, mult = cfg["model_channels"], cfg["channel_mult"]
    nrb, attn_res = cfg["num_res_blocks"], cfg["attention_resolutions"]
    inp = [("conv_in",)]
    chans = [mc]
    ds = 1
    for level, m in enumerate(mult):
        inp.append(("audio", level))
        for _ in range(nrb):
            layers = ["res"]
            if ds in attn_res:
                layers.append("attn")
            if cfg.get("s4_layer", False):
                layers.append("s4")
            inp.append(("seq", layers))
            chans.append(m * mc)
        if level != len(mult) - 1:
            inp.append(("down",))
            chans.append(m * mc)
            ds *= 2
    out = []
    for level, m in list(enumerate(mult))[::-1]:
        out.append(("audio", level))
        for i in range(nrb + 1):
            layers = ["res"]
            if ds in attn_res:
                layers.append("attn")
            if cfg.get("s4_layer", False) and i != nrb:
                layers.append("s4")
            if level and i == nrb:
                layers.append("up")
                ds //= 2
            out.append(("seq", layers))
    return inp, out


def _run_seq(sd, p, layers, h, emb, ctx, heads, kc):
    for j, kind in enumerate(layers):
        q = "%s.%d" % (p, j)
        if kind == "res":
            h = timestep_resblock(sd, q, h, emb)
        elif kind == "attn":
            h = contextual_transformer(sd, q, h, ctx, heads)
        elif kind == "s4":
            h = s4_layer(sd, q, h, kc)
        elif kind == "up":
            h = upsample(sd, q, h)
    return h


def unet_forward(sd, cfg, x, t, context, audios, prefix="model.unet_model", kernel_cache=None):
    """mug/diffusion/unet.py:511-550.  x (B,16,z), t (B,) long, context (B,128,21),
    audios = list of the wave-encoder maps (the last len(channel_mult) are used)."""
    p = prefix
    heads = cfg["num_heads"]
    nl = len(cfg["channel_mult"])
    emb = timestep_embedding(t, cfg["model_channels"])
    emb = linear(sd, p + ".time_embed.2", F.silu(linear(sd, p + ".time_embed.0", emb)))
    inp, out = unet_plan(cfg)
    hs = []
    h = x
    ai = -nl
    for i, mod in enumerate(inp):
        q = "%s.input_blocks.%d" % (p, i)
        if mod[0] == "audio":
            h = torch.cat([h, audios[ai]], dim=1)
            ai += 1
            continue
        if mod[0] == "conv_in":
            h = conv1d(sd, q + ".0", h, padding=1)
        elif mod[0] == "down":
            h = downsample(sd, q + ".0", h)
        else:
            h = _run_seq(sd, q, mod[1], h, emb, context, heads, kernel_cache)
        hs.append(h)
    ai = -1
    q = p + ".middle_block"
    h = timestep_resblock(sd, q + ".0", h, emb)
    h = contextual_transformer(sd, q + ".1", h, context, heads)
    h = timestep_resblock(sd, q + ".2", h, emb)
    for i, mod in enumerate(out):
        q = "%s.output_blocks.%d" % (p, i)
        if mod[0] == "audio":
            h = torch.cat([h, audios[ai]], dim=1)
            ai -= 1
            continue
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_seq(sd, q, mod[1], h, emb, context, heads, kernel_cache)
    h = F.silu(group_norm(sd, p + ".out.0", h, 32))
    return conv1d(sd, p + ".out.2", h, padding=1)


# ----------------------------------------------------------------------------
# wave encoder (mug/cond/wave.py:398-464)
# ----------------------------------------------------------------------------

WAVE_DEFAULT = dict(n_freq=128, middle_channels=128, attention_resolutions=[128, 256, 512],
                    num_res_blocks=2, num_heads=8, num_groups=32,
                    channel_mult=[1, 1, 1, 1, 2, 2, 2, 4, 4, 4])


def wave_encode(sd, cfg, mel, prefix="model.wave_model"):
    """MelspectrogramScaleEncoder1D.forward: mel (B,128,Ta) -> list of 10 maps."""
    p = prefix
    g = cfg["num_groups"]
    h = conv1d(sd, p + ".conv_in", mel, padding=1)
    hs = []
    ds = 1
    for lvl in range(len(cfg["channel_mult"])):
        q = "%s.down.%d" % (p, lvl)
        if lvl != 0:
            h = downsample(sd, q + ".downsample", h)
            ds *= 2
        for ib in range(cfg["num_res_blocks"]):
            dil = (1, 2) if ib % 2 == 0 else (4, 8)
            h = resnet_block(sd, "%s.block.%d" % (q, ib), h, g, dil)
            if ds in cfg["attention_resolutions"]:
                h = contextual_transformer(sd, "%s.attn.%d" % (q, ib), h, None, cfg["num_heads"])
        hs.append(h)
    return hs


# ----------------------------------------------------------------------------
# VAE decoder (mug/firststage/autoencoder.py:75-77, 268-354)
# ----------------------------------------------------------------------------

VAE_DEFAULT = dict(x_channels=16, middle_channels=64, z_channels=16, num_groups=8,
                   channel_mult=[1, 2, 4, 4], num_res_blocks=1)


def vae_decode(sd, cfg, z, prefix="model.first_stage_model", scale=1.0):
    p = prefix + ".decoder"
    g = cfg["num_groups"]
    h = conv1d(sd, p + ".conv_in", z / scale, padding=1)
    h = resnet_block(sd, p + ".mid.block_1", h, g)
    h = resnet_block(sd, p + ".mid.block_2", h, g)
    nres = len(cfg["channel_mult"])
    for lvl in reversed(range(nres)):
        for ib in range(cfg["num_res_blocks"] + 1):
            h = resnet_block(sd, "%s.up.%d.block.%d" % (p, lvl, ib), h, g)
        if lvl != 0:
            h = upsample(sd, "%s.up.%d.upsample" % (p, lvl), h)
    h = F.silu(group_norm(sd, p + ".norm_out", h, g))
    return conv1d(sd, p + ".conv_out", h, padding=1)


def vae_encode(sd, cfg, x, prefix="model.first_stage_model"):
    """Encoder.forward (mug/firststage/autoencoder.py:244-265) -> the `moments` tensor (B, 2*z_channels, T / 2^(levels-1)) that
    AutoencoderKL.encode wraps in DiagonalGaussianDistribution (:67-73, 356-362): mean = first half, logvar = clamp(second, -10, 20)."""
    p = prefix + ".encoder"
    g = cfg["num_groups"]
    h = conv1d(sd, p + ".conv_in", x, padding=1)
    nres = len(cfg["channel_mult"])
    for lvl in range(nres):
        for ib in range(cfg["num_res_blocks"]):
            h = resnet_block(sd, "%s.down.%d.block.%d" % (p, lvl, ib), h, g)
        if lvl != nres - 1:
            h = downsample(sd, "%s.down.%d.downsample" % (p, lvl), h)
    h = resnet_block(sd, p + ".mid.block_1", h, g)
    h = resnet_block(sd, p + ".mid.block_2", h, g)
    h = F.silu(group_norm(sd, p + ".norm_out", h, g))
    return conv1d(sd, p + ".conv_out", h, padding=1)


# ----------------------------------------------------------------------------
# prompt embedding (mug/cond/feature.py:15-21)
# ----------------------------------------------------------------------------


def cond_embed(sd, ids, prefix="model.cond_stage_model"):
    """ids (B,21) -> (B,128,21)."""
    return sd[prefix + ".embedding.weight"][ids.long()].transpose(1, 2)
