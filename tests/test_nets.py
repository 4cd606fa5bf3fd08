"""Network-level parity through the C ABI: U-Net forward, DDIM loop (+CFG), VAE decode and wave
encoder against (a) the golden fixtures produced by the REAL reference (tests/golden, see
oracle/gen_golden.py) and (b) the oracle on fresh seeded inputs.

`tiny` (a structurally complete miniature) runs on the emulated build in the CPU suite and on
the GPU; `full` (the shipped configs/mug/mug_diffusion.yaml shapes, 151 M parameters) runs on
the MI355X only.  Tolerances: fp32 level -- conv / Linear products run on split-f16 operands with fp32 accumulation (csrc/conv_body.h: H3,
fp32-equivalent at every operand scale: tests/test_ops.py operand-scale tests), attention / S4 / norms in fp32 -- so summation order and
last-bit rounding differ from PyTorch-CPU; the note grid (logit > 0) must match bit-for-bit except where the reference logit itself is
inside the measured logit error (and in any case within GRID_EPS of the threshold): grid_check prints every such cell."""
import os

import numpy as np
import pytest
import torch

from oracle import cases, host, nets, sampler, weights

G = cases.GOLDEN
GRID_EPS = 2e-3


def golden(name):
    return np.load(os.path.join(G, name))


_sd_cache = {}


def state_dict(case, z):
    key = (case["name"], z)
    if key not in _sd_cache:
        _sd_cache.clear()
        man = weights.load_manifest(os.path.join(G, case["manifest"]))
        _sd_cache[key] = weights.set_s4_lengths(weights.make_state_dict(man, seed=0), case["unet"], z)
    return _sd_cache[key]


_net_cache = {}


def native(lib, case, z, kind):
    key = (id(lib), case["name"], z, kind)
    if key not in _net_cache:
        sd = state_dict(case, z)
        if kind == "unet":
            n = lib.unet(case["unet"])
            n.set_params(sd, "model.unet_model.")
        elif kind == "vae":
            n = lib.vae(case["vae"])
            n.set_params(sd, "model.first_stage_model.")
        else:
            n = lib.wave(case["wave"])
            n.set_params(sd, "model.wave_model.")
        _net_cache[key] = n
    return _net_cache[key]


def same(a, b):
    """Two runs of the same program: bit-identical except for the last-bit effect of the fp64 row-sum atomics' order on the
    GroupNorm statistics -- a few ulp at most."""
    a, b = a.detach().cpu(), b.detach().cpu()
    return a.shape == b.shape and (a - b).abs().max().item() <= 2e-6 * max(1.0, b.abs().max().item())


def report(what, got, ref):
    got = got.detach().cpu()
    d = (got - ref).abs().max().item()
    s = ref.abs().max().item()
    print("%s: max|diff| %.3e  (ref max %.3e, rel %.2e)" % (what, d, s, d / max(s, 1e-30)))
    return d, s


CASES = [pytest.param(cases.TINY, 32, id="tiny"),
         pytest.param(cases.FULL, 96, id="full", marks=pytest.mark.gpu),
         pytest.param(cases.FULL, 512, id="full-z512", marks=pytest.mark.gpu)]     # the headline length: the reference's own outputs


def lib_for(request_lib, case):
    if case["name"] == "full" and request_lib.device.type != "cuda":
        pytest.skip("the full-size networks only run on the GPU build")
    return request_lib


@pytest.mark.parametrize("case,z", CASES)
@pytest.mark.parametrize("B", [1, 2])
def test_unet_forward_vs_reference_golden(lib, case, z, B):
    lib = lib_for(lib, case)
    if not os.path.exists(os.path.join(cases.GOLDEN, "%s_unet_z%d_b%d.npz" % (case["name"], z, B))):
        pytest.skip("no reference fixture for this batch size at this length")
    g = golden("%s_unet_z%d_b%d.npz" % (case["name"], z, B))
    seed = int(g["seed"])
    x, t = cases.x_T(seed, B, z), torch.from_numpy(g["t"])
    c, w = cases.context(case, seed, B), cases.audio_maps(case, seed, B, z)
    ref = torch.from_numpy(g["eps"])
    sd = state_dict(case, z)
    # the oracle reproduces the reference's own output ...
    assert (nets.unet_forward(sd, case["unet"], x, t, c, w) - ref).abs().max().item() < 1e-5
    # ... and the HIP path matches it
    got = native(lib, case, z, "unet").forward(x, t, c, w)
    d, s = report("unet %s z=%d B=%d" % (case["name"], z, B), got, ref)
    assert d < 2e-4 * max(1.0, s)


@pytest.mark.parametrize("sx,sa", [(1e6, 1e5), (1e-6, 1e-5), (1e4, 1e-6)])
@pytest.mark.parametrize("case,z", CASES[:2])
def test_unet_forward_at_input_scales_outside_the_f16_range(lib, case, z, sx, sa):
    """The domain of conv_gemm's split-f16 arithmetic end to end (csrc/conv_body.h: "The DOMAIN of H3"): a latent and audio feature maps far above
    65504 / far below 2^-14 enter the U-Net through RAW operands (the input conv, every skip conv over [h | audio], the first GroupNorm only
    normalises what follows) -- the fp32 oracle (= the reference's arithmetic, /root/reference/mug/diffusion/unet.py:27-33) has no trouble with
    them, round 4's kernels returned NaN.  Same network-level tolerance as the golden test, no non-finite value."""
    lib = lib_for(lib, case)
    B, seed = 2, 77
    x, t = cases.x_T(seed, B, z) * sx, torch.full((B,), 481, dtype=torch.long)
    c, w = cases.context(case, seed, B), [m * sa for m in cases.audio_maps(case, seed, B, z)]
    sd = state_dict(case, z)
    ref = nets.unet_forward(sd, case["unet"], x, t, c, w)
    assert torch.isfinite(ref).all()
    got = native(lib, case, z, "unet").forward(x, t, c, w)
    assert torch.isfinite(got.detach().cpu()).all()
    d, s = report("unet %s z=%d latent x %g, audio x %g" % (case["name"], z, sx, sa), got, ref)
    assert d < 2e-4 * max(1.0, s)


@pytest.mark.parametrize("case,z", CASES)
def test_vae_decode_vs_reference_golden(lib, case, z):
    lib = lib_for(lib, case)
    g = golden("%s_vae_z%d.npz" % (case["name"], z))
    zlat = cases.randn(int(g["seed"]), 2, (2, 16, z))
    ref = torch.from_numpy(g["logits"])
    got = native(lib, case, z, "vae").decode(zlat)
    d, s = report("vae %s" % case["name"], got, ref)
    assert d < 2e-4 * max(1.0, s)


@pytest.mark.parametrize("case,z", CASES[:2])
def test_vae_encode_vs_reference_golden(lib, case, z):
    """AutoencoderKL.encode (inpainting row): moments against the real reference's, and the oracle agrees bit for bit."""
    lib = lib_for(lib, case)
    g = golden("%s_vaeenc_z%d.npz" % (case["name"], z))
    up = 2 ** (len(case["vae"]["channel_mult"]) - 1)
    x = cases.randn(int(g["seed"]), 3, (2, case["vae"]["x_channels"], z * up))
    ref = torch.from_numpy(g["moments"])
    sd = state_dict(case, z)
    assert torch.equal(nets.vae_encode(sd, case["vae"], x), ref)
    enc = lib.vae(case["vae"], encoder=True)
    enc.set_params(sd, "model.first_stage_model.")
    got = enc.vae_encode(x)
    d, s = report("vae encode %s" % case["name"], got, ref)
    assert d < 2e-4 * max(1.0, s)
    assert np.allclose(got[:, :ref.shape[1] // 2].cpu().numpy(), g["mode"], atol=2e-4 * max(1.0, s))


@pytest.mark.parametrize("case,z", CASES)
def test_wave_encoder_vs_reference_golden(lib, case, z):
    lib = lib_for(lib, case)
    g = golden("%s_wave_z%d.npz" % (case["name"], z))
    mel = cases.mel_input(case, int(g["seed"]), 1, z * case["audio_ratio"])
    nl = len(case["unet"]["channel_mult"])
    outs = native(lib, case, z, "wave").encode(mel)
    for i in range(nl):
        ref = torch.from_numpy(g["w%d" % i])
        d, s = report("wave %s map -%d" % (case["name"], nl - i), outs[len(outs) - nl + i], ref)
        assert d < 3e-4 * max(1.0, s)
    am = np.array([o.abs().mean().item() for o in outs])
    assert np.allclose(am, g["absmean"], rtol=1e-3)


@pytest.mark.parametrize("case,z", CASES)
def test_s4_kernels_vs_reference_module_output(lib, case, z):
    """SSKernelNPLR.forward (s4.py:706-832) of the first and last S4 layer of the U-Net at full size: k0 / k1 are the REFERENCE
    module's own outputs (oracle/gen_golden.py), k*_exact the real-number kernel in fp64."""
    lib = lib_for(lib, case)
    g = golden("%s_s4kernel_z%d.npz" % (case["name"], z))
    sd = state_dict(case, z)
    for i, name in enumerate(g["names"]):
        p = "model.unet_model." + str(name)
        Lint = int(sd[p + ".L"])
        got = lib.op_s4_kernel(sd[p + ".C"], sd[p + ".B"], sd[p + ".P"], sd[p + ".inv_w_real"], sd[p + ".w_imag"],
                               sd[p + ".log_dt"], Lint, g["k%d" % i].shape[1]).cpu()
        ref, exact = torch.from_numpy(g["k%d" % i]), torch.from_numpy(g["k%d_exact" % i])
        e_hip, e_ref = (got - exact).abs().max().item(), (ref - exact).abs().max().item()
        d, s_ = report("s4 kernel %s L=%d" % (name, ref.shape[1]), got, ref)
        assert e_hip < max(4 * e_ref, 2e-5), (e_hip, e_ref)
        assert d < 3e-5 * max(1.0, s_)


def run_ddim(lib, case, z, S, B, scale, g):
    sd = state_dict(case, z)
    seed = int(g["seed"])
    ids = torch.from_numpy(g["ids"])
    c = nets.cond_embed(sd, ids)
    assert torch.equal(lib.cond_embed(sd["model.cond_stage_model.embedding.weight"], ids).cpu(), c)
    fy = cases.feature_yaml()
    uc = nets.cond_embed(sd, torch.tensor([host.feature_ids({}, fy)] * B, dtype=torch.float32))
    mel = cases.mel_input(case, int(g["mel_seed"]), 1, z * case["audio_ratio"])
    wave = native(lib, case, z, "wave")
    nl = len(case["unet"]["channel_mult"])
    w = wave.encode(mel)[-nl:]          # batch 1: the B seeds share one copy of the audio features
    xT = cases.x_T(seed, B, z)
    steps = sampler.ddim_step_scalars(sd["alphas_cumprod"].numpy(), S, 0.0)
    ts = [s["t"] for s in steps]
    sched = [[s["a_t"], s["a_prev"], s["sigma"], s["sqrt_1m_at"]] for s in steps]
    unet = native(lib, case, z, "unet")
    lat = unet.ddim_sample(xT, c, w, ts, sched, uc=uc if scale != 1.0 else None, scale=scale)
    logits = native(lib, case, z, "vae").decode(lat)
    return lat, logits


def grid_check(logits, ref_logits):
    """The note grid is a hard threshold on the decoder's logits (/root/reference/mug/data/convertor.py:212-216), so two correct fp32
    implementations may differ exactly in the cells whose logit is smaller than their logit error.  Prints the logit error over the grid
    rows, how many cells sit inside it (the only ones that CAN differ), and every cell that does; asserts that no cell flips outside
    the error bound and none where the reference logit is GRID_EPS or more away from the threshold."""
    got = logits.detach().cpu().numpy()
    gs, gh = host.note_grid(got)
    rs, rh = host.note_grid(ref_logits)
    rows = list(range(0, 4)) + list(range(8, 12))                      # start rows, hold rows (note_grid's thresholded channels)
    refv = ref_logits[..., rows, :]
    gotv = got[..., rows, :]
    flips = np.concatenate([(gs != rs), (gh != rh)], axis=-2)
    assert flips.shape == refv.shape, (flips.shape, refv.shape)
    err = float(np.abs(gotv - refv).max())
    at_risk = int((np.abs(refv) <= err).sum())
    n_flip = int(flips.sum())
    worst = float(np.abs(refv[flips]).max()) if n_flip else 0.0
    print("note grid: %d / %d cells flipped (largest |ref logit| among them %.2e); logit error over the grid rows %.2e, cells with |ref logit| inside it: %d, "
          "smallest |ref logit| %.2e" % (n_flip, flips.size, worst, err, at_risk, float(np.abs(refv).min())))
    for idx in np.argwhere(flips)[:20]:
        i = tuple(int(v) for v in idx)
        print("    flipped cell %s: ref logit %+.3e, got %+.3e" % (i, float(refv[i]), float(gotv[i])))
    assert worst <= err and n_flip <= at_risk, "a cell flipped outside the logit error bound"
    assert worst < GRID_EPS, "a note cell flipped although the reference logit is not near the threshold"
    return n_flip


@pytest.mark.parametrize("scale", [1.0, 5.0])
def test_ddim_tiny_vs_reference_golden(lib, scale):
    case, z, S, B = cases.TINY, 32, 4, 2
    g = golden("tiny_ddim_z32_S4_B2_cfg%g.npz" % scale)
    lat, logits = run_ddim(lib, case, z, S, B, scale, g)
    d, s = report("ddim tiny cfg=%g latent" % scale, lat, torch.from_numpy(g["latent"]))
    assert d < 1e-3 * max(1.0, s)
    grid_check(logits, g["logits"])


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [1.0, 5.0])
def test_ddim_full_vs_reference_golden(gpu_lib, scale):
    """BASELINE.json configs[0]: 30 s audio (z=96), 10 DDIM steps, batch 1 -- the reference's own output."""
    case, z, S, B = cases.FULL, 96, 10, 1
    g = golden("full_ddim_z96_S10_B1_cfg%g.npz" % scale)
    lat, logits = run_ddim(gpu_lib, case, z, S, B, scale, g)
    d, s = report("ddim full cfg=%g latent" % scale, lat, torch.from_numpy(g["latent"]))
    assert d < 2e-3 * max(1.0, s)
    grid_check(logits, g["logits"])


REDUCED_EPS = 0.05        # reduced-precision mode: a note cell may flip only where the reference logit is this close to the threshold
REDUCED_FLIP_FRACTION = 0.005


@pytest.mark.parametrize("case,z,S,B", [pytest.param(cases.TINY, 32, 4, 2, id="tiny"),
                                        pytest.param(cases.FULL, 96, 10, 1, id="full", marks=pytest.mark.gpu)])
def test_reduced_precision_mode_bf16_weights(lib, case, z, S, B):
    """mugd_set_weight_precision(1): packed conv / linear weights in bfloat16, everything else fp32 (include/mugd.h).  NOT the
    reference's arithmetic and never the default; reported separately.  Bound ("fp16 tolerance" metric): against the
    REFERENCE's own DDIM output, the latent stays within 2 % of its range, at most 0.5 % of the note cells flip, and none of
    them where the reference logit is farther than REDUCED_EPS from the threshold."""
    lib = lib_for(lib, case)
    g = golden("%s_ddim_z%d_S%d_B%d_cfg1.npz" % (case["name"], z, S, B))
    _net_cache.clear()
    lib.set_weight_precision(True)
    try:
        lat, logits = run_ddim(lib, case, z, S, B, 1.0, g)
    finally:
        lib.set_weight_precision(False)
        _net_cache.clear()
    d, s_ = report("ddim %s bf16 weights latent" % case["name"], lat, torch.from_numpy(g["latent"]))
    assert 1e-6 * s_ < d < 2e-2 * max(1.0, s_)               # differs from fp32 (the mode is really on) but stays close
    got = logits.detach().cpu().numpy()
    gs, gh = host.note_grid(got)
    rs, rh = host.note_grid(g["logits"])
    flips = np.concatenate([(gs != rs).ravel(), (gh != rh).ravel()])
    refv = np.concatenate([g["logits"][..., 0:4, :].ravel(), g["logits"][..., 8:12, :].ravel()])
    worst = float(np.abs(refv[flips]).max()) if flips.any() else 0.0
    print("bf16-weight mode: %d / %d note cells flipped, largest |ref logit| among them %.2e" % (int(flips.sum()), flips.size, worst))
    assert flips.mean() <= REDUCED_FLIP_FRACTION and worst < REDUCED_EPS


def test_graph_and_eager_agree(lib):
    case, z, S, B = cases.TINY, 32, 4, 1
    g = golden("tiny_ddim_z32_S4_B2_cfg1.npz")
    sd = state_dict(case, z)
    c = nets.cond_embed(sd, torch.from_numpy(g["ids"])[:1])
    w = cases.audio_maps(case, 5, B, z)
    xT = cases.x_T(9, B, z)
    steps = sampler.ddim_step_scalars(sd["alphas_cumprod"].numpy(), S, 1.0)
    ts = [s["t"] for s in steps]
    sched = [[s["a_t"], s["a_prev"], s["sigma"], s["sqrt_1m_at"]] for s in steps]
    noise = torch.stack([cases.randn(77, i, (B, 16, z)) for i in range(len(ts))])
    unet = native(lib, case, z, "unet")
    lib.set_graph_mode(1)                      # one hipGraph per step, replayed
    a, pa = unet.ddim_sample(xT, c, w, ts, sched, noise=noise, want_pred_x0=True)
    lib.set_graph_mode(2)                      # the whole loop as one graph
    a2, pa2 = unet.ddim_sample(xT, c, w, ts, sched, noise=noise, want_pred_x0=True)
    lib.set_graph_mode(0)                      # eager launches: the default
    b, pb = unet.ddim_sample(xT, c, w, ts, sched, noise=noise, want_pred_x0=True)
    assert same(a, b) and same(pa, pb) and same(a2, b) and same(pa2, pb)
    # eta = 1 path against the oracle with the same explicit noise
    ref = sampler.ddim_sample(sd, case["unet"], S, c, w, xT, eta=1.0, noise=list(noise))
    d, s = report("ddim eta=1", a, ref)
    assert d < 1e-3 * max(1.0, s)


@pytest.mark.gpu
@pytest.mark.parametrize("z,B", [(512, 4), (1632, 1)])
def test_full_unet_at_benchmark_and_long_lengths_vs_oracle(gpu_lib, z, B):
    """BASELINE configs[1] (3-min audio, z = 512, batch 4: the shape bench.py times, with every fused path active --
    GroupNorm / LayerNorm from producer sums, 16-wide tiles, in-kernel S4 GroupNorm) and configs[3] as the reference
    actually runs it (10-min audio = the whole sequence at z = 1632) against the oracle on the same seeded inputs."""
    case = cases.FULL
    sd = state_dict(case, z)
    x, t = cases.x_T(11, B, z), torch.full((B,), 481, dtype=torch.long)
    c, w = cases.context(case, 11, B), cases.audio_maps(case, 11, 1, z)
    ref = nets.unet_forward(sd, case["unet"], x, t, c, [m.repeat(B, 1, 1) for m in w])
    got = native(gpu_lib, case, z, "unet").forward(x, t, c, w)          # audio maps shared by the batch rows (audio_batch 1)
    d, s = report("unet full z=%d B=%d vs oracle" % (z, B), got, ref)
    assert d < 2e-4 * max(1.0, s)
    _net_cache.clear()


@pytest.mark.gpu
def test_headline_workload_end_to_end_vs_oracle(gpu_lib):
    """BASELINE configs[1] in full: 3-minute audio (32768 mel frames, z = 512), 50 DDIM steps, batch 4, no guidance --
    wave encoder -> the whole DDIM loop (one native call, graph replay) -> VAE decode -> note grid, against the oracle
    on the same seeded weights and inputs (the oracle needs ~10 s of the box's host cores for its 50 U-Net evaluations)."""
    case, z, S, B = cases.FULL, 512, 50, 4
    sd = state_dict(case, z)
    fy = cases.feature_yaml()
    prompts = [{"sr": 4.0, "rank_status": "ranked"}, {"sr": 2.5, "ln_ratio": 0.4}, {"sr": 6.0}, {}]
    ids = torch.tensor([host.feature_ids(p, fy) for p in prompts], dtype=torch.float32)
    c = nets.cond_embed(sd, ids)
    mel = cases.mel_input(case, 29, 1, z * case["audio_ratio"])
    xT = cases.x_T(29, B, z)
    nl = len(case["unet"]["channel_mult"])
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, threads))           # the oracle's small-tensor ops collapse on a 256-thread pool
    try:
        w_ref = nets.wave_encode(sd, case["wave"], mel)[-nl:]
        lat_ref = sampler.ddim_sample(sd, case["unet"], S, c, [m.repeat(B, 1, 1) for m in w_ref], xT)
        logits_ref = nets.vae_decode(sd, case["vae"], lat_ref)
    finally:
        torch.set_num_threads(threads)
    w = native(gpu_lib, case, z, "wave").encode(mel)[-nl:]
    steps = sampler.ddim_step_scalars(sd["alphas_cumprod"].numpy(), S, 0.0)
    lat = native(gpu_lib, case, z, "unet").ddim_sample(xT, c, w, [s["t"] for s in steps],
                                                       [[s["a_t"], s["a_prev"], s["sigma"], s["sqrt_1m_at"]] for s in steps])
    logits = native(gpu_lib, case, z, "vae").decode(lat)
    d, s = report("headline ddim z=512 S=50 B=4 latent", lat, lat_ref)
    assert d < 2e-3 * max(1.0, s)
    flips = grid_check(logits, logits_ref.numpy())
    print("headline workload: %d of %d note cells differ" % (flips, B * 8 * z * 8))
    _net_cache.clear()


@pytest.mark.gpu
def test_ten_minute_audio_end_to_end_vs_oracle(gpu_lib):
    """BASELINE configs[3] as the reference actually runs it (no chunking exists -- a 10-minute audio is ONE sequence,
    webui.py:349-367): 104 448 mel frames -> wave encoder -> z = 1632, 10 DDIM steps, batch 1 -> VAE decode at 13 056 frames ->
    note grid, against the oracle on the same seeded weights and inputs (long-sequence attention, S4 at L = 1632, every conv at
    its longest T)."""
    case, z, S, B = cases.FULL, 1632, 10, 1
    sd = state_dict(case, z)
    fy = cases.feature_yaml()
    ids = torch.tensor([host.feature_ids({"sr": 5.0, "ln_ratio": 0.2}, fy)], dtype=torch.float32)
    c = nets.cond_embed(sd, ids)
    mel = cases.mel_input(case, 31, 1, z * case["audio_ratio"])
    xT = cases.x_T(31, B, z)
    nl = len(case["unet"]["channel_mult"])
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, threads))
    try:
        w_ref = nets.wave_encode(sd, case["wave"], mel)[-nl:]
        lat_ref = sampler.ddim_sample(sd, case["unet"], S, c, w_ref, xT)
        logits_ref = nets.vae_decode(sd, case["vae"], lat_ref)
    finally:
        torch.set_num_threads(threads)
    w = native(gpu_lib, case, z, "wave").encode(mel)[-nl:]
    for i in range(nl):
        d, s_ = report("10-min wave map -%d" % (nl - i), w[i], w_ref[i])
        assert d < 3e-4 * max(1.0, s_)
    steps = sampler.ddim_step_scalars(sd["alphas_cumprod"].numpy(), S, 0.0)
    lat = native(gpu_lib, case, z, "unet").ddim_sample(xT, c, w, [s["t"] for s in steps],
                                                       [[s["a_t"], s["a_prev"], s["sigma"], s["sqrt_1m_at"]] for s in steps])
    logits = native(gpu_lib, case, z, "vae").decode(lat)
    d, s_ = report("10-min ddim z=1632 S=10 latent", lat, lat_ref)
    assert d < 2e-3 * max(1.0, s_)
    assert logits.shape[-1] == 8 * z
    flips = grid_check(logits, logits_ref.numpy())
    print("10-minute workload: %d of %d note cells differ" % (flips, 8 * z * 8))
    _net_cache.clear()


def test_one_handle_follows_changing_lengths_and_batches(lib):
    """webui.py:349-367 mutates z_length per audio and `count` per request: one native handle is recompiled for each
    (batch, length) and must give the same numbers as a fresh handle (program cache, row-sum block, baked S4 kernels)."""
    case = cases.TINY
    zs = [(32, 2), (64, 1), (32, 2), (32, 1)]
    sd64 = state_dict(case, 64)                      # S4 buffers long enough for both lengths
    shared = lib.unet(case["unet"])
    shared.set_params(sd64, "model.unet_model.")
    first = {}
    for z, B in zs:
        x, t = cases.x_T(3, B, z), torch.full((B,), 301, dtype=torch.long)
        c, w = cases.context(case, 3, B), cases.audio_maps(case, 3, 1, z)
        got = shared.forward(x, t, c, w).cpu()
        fresh = lib.unet(case["unet"])
        fresh.set_params(sd64, "model.unet_model.")
        ref = fresh.forward(x, t, c, w).cpu()
        assert same(got, ref), (z, B)
        if (z, B) in first:
            assert same(got, first[(z, B)])
        first[(z, B)] = got
        orc = nets.unet_forward(sd64, case["unet"], x, t, c, [m.repeat(B, 1, 1) for m in w])
        assert (got - orc).abs().max().item() < 2e-4 * max(1.0, orc.abs().max().item())


def test_host_enqueue_hook_replays_the_compiled_program(lib):
    """mugd_net_host_enqueue (include/mugd.h): wall clock around back-to-back enqueues of the last compiled program -- it reports the program's
    op count and a positive time, refuses a network that has not run yet, and leaves the program's results as they were (the program includes
    its own accumulator reset)."""
    case = cases.TINY
    z, B = 32, 2
    sd = state_dict(case, z)
    n = lib.unet(case["unet"])
    n.set_params(sd, "model.unet_model.")
    with pytest.raises(Exception):
        n.host_enqueue(1)
    x = cases.x_T(1, B, z)
    t = torch.tensor([501, 21], dtype=torch.long)
    c = cases.context(case, 1, B)
    w = [m.repeat(B, 1, 1) for m in cases.audio_maps(case, 1, 1, z)]
    a = n.forward(x, t, c, w).detach().cpu()
    us, ops = n.host_enqueue(3)
    assert ops > 10 and us > 0.0
    b = n.forward(x, t, c, w).detach().cpu()
    assert (a - b).abs().max().item() <= 2e-6 * max(1.0, a.abs().max().item())
    n.close()


_GN_FORMS_SCRIPT = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "mug-diffusion_amd")); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from oracle import cases, weights
import conftest
lib = conftest.real_lib() if %(gpu)r else conftest.emu_lib()
case, z, B = (cases.FULL, 96, 2) if %(gpu)r else (cases.TINY, 32, 2)
man = weights.load_manifest(os.path.join(cases.GOLDEN, case["manifest"]))
sd = weights.set_s4_lengths(weights.make_state_dict(man, 0), case["unet"], z)
unet = lib.unet(case["unet"]); unet.set_params(sd, "model.unet_model.")
x, t = cases.x_T(3, B, z), torch.full((B,), 417, dtype=torch.long)
c, w = cases.context(case, 3, B), cases.audio_maps(case, 3, B, z)
outs = [unet.forward(x, t, c, w).detach().cpu().numpy() for _ in range(2)]          # twice: the accumulators are cleared per evaluation
np.save(sys.argv[1], np.stack(outs))
"""


def _gn_forms(tmp_path, gpu):
    """U-Net forward in two fresh processes (the switches are read once per process): group tables + column sums forced on / off; returns both outputs and the logs."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ("1", "0"):
        out = str(tmp_path / ("eps_%s.npy" % mode))
        env = dict(os.environ, MUGD_GN_GROUP=mode, MUGD_LN_SUMS=mode, MUGD_GN_GROUP_LOG="1")
        p = subprocess.run([sys.executable, "-c", _GN_FORMS_SCRIPT % {"root": root, "gpu": gpu}, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[mode] = (np.load(out), p.stderr)
    return res


def _check_gn_forms(res):
    (g, glog), (r, rlog) = res["1"], res["0"]
    assert "group table:" in glog and "accumulate group sums only" in glog       # the tables are in use, some producers dropped their row sums
    assert "column sums:" in glog                                                # ... and so are the LayerNorm column sums
    assert "group table:" not in rlog and "column sums:" not in rlog
    scale = max(1.0, float(np.abs(r).max()))
    assert float(np.abs(g[0] - g[1]).max()) <= 2e-6 * scale                      # second evaluation = first: the tables are cleared with the row sums
    assert float(np.abs(g - r).max()) <= 2e-6 * scale                            # fp64 sums in another order: last fp32 bits of a statistic at most


def test_unet_group_tables_agree_with_row_sums(lib, tmp_path):
    """ConvArgs::gn_table / gsink and ConvArgs::colsum (round 6): the producers' tiles add GROUP sums / finished column sums and the consuming
    GroupNorm / LayerNorm loads them, instead of mapping, fetching and reducing the producers' row sums / per-row-tile column parts behind
    workgroup barriers -- same statistics, same network output."""
    if lib.device.type == "cuda":
        pytest.skip("the GPU variant of this test runs the full-size network")
    _check_gn_forms(_gn_forms(tmp_path, False))


@pytest.mark.gpu
def test_unet_group_tables_agree_with_row_sums_gpu(gpu_lib, tmp_path):
    _check_gn_forms(_gn_forms(tmp_path, True))
