"""The drop-in boundary: the `mug` package (mug-diffusion_amd/mug) must expose the reference's module
surface -- constructible from the YAML `target:` strings, state-dict key/shape/dtype identical to the
reference's (fixtures tests/golden/manifest_*.json were dumped from the real reference), and its
sample / decode / encode entry points must reproduce the reference's outputs through libmugd."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import cases, host, weights

G = cases.GOLDEN


def model_config(case):
    """The reference's configs/mug/mug_diffusion.yaml shape (model.params), for `case`'s sizes."""
    return dict(target="mug.diffusion.diffusion.DDPM", params=dict(
        linear_start=0.0001, linear_end=0.02, log_every_t=100, timesteps=1000, z_channels=16, z_length=512,
        parameterization="eps", loss_type="smooth_l1", monitor="val/loss_simple",
        unet_config=dict(target="mug.diffusion.unet.UNetModel",
                         params=dict(dropout=0.0, lstm_last=False, lstm_layer=False, use_checkpoint=False, **case["unet"])),
        first_stage_config=dict(target="mug.firststage.autoencoder.AutoencoderKL",
                                params=dict(monitor="val/loss", kl_weight=1e-6, ddconfig=dict(case["vae"]),
                                            lossconfig=dict(target="torch.nn.Identity"))),
        cond_stage_config=dict(target="mug.cond.feature.BeatmapFeatureEmbedder",
                               params=dict(path_to_yaml=os.path.join(G, "mania_beatmap_features.yaml"),
                                           embed_dim=case["unet"]["context_dim"])),
        wave_stage_config=dict(target="mug.cond.wave.MelspectrogramScaleEncoder1D",
                               params=dict(dropout=0.0, use_checkpoint=True, **case["wave"]))))


def build(case):
    from mug.util import instantiate_from_config
    return instantiate_from_config(model_config(case)).eval()


@pytest.mark.parametrize("case", [cases.TINY, cases.FULL], ids=["tiny", "full"])
def test_state_dict_matches_reference_manifest(case):
    model = build(case)
    man = weights.load_manifest(os.path.join(G, case["manifest"]))
    got = weights.manifest_of(model.state_dict())
    assert len(got) == len(man)
    assert got == man, [(a, b) for a, b in zip(got, man) if a != b][:5]      # same keys, order, shapes, dtypes
    if case is cases.FULL:
        assert len(man) == 1515
    # and the reference's loading idiom (webui.py:57) works
    sd = weights.make_state_dict(man, seed=0)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected


def test_schedule_buffers_and_ddim_arrays_match_reference():
    from mug.diffusion.ddim import DDIMSampler
    model = build(cases.TINY)
    ref = json.load(open(os.path.join(G, "host_golden.json")))
    s = DDIMSampler(model)
    for key, g in ref["ddim"].items():
        S, eta = int(key.split("_")[0][1:]), float(key.split("eta")[1])
        if S == 7:
            continue            # 1000 // 7 steps index past the schedule in the reference too (ddim.py:31 / utils.py:54)
        s.make_schedule(S, ddim_eta=eta, verbose=False)
        assert s.ddim_timesteps.tolist() == g["ts"]
        assert np.array_equal(np.asarray(s.ddim_alphas, dtype=np.float64), np.asarray(g["alphas"]))
        assert np.array_equal(np.asarray(s.ddim_alphas_prev, dtype=np.float64), np.asarray(g["alphas_prev"]))
        assert np.allclose(s.ddim_sigmas, g["sigmas"], rtol=0, atol=0 if eta == 0 else 1e-7)


def test_prompt_ids_match_reference():
    from mug.util import feature_dict_to_embedding_ids, count_beatmap_features
    fy = cases.feature_yaml()
    ref = json.load(open(os.path.join(G, "host_golden.json")))
    assert count_beatmap_features(fy) == ref["table_rows"] == 329
    for row in ref["prompts"]:
        assert feature_dict_to_embedding_ids(row["prompt"], fy) == row["ids"]


def test_s4_host_setup_C_matches_reference():
    """mug/model/s4.py (length growth C -> C~, doubling) against the reference's own module states."""
    from mug.model import s4 as s4host
    g = np.load(os.path.join(G, "s4_setup_C.npz"))
    p = {n: torch.from_numpy(g["init_" + n].copy()) for n in ["C", "B", "P", "inv_w_real", "w_imag", "log_dt"]}
    p["L"] = torch.tensor(int(g["init_L"]), dtype=torch.int64)
    assert s4host.ensure_length_(p, 24)
    assert int(p["L"]) == int(g["L_after24"]) == 24
    assert np.abs(p["C"].numpy() - g["C_after24"]).max() < 2e-5 * np.abs(g["C_after24"]).max()
    assert s4host.ensure_length_(p, 40)
    assert int(p["L"]) == int(g["L_after40"]) == 48
    assert np.abs(p["C"].numpy() - g["C_after40"]).max() < 2e-5 * np.abs(g["C_after40"]).max()
    assert not s4host.ensure_length_(p, 40)
    # fresh-model initialisation: spectrum and magnitudes of the HiPPO-LegS NPLR form (eigenvector phases are
    # LAPACK-dependent, so B and P are compared up to the per-mode phase)
    w, P, B = s4host.hippo_legs_nplr(64)
    assert np.allclose(np.log(-w.real), g["init_inv_w_real"][0], atol=1e-5)
    assert np.allclose(w.imag, g["init_w_imag"][0], rtol=1e-5, atol=1e-4)
    refB = g["init_B"][0, 0, :, 0] + 1j * g["init_B"][0, 0, :, 1]
    refP = g["init_P"][0, 0, :, 0] + 1j * g["init_P"][0, 0, :, 1]
    assert np.allclose(np.abs(B), np.abs(refB), rtol=1e-4) and np.allclose(np.abs(P), np.abs(refP), rtol=1e-4)
    assert np.allclose(B * P.conj(), refB * refP.conj(), rtol=1e-3, atol=1e-4)


@pytest.fixture
def as_default_lib(lib):
    """Makes the library under test the one the `mug` modules use (the emulated build on CPU)."""
    import mug._native as N
    old = N._default
    N._default = lib
    yield lib
    N._default = old


@pytest.mark.parametrize("scale", [1.0, 5.0])
def test_python_surface_end_to_end_tiny(as_default_lib, scale):
    """webui.py:startMapping's call sequence on the drop-in classes: cond_stage_model -> wave_model ->
    DDIMSampler.sample -> model.decode, against the reference's own outputs."""
    from mug.diffusion.ddim import DDIMSampler
    lib = as_default_lib
    case, z, S, B = cases.TINY, 32, 4, 2
    g = np.load(os.path.join(G, "tiny_ddim_z32_S4_B2_cfg%g.npz" % scale))
    model = build(case)
    man = weights.load_manifest(os.path.join(G, case["manifest"]))
    sd = weights.set_s4_lengths(weights.make_state_dict(man, seed=0), case["unet"], z)
    model.load_state_dict(sd)
    model = model.to(lib.device)
    fy = cases.feature_yaml()
    ids = torch.from_numpy(g["ids"]).to(lib.device)
    c = model.model.cond_stage_model(ids)
    uc = model.model.cond_stage_model(torch.tensor([host.feature_ids({}, fy)] * B, dtype=torch.float32, device=lib.device))
    mel = cases.mel_input(case, int(g["mel_seed"]), 1, z * case["audio_ratio"]).to(lib.device)
    w = model.model.wave_model(torch.cat([mel] * B))          # webui.py:369-374 stacks `count` copies
    assert len(w) == len(case["wave"]["channel_mult"])
    model.z_length = z
    sampler = DDIMSampler(model, lib.device)                   # webui.py:105 passes the device positionally
    seen = []
    samples, inter = sampler.sample(S=S, c=c, w=w, batch_size=B, shape=None, verbose=False, eta=0.0,
                                    x_T=cases.x_T(int(g["seed"]), B, z).to(lib.device),
                                    unconditional_guidance_scale=scale, unconditional_conditioning=uc,
                                    tqdm_class=_GradioShapedTqdm(seen))
    logits = model.model.decode(samples)
    d = (samples.cpu() - torch.from_numpy(g["latent"])).abs().max().item()
    assert d < 1e-3 * max(1.0, float(np.abs(g["latent"]).max())), d
    gs, gh = host.note_grid(logits.cpu().numpy())
    rs, rh = host.note_grid(g["logits"])
    assert (gs == rs).all() and (gh == rh).all()
    assert seen == ["open %d" % S] + ["step"] * S + ["closed"] and len(inter["x_inter"]) == 3 and len(inter["pred_x0"]) == 3


class _GradioShapedTqdm:
    """Stands in for `gradio.Progress().tqdm` (webui.py:388): signature tqdm(iterable, desc=None, total=None, ...) with the
    iterable REQUIRED, and the returned object is only iterated -- there is no update(), and close() needs an argument.
    Records how the sampler drives it."""

    def __init__(self, seen):
        self.seen = seen

    def __call__(self, iterable, desc=None, total=None, unit="steps"):
        seen = self.seen
        items = list(iterable)
        seen.append("open %d" % (total if total is not None else len(items)))

        class _It:
            def __iter__(self_inner):
                for _ in items:
                    seen.append("step")
                    yield _
                seen.append("closed")

            def close(self_inner, _tqdm):            # gradio's signature: unusable without the argument
                raise AssertionError("close() must not be called on a foreign progress object")

        return _It()


def test_native_module_refuses_wrong_device(as_default_lib):
    model = build(cases.TINY)
    if as_default_lib.device.type == "cpu":
        pytest.skip("device mismatch can only be provoked against the GPU build")
    with pytest.raises(RuntimeError):
        model.model.first_stage_model.decode(torch.zeros(1, 16, 32))


def test_convertor_threshold_contract():
    from mug.data.convertor import OsuManiaConvertor, BeatmapMeta
    conv = OsuManiaConvertor(frame_ms=128 / 22050 * 8 * 1000, max_frame=4096, from_logits=True)
    a = np.random.default_rng(0).standard_normal((16, 200)).astype(np.float32)
    assert conv.array_to_objects(a, BeatmapMeta(cs=4)) == host.array_to_objects(a, conv.frame_ms)
    gs, gh = conv.note_grid(a)
    assert (gs == (a[0:4] > 0)).all() and (gh == (a[8:12] > 0)).all()


@pytest.mark.skipif(not os.path.isdir("/root/reference/mug"), reason="needs the reference checkout (authoring container only)")
def test_non_hot_path_modules_fall_through_to_the_reference():
    """the drop-in shadows only the hot path; e.g. mug.lr_scheduler stays the reference's file."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(1, '/root/reference'); "
            "import mug.diffusion.unet as u, mug.lr_scheduler as l; print(u.__file__); print(l.__file__)"
            % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mug-diffusion_amd"))
    out = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True, check=True).stdout.split()
    assert "mug-diffusion_amd" in out[0] and out[1].startswith("/root/reference/")


def test_sampler_call_paths_agree_and_inpainting_round_trip(as_default_lib):
    """The three ways DDIMSampler drives the library give the same chart: one native call (default), per-step calls
    (callbacks, ddim.py:146-152) and the masked path (ddim.py:141-144) with mask = 0; with mask = 1 every step restarts from
    q_sample(x0), so the result depends on x0 only through the encoder path: AutoencoderKL.encode(...).mode() feeds it."""
    from mug.diffusion.ddim import DDIMSampler
    lib = as_default_lib
    case, z, S, B = cases.TINY, 32, 4, 2
    g = np.load(os.path.join(G, "tiny_ddim_z32_S4_B2_cfg1.npz"))
    model = build(case)
    man = weights.load_manifest(os.path.join(G, case["manifest"]))
    model.load_state_dict(weights.set_s4_lengths(weights.make_state_dict(man, seed=0), case["unet"], z))
    model = model.to(lib.device)
    c = model.model.cond_stage_model(torch.from_numpy(g["ids"]).to(lib.device))
    mel = cases.mel_input(case, int(g["mel_seed"]), 1, z * case["audio_ratio"]).to(lib.device)
    w = model.model.wave_model(mel)
    model.z_length = z
    sampler = DDIMSampler(model)
    xT = cases.x_T(int(g["seed"]), B, z).to(lib.device)
    kw = dict(S=S, c=c, w=w, batch_size=B, verbose=False, eta=0.0, x_T=xT, tqdm_class=_GradioShapedTqdm([]))
    one, _ = sampler.sample(**kw)
    steps_seen = []
    per, inter = sampler.sample(callback=steps_seen.append, **kw)
    close_ = lambda a, b: (a.cpu() - b.cpu()).abs().max().item() <= 2e-6 * max(1.0, b.abs().max().item())   # fp64-atomic order: last bit
    assert steps_seen == list(range(S)) and close_(one, per)
    assert len(inter["x_inter"]) == 3
    # inpainting plumbing: the posterior mode of a decoded chart is a latent of the right shape; mask = 0 leaves sampling unchanged
    logits = model.model.decode(one)
    x0 = model.model.first_stage_model.encode(logits).mode()
    assert x0.shape == one.shape and torch.isfinite(x0).all()
    masked, _ = sampler.sample(mask=torch.zeros_like(one), x0=x0, **kw)
    assert close_(masked, one)
    torch.manual_seed(0)
    full, _ = sampler.sample(mask=torch.ones_like(one), x0=x0, **kw)
    assert full.shape == one.shape and torch.isfinite(full).all() and not close_(full, one)
    # generator state after a sample: the reference's p_sample_ddim draws randn(shape) on every step even with eta = 0
    # (ddim.py:192), so a following x_T = None sample starts from the state "S draws later" -- reproduce that
    dev = one.device
    torch.manual_seed(123)
    sampler.sample(**kw)
    after = torch.randn(4, device=dev)
    torch.manual_seed(123)
    for _ in range(S):
        torch.randn(tuple(one.shape), device=dev)
    assert torch.equal(after, torch.randn(4, device=dev))


@pytest.mark.parametrize("which", ["tiny", pytest.param("full", marks=pytest.mark.gpu)])
def test_masked_sampling_vs_reference_golden(as_default_lib, monkeypatch, which):
    """Inpainting path (ddim.py:141-144) against the REAL reference's DDIMSampler.sample(mask=, x0=) output (oracle/gen_golden.py
    --masked-only): same seed, same generator consumption (q_sample's randn_like, then the step's randn, every step).  The
    reference drew from the CPU generator; on the GPU the draws are routed through it too so the streams coincide."""
    from mug.diffusion.ddim import DDIMSampler
    lib = as_default_lib
    case, z, S, B = (cases.TINY, 32, 4, 2) if which == "tiny" else (cases.FULL, 96, 10, 1)
    if which == "full" and lib.device.type != "cuda":
        pytest.skip("the full-size networks only run on the GPU build")
    g = np.load(os.path.join(G, "%s_ddim_masked_z%d_S%d_B%d.npz" % (case["name"], z, S, B)))
    model = build(case)
    man = weights.load_manifest(os.path.join(G, case["manifest"]))
    model.load_state_dict(weights.set_s4_lengths(weights.make_state_dict(man, seed=0), case["unet"], z))
    model = model.to(lib.device)
    dev = lib.device
    if dev.type == "cuda":
        real_randn, real_like = torch.randn, torch.randn_like
        monkeypatch.setattr(torch, "randn", lambda *a, device=None, **k: real_randn(*a, **k).to(device) if device is not None else real_randn(*a, **k))
        monkeypatch.setattr(torch, "randn_like", lambda t, **k: real_randn(t.shape, dtype=t.dtype).to(t.device))
    c = model.model.cond_stage_model(torch.from_numpy(g["ids"]).to(dev))
    mel = cases.mel_input(case, int(g["mel_seed"]), 1, z * case["audio_ratio"]).to(dev)
    w = model.model.wave_model(mel)
    seed = int(g["seed"])
    xT, x0 = cases.x_T(seed, B, z).to(dev), cases.randn(seed, 5, (B, 16, z)).to(dev)
    mask = torch.zeros(B, 16, z, device=dev)
    mask[:, :, : z // 2] = 1.0
    model.z_length = z
    torch.manual_seed(seed)
    lat, _ = DDIMSampler(model).sample(S=S, c=c, w=w, batch_size=B, eta=0.0, verbose=False, x_T=xT, mask=mask, x0=x0,
                                       tqdm_class=lambda it, **k: it)
    ref = torch.from_numpy(g["latent"])
    d = (lat.cpu() - ref).abs().max().item()
    print("masked ddim %s: latent max|diff| %.3e (ref max %.3e)" % (which, d, ref.abs().max().item()))
    assert d < 2e-3 * max(1.0, ref.abs().max().item())
    logits = model.model.decode(lat).cpu().numpy()
    gs, gh = host.note_grid(logits)
    rs, rh = host.note_grid(g["logits"])
    flips = np.concatenate([(gs != rs).ravel(), (gh != rh).ravel()])
    refv = np.concatenate([g["logits"][..., 0:4, :].ravel(), g["logits"][..., 8:12, :].ravel()])
    assert not flips.any() or np.abs(refv[flips]).max() < 2e-3


def test_audio_ingest_resamples_on_the_device_when_only_soundfile_is_present(as_default_lib, monkeypatch, tmp_path):
    """mug.util.load_audio_without_cache (mug/util.py:133-144 of the reference) without librosa: the file is decoded by
    soundfile at its native rate (stereo 44.1 kHz here), mixed down, cut to max_duration, converted to 22.05 kHz by
    mugd_resample_poly and turned into the fp16 log-mel by mugd_log_mel -- equal to the oracle's host pipeline."""
    import sys
    import types
    import mug.util as U
    sr_file, seconds = 44100, 1.5
    g = np.random.default_rng(3)
    t = np.arange(int(sr_file * seconds)) / sr_file
    stereo = np.stack([0.4 * np.sin(2 * np.pi * 440 * t) + 0.05 * g.standard_normal(len(t)),
                       0.3 * np.sin(2 * np.pi * 1320 * t)], axis=1).astype(np.float32)
    fake = types.ModuleType("soundfile")
    fake.read = lambda path, dtype="float32", always_2d=True: (stereo.copy(), sr_file)
    monkeypatch.setitem(sys.modules, "soundfile", fake)
    monkeypatch.setitem(sys.modules, "librosa", None)            # `import librosa` raises ImportError
    got = U.load_audio_without_cache(str(tmp_path / "song.ogg"), n_mels=128, audio_hop_length=128, n_fft=512, sr=22050,
                                     max_duration=1.0)
    mono = stereo.mean(axis=1)[: int(1.0 * sr_file)]
    want = host.log_mel(host.resample_poly(mono, 22050, sr_file))
    assert got.dtype == np.float16 and got.shape == want.shape == (128, 1 + (len(mono) // 2) // 128)
    # fp16 log-mel bins may round differently where the fp32 value sits on a rounding boundary: allow 1 fp16 ulp on <= 2 % of bins (as tests/test_ops.py::test_log_mel)
    diff = np.abs(got.astype(np.float32) - want.astype(np.float32))
    ulp = np.spacing(np.abs(want).astype(np.float16)).astype(np.float32)
    assert (diff <= ulp).all() and (diff > 0).mean() < 0.02, (diff.max(), (diff > 0).mean())


# ------------------------------------------------------------------------------------------------------------------------------
# the reference's own headless caller, unchanged, on the drop-in
# ------------------------------------------------------------------------------------------------------------------------------
_MAPPING = "/root/reference/scripts/mapping.py"


def _sections(path):
    """[TimingPoints] and [HitObjects] of a written chart (the metadata lines carry names the two callers choose differently)."""
    out, cur = {}, None
    for line in open(path, encoding="utf8").read().splitlines():
        if line.startswith("["):
            cur = line.strip()
            out[cur] = []
        elif cur and line.strip():
            out[cur].append(line)
    return out["[TimingPoints]"], out["[HitObjects]"]


@pytest.mark.skipif(not os.path.exists(_MAPPING), reason="needs the reference checkout (authoring container only)")
def test_reference_cli_runs_unchanged_on_the_drop_in(tmp_path, monkeypatch):
    """`python scripts/mapping.py ...` of the REFERENCE (scripts/mapping.py:308-522, run through runpy as __main__, not a line
    changed) with the drop-in `mug` package ahead of it on sys.path: OmegaConf.load(models/ckpt/model.yaml) ->
    instantiate_from_config -> load_state_dict(strict=False) -> model.cuda() -> load_audio_without_cache -> wave_model ->
    DDIMSampler.sample -> decode -> parse_osu_file / save_osu_file(gridify=mini-jack pass + gridify).  Stubs: eyed3 and omegaconf
    only (tests/refcaller_stubs; neither is installed).  Test-side adaptations, all outside the caller: the emulated library is
    the process default, nn.Module.cuda is the identity (no GPU in this container), and DDIMSampler.sample is wrapped to seed the
    global RNG the caller draws x_T from (mapping.py has no seed argument).  The charts it writes must equal, in their
    [TimingPoints] and [HitObjects], the ones mug.job produces for the same seed (one sample) and the ones a direct use of the
    drop-in API produces for the same x_T (two samples)."""
    import runpy
    import sys
    import wave as wavmod
    import yaml
    from conftest import emu_lib, PKG
    import mug._native as N
    from mug import job
    from mug.diffusion.ddim import DDIMSampler
    case, z, S, seed = cases.TINY, 32, 4, 4321
    lib = emu_lib()
    monkeypatch.setattr(N, "_default", lib)
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, device=None: self)
    # ---- the working directory the caller expects: models/ckpt/model.{yaml,ckpt}, a WAV, a template, the feature YAML
    n_fft, sr, ratio = 512, 22050, case["audio_ratio"]
    frames = z * ratio
    cfg = dict(version="test", model=model_config(case),
               data=dict(params=dict(common_params=dict(n_fft=n_fft, sr=sr, n_mels=case["wave"]["n_freq"], max_audio_frame=frames,
                                                         audio_note_window_ratio=max(1, ratio // 2 ** (len(case["vae"]["channel_mult"]) - 1))))))
    cfg["model"]["params"]["z_length"] = z
    (tmp_path / "models" / "ckpt").mkdir(parents=True)
    (tmp_path / "models" / "ckpt" / "model.yaml").write_text(yaml.safe_dump(cfg))
    man = weights.load_manifest(os.path.join(G, case["manifest"]))
    sd = weights.set_s4_lengths(weights.make_state_dict(man, seed=0), case["unet"], z)
    torch.save({"state_dict": sd}, tmp_path / "models" / "ckpt" / "model.ckpt")
    pcm = (0.4 * np.sin(2 * np.pi * 440 * np.arange(sr) / sr) + 0.2 * np.random.default_rng(3).standard_normal(sr)).clip(-1, 1)
    with wavmod.open(str(tmp_path / "song.wav"), "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr)
        f.writeframes((pcm * 32767).astype("<i2").tobytes())
    from test_osu_io import TEMPLATE
    (tmp_path / "template.osu").write_text(TEMPLATE, encoding="utf-8")
    fy_path = os.path.join(G, "mania_beatmap_features.yaml")
    (tmp_path / "prompts").mkdir()
    for i in (1, 2):                                   # --prompt_dir has a default the caller always reads (mapping.py:422-425)
        (tmp_path / "prompts" / ("feature_%d.yaml" % i)).write_text("{}\n")

    orig_sample = DDIMSampler.sample

    def seeded(self, *a, **k):
        torch.manual_seed(seed)
        return orig_sample(self, *a, **k)
    monkeypatch.setattr(DDIMSampler, "sample", seeded)
    monkeypatch.chdir(tmp_path)
    stubs = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refcaller_stubs")
    monkeypatch.setattr(sys, "path", [PKG, stubs] + [p for p in sys.path if "refstubs" not in p])
    for m in ("omegaconf", "eyed3"):
        monkeypatch.delitem(sys.modules, m, raising=False)

    def run_cli(n_samples, outdir):
        monkeypatch.setattr(sys, "argv", ["mapping.py", "--audio", "song.wav", "--feature_yaml", fy_path, "--template_beatmap", "template.osu",
                                          "--prompt_dir", "prompts", "--outdir", outdir, "--ddim_steps", str(S), "--n_samples", str(n_samples)])
        ns = runpy.run_path(_MAPPING, run_name="__main__")
        assert "mug-diffusion_amd" in sys.modules["mug.diffusion.ddim"].__file__ and ns["DDIMSampler"] is DDIMSampler
        d = os.path.join(outdir, "Test Artist - Test Song")
        return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".osu"))

    one = run_cli(1, "out1")
    two = run_cli(2, "out2")
    assert len(one) == 1 and len(two) == 2 and os.path.exists(os.path.join("out1", "Test Artist - Test Song", "audio.mp3"))
    monkeypatch.setattr(DDIMSampler, "sample", orig_sample)

    # ---- (a) mug.job for the same seed: one unit whose x_T comes from a CPU generator seeded like the caller's global RNG
    from mug.util import instantiate_from_config, load_audio_without_cache
    model = instantiate_from_config(cfg["model"]).eval()
    model.load_state_dict(sd, strict=False)
    hop = n_fft // 4
    mel = torch.from_numpy(load_audio_without_cache("song.wav", case["wave"]["n_freq"], hop, n_fft, sr, hop / sr * frames).astype(np.float32))
    cp = cfg["data"]["params"]["common_params"]
    frame_ms = hop / sr * cp["audio_note_window_ratio"] * 1000
    on_chart, written = job.chart_writer("job_out", "template.osu", frame_ms, frames // cp["audio_note_window_ratio"], lib=lib)
    with open(fy_path) as f:
        fy = yaml.safe_load(f)
    # the caller pads / truncates to max_audio_frame and samples at model.z_length; run_job's length rule gives the same z when the
    # mel is exactly max_audio_frame - 1 frames long or shorter by less than one latent block: hand it the padded mel and pin z
    sampler = DDIMSampler(model)
    units = [job.Unit(audio=0, seed=seed, prompt={})]
    monkeypatch.setattr(job, "z_length_for", lambda *a, **k: z)
    job.run_job(model, sampler, units, lambda a: mel[:, :frames], fy, steps=S, batch=1, max_audio_frame=frames, z_length_cfg=z, on_chart=on_chart)
    assert _sections(written[0]) == _sections(one[0])
    assert len(_sections(one[0])[1]) > 10

    # ---- (b) two samples: the same x_T through the drop-in API directly
    from mug.util import feature_dict_to_embedding_ids
    x_T = torch.randn((2, 16, z), generator=torch.Generator().manual_seed(seed))
    ids = torch.tensor([feature_dict_to_embedding_ids({}, fy)] * 2, dtype=torch.float32)
    m2 = job.fit_mel(mel, z, ratio)
    w = model.model.wave_model(torch.stack([m2, m2]))
    model.z_length = z
    lat, _ = sampler.sample(S=S, c=model.model.cond_stage_model(ids), w=w, batch_size=2, shape=None, verbose=False, x_T=x_T, eta=0.0)
    logits = model.model.decode(lat)
    on2, written2 = job.chart_writer("api_out", "template.osu", frame_ms, frames // cp["audio_note_window_ratio"], lib=lib)
    for i in range(2):
        on2(i, job.Unit(audio=0, seed=seed + i, prompt={}), logits[i])
    assert [_sections(p) for p in written2] == [_sections(p) for p in two]


# ------------------------------------------------------------------------------------------------------------------------------
# the reference's web UI flow, unchanged, on the drop-in
# ------------------------------------------------------------------------------------------------------------------------------
_WEBUI = "/root/reference/webui.py"


@pytest.mark.skipif(not os.path.exists(_WEBUI), reason="needs the reference checkout (authoring container only)")
def test_reference_webui_start_mapping_runs_unchanged_on_the_drop_in(tmp_path, monkeypatch):
    """`webui.py` of the REFERENCE imported through runpy (not a line changed; run_name != "__main__", so the Gradio Blocks are not
    built) with the drop-in `mug` package ahead of it on sys.path, then its `startMapping` called the way the Generate button calls it
    (webui.py:277-482): module level OmegaConf.load -> load_model_from_config (instantiate_from_config from the star imports,
    load_state_dict(strict=False)) -> model.to(device) -> DDIMSampler(model, device); per click: seeds, generate_feature_dict ->
    parse_feature twice (uc because scale = 5 != 1) -> load_audio_without_cache -> the length rule that MUTATES
    dataset.max_audio_frame and model.z_length (webui.py:349-367; the clip is cut so that z_length really changes: 64 -> 32) -> `count`
    stacked copies of one audio through wave_model -> sampler.sample(..., unconditional_guidance_scale=5.0, unconditional_conditioning=uc,
    tqdm_class=progress.tqdm) = U-Net batch 2 x count -> decode -> parse_osu_file / save_osu_file with ITS gridify order (gridify, then
    the mini-jack pass) -> the .osz.  Stubs (tests/refcaller_stubs, none of them installed here): gradio, minacalc, reamber, audioread,
    eyed3, omegaconf; `requests.get` raises (no network: webui.py's own try / except swallows it) and a failing `ffmpeg` stands on PATH
    (webui.py then copies the audio).  The charts it writes must equal the ones a direct use of the drop-in API produces for the same
    seed, and mug.job's logits at scale = 5.0 for the same units written in the web UI's post-processing order."""
    import runpy
    import sys
    import wave as wavmod
    import yaml
    from conftest import emu_lib, PKG
    import mug._native as N
    from mug import job
    from mug.diffusion.ddim import DDIMSampler
    case, S, seed, count, scale = cases.TINY, 4, 777, 2, 5.0
    z_cfg, z = 64, 32                                  # configured latent length / the one the length rule picks for the short clip
    lib = emu_lib()
    monkeypatch.setattr(N, "_default", lib)
    n_fft, sr, ratio = 512, 22050, case["audio_ratio"]
    hop = n_fft // 4
    nw_ratio = max(1, ratio // 2 ** (len(case["vae"]["channel_mult"]) - 1))
    cfg = dict(version="test", model=model_config(case),
               data=dict(params=dict(common_params=dict(n_fft=n_fft, sr=sr, n_mels=case["wave"]["n_freq"], max_audio_frame=z_cfg * ratio,
                                                         audio_note_window_ratio=nw_ratio))))
    cfg["model"]["params"]["z_length"] = z_cfg
    (tmp_path / "models" / "ckpt").mkdir(parents=True)
    (tmp_path / "models" / "ckpt" / "model.yaml").write_text(yaml.safe_dump(cfg))
    man = weights.load_manifest(os.path.join(G, case["manifest"]))
    sd = weights.set_s4_lengths(weights.make_state_dict(man, seed=0), case["unet"], z)
    torch.save({"state_dict": sd}, tmp_path / "models" / "ckpt" / "model.ckpt")
    n_pcm = hop * (z * ratio - 40)                     # t = z * ratio - 39 mel frames: int(t / ratio / 32) + 1 = 1 -> z = 32
    pcm = (0.4 * np.sin(2 * np.pi * 440 * np.arange(n_pcm) / sr) + 0.2 * np.random.default_rng(3).standard_normal(n_pcm)).clip(-1, 1)
    with wavmod.open(str(tmp_path / "song.wav"), "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(sr)
        f.writeframes((pcm * 32767).astype("<i2").tobytes())
    from test_osu_io import TEMPLATE
    (tmp_path / "asset").mkdir()
    (tmp_path / "asset" / "template.osu").write_text(TEMPLATE, encoding="utf-8")
    (tmp_path / "asset" / "bg.jpg").write_bytes(b"jpg")
    fy_path = os.path.join(G, "mania_beatmap_features.yaml")
    (tmp_path / "configs" / "mug").mkdir(parents=True)
    (tmp_path / "configs" / "mug" / "mania_beatmap_features.yaml").write_text(open(fy_path).read())
    (tmp_path / "bin").mkdir()
    (tmp_path / "bin" / "ffmpeg").write_text("#!/bin/sh\nexit 1\n")
    os.chmod(tmp_path / "bin" / "ffmpeg", 0o755)
    monkeypatch.setenv("PATH", str(tmp_path / "bin") + os.pathsep + os.environ.get("PATH", ""))
    monkeypatch.chdir(tmp_path)
    stubs = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refcaller_stubs")
    monkeypatch.setattr(sys, "path", [PKG, stubs] + [p for p in sys.path if "refstubs" not in p])
    for m in ("omegaconf", "eyed3", "gradio", "minacalc", "reamber", "audioread", "audioread.ffdec"):
        monkeypatch.delitem(sys.modules, m, raising=False)
    import requests

    def no_network(*a, **k):
        raise requests.exceptions.ConnectionError("no network in the authoring container")
    monkeypatch.setattr(requests, "get", no_network)

    ns = runpy.run_path(_WEBUI, run_name="webui_under_test")
    assert "mug-diffusion_amd" in sys.modules["mug.diffusion.ddim"].__file__ and ns["DDIMSampler"] is DDIMSampler
    model = ns["model"]
    assert model.z_length == z_cfg and isinstance(ns["sampler"], DDIMSampler)

    class Upload:                                       # gr.File value: webui.py:289 reads .name
        name = str(tmp_path / "song.wav")
    import gradio
    progress = gradio.Progress()
    off = dict(cjs=False, cj="more", cjss=False, cjsc=20.0, stas=False, sta="more", stass=False, stasc=20.0, sss=True, ss="more chordjack", ssss=False, sssc=20.0,
               jss=False, js="more", jsss=False, jssc=20.0, hss=False, hs="more", hsss=False, hssc=20.0, jsps=False, jsp="more", jspss=False, jspsc=20.0,
               techs=False, tech="more", techss=False, techsc=20.0)
    args = dict(audioPath=Upload(), audioTitle="Test Song", audioArtist="Test Artist", rss=True, rs="ranked/stable", srs=True, sr=4.0, etts=False, ett=20.0,
                mts=True, lnrs=True, mapType="Hybrid (both)", lnr=0.3, count=count, step=S, scale=scale, rm_jack_interval=90, auto_snap=True, seed=seed,
                progress=progress, **off)
    out = ns["startMapping"](**args)
    assert model.z_length == z                          # the length rule rewrote it (webui.py:356)
    assert isinstance(out, list) and len(out) == 5 and out[1]["value"].endswith("Test Artist - Test Song.osz") and os.path.exists(out[1]["value"])
    descs = [d for d, _ in progress.calls]
    assert descs[0] == "Process prompts and audio" and "Post process charts" in descs and len(descs) == 3       # + the sampler's step loop
    save_dir = os.path.join("outputs", "beatmaps", "Test Artist - Test Song")
    charts = sorted(os.path.join(save_dir, f) for f in os.listdir(save_dir) if f.endswith(".osu"))
    assert len(charts) == count and os.path.exists(os.path.join(save_dir, "audio.wav")) and os.path.exists(os.path.join(save_dir, "bg.jpg"))
    assert len(_sections(charts[0])[1]) >= 5 and _sections(charts[0]) != _sections(charts[1])

    # ---- the same click through the drop-in API directly: same prompt ids, same seed -> same x_T, same post-processing order
    from mug.data import convertor
    from mug.data.utils import gridify, remove_intractable_mania_mini_jacks
    from mug.util import feature_dict_to_embedding_ids, load_audio_without_cache
    with open(fy_path) as f:
        fy = yaml.safe_load(f)
    fd, _ = ns["generate_feature_dict"](*[args[k] for k in ("audioPath", "audioTitle", "audioArtist")],
                                        *[args[k] for k in "rss rs srs sr etts ett cjs cj cjss cjsc stas sta stass stasc sss ss ssss sssc jss js jsss jssc hss hs hsss hssc "
                                                           "jsps jsp jspss jspsc techs tech techss techsc mts lnrs mapType lnr count step scale rm_jack_interval auto_snap seed".split()])
    ids = torch.tensor([feature_dict_to_embedding_ids(fd, fy)] * count, dtype=torch.float32)
    uids = torch.tensor([feature_dict_to_embedding_ids({}, fy)] * count, dtype=torch.float32)
    mel = torch.from_numpy(load_audio_without_cache("song.wav", case["wave"]["n_freq"], hop, n_fft, sr, None).astype(np.float32))
    assert mel.shape[1] == z * ratio - 39
    m2 = job.fit_mel(mel, z, ratio)
    w = model.model.wave_model(torch.stack([m2] * count))
    x_T = torch.randn((count, 16, z), generator=torch.Generator().manual_seed(seed))
    lat, _ = DDIMSampler(model).sample(S=S, c=model.model.cond_stage_model(ids), w=w, batch_size=count, shape=None, verbose=False, x_T=x_T, eta=0.0,
                                       unconditional_guidance_scale=scale, unconditional_conditioning=model.model.cond_stage_model(uids))
    logits = model.model.decode(lat).cpu().numpy()
    frame_ms = hop / sr * nw_ratio * 1000

    def webui_order(hit_objects):                       # webui.py:400-406 with auto_snap on
        snapped, bpm, offset = gridify(hit_objects, verbose=False)
        return bpm, offset, remove_intractable_mania_mini_jacks(snapped, verbose=False, jack_interval=90)

    def write(rows, outdir):
        os.makedirs(outdir, exist_ok=True)
        _, meta = convertor.parse_osu_file("asset/template.osu", dict(frame_ms=frame_ms, max_frame=z * ratio // nw_ratio, from_logits=True))
        paths = []
        for i, row in enumerate(rows):
            paths.append(os.path.join(outdir, "c%d.osu" % i))
            convertor.save_osu_file(meta, row, path=paths[-1], override={"Version": "AI v%d" % (i + 1)}, gridify=webui_order)
        return paths
    assert [_sections(p) for p in write(logits, "api_out")] == [_sections(p) for p in charts]

    # ---- mug.job at scale = 5.0: `count` units of one audio in one launch, x_T rows drawn per unit seed -> compare unit by unit with the API
    rows = {}
    units = [job.Unit(audio=0, seed=seed + 10 * i, prompt=fd) for i in range(count)]
    monkeypatch.setattr(job, "z_length_for", lambda *a, **k: z)
    job.run_job(model, DDIMSampler(model), units, lambda a: mel, fy, steps=S, scale=scale, batch=count, max_audio_frame=z * ratio, z_length_cfg=z,
                on_chart=lambda i, u, row: rows.__setitem__(i, row.detach().cpu().numpy()))
    xj = torch.stack([torch.randn((16, z), generator=torch.Generator().manual_seed(u["seed"])) for u in units])
    latj, _ = DDIMSampler(model).sample(S=S, c=model.model.cond_stage_model(ids), w=w, batch_size=count, shape=None, verbose=False, x_T=xj, eta=0.0,
                                        unconditional_guidance_scale=scale, unconditional_conditioning=model.model.cond_stage_model(uids))
    want = model.model.decode(latj).cpu().numpy()
    assert sorted(rows) == list(range(count))
    assert [_sections(p) for p in write([rows[i] for i in range(count)], "job_out")] == [_sections(p) for p in write(want, "job_ref")]
