"""TEST INFRASTRUCTURE (see oracle/__init__.py).

Generates tests/golden/* by running the UNMODIFIED reference (imported from
/root/reference through oracle/refimport.py) on factory weights + seeded
inputs, and at the same time checks the oracle restatement against it.

Run in the authoring container only:   python -m oracle.gen_golden
(/root/reference does not exist on the GPU box; the committed fixtures travel.)
"""
import json
import os
import sys
import time

import numpy as np

from . import refimport

refimport.activate()          # must precede any `import mug`

import torch  # noqa: E402
import yaml  # noqa: E402

from . import cases, host, nets, s4 as s4o, sampler, weights  # noqa: E402

torch.set_grad_enabled(False)
G = cases.GOLDEN
os.makedirs(G, exist_ok=True)
REPORT = {}


def note(key, val):
    REPORT[key] = float(val)
    print("  %-58s %.3e" % (key, val))


def maxdiff(a, b):
    return (a.double() - b.double()).abs().max().item()


def ref_model(case):
    """Builds the reference DDPM from a config dict shaped like configs/mug/mug_diffusion.yaml."""
    from mug.util import instantiate_from_config
    cfg = dict(target="mug.diffusion.diffusion.DDPM", params=dict(
        linear_start=0.0001, linear_end=0.02, log_every_t=100, timesteps=1000,
        z_channels=16, z_length=512, parameterization="eps", loss_type="smooth_l1",
        monitor="val/loss_simple",
        unet_config=dict(target="mug.diffusion.unet.UNetModel",
                         params=dict(dropout=0.0, lstm_last=False, lstm_layer=False, use_checkpoint=False,
                                     **case["unet"])),
        first_stage_config=dict(target="mug.firststage.autoencoder.AutoencoderKL",
                                params=dict(monitor="val/loss", kl_weight=1e-6, ddconfig=dict(case["vae"]),
                                            lossconfig=dict(target="torch.nn.Identity"))),
        cond_stage_config=dict(target="mug.cond.feature.BeatmapFeatureEmbedder",
                               params=dict(path_to_yaml="configs/mug/mania_beatmap_features.yaml",
                                           embed_dim=case["unet"]["context_dim"])),
        wave_stage_config=dict(target="mug.cond.wave.MelspectrogramScaleEncoder1D",
                               params=dict(dropout=0.0, use_checkpoint=True, **case["wave"]))))
    cwd = os.getcwd()
    os.chdir(refimport.REF)
    try:
        torch.manual_seed(0)
        m = instantiate_from_config(cfg)
    finally:
        os.chdir(cwd)
    return m.eval()


def load(model, sd):
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing[:5], unexpected[:5])


def run_case(case, z_list, ddim_cfg):
    tag = case["name"]
    print("== case %s" % tag)
    model = ref_model(case)
    man = weights.manifest_of(model.state_dict())
    with open(os.path.join(G, case["manifest"]), "w") as f:
        json.dump(man, f, separators=(",", ":"))
    print("  manifest entries:", len(man))

    from mug.diffusion.ddim import DDIMSampler
    for z in z_list:
        sd = weights.set_s4_lengths(weights.make_state_dict(man, seed=0), case["unet"], z)
        load(model, sd)
        # schedule buffers of the factory must equal the reference's own
        for k, v in sampler.register_schedule().items():
            assert np.array_equal(v, model.state_dict()[k].numpy()), k
        for B in ([1, 2] if z <= 128 else [1]):
            seed = 100 + z + B
            x = cases.x_T(seed, B, z)
            t = torch.tensor([981, 21][:B], dtype=torch.long)
            c = cases.context(case, seed, B)
            w = cases.audio_maps(case, seed, B, z)
            t0 = time.time()
            e_ref = model.model.unet_model(x, t, c, *w)
            t1 = time.time()
            e_orc = nets.unet_forward(sd, case["unet"], x, t, c, w)
            t2 = time.time()
            note("%s unet z=%d B=%d |oracle-ref|max (ref %.2fs orc %.2fs)" % (tag, z, B, t1 - t0, t2 - t1),
                 maxdiff(e_ref, e_orc))
            note("%s unet z=%d B=%d |ref|max" % (tag, z, B), e_ref.abs().max().item())
            np.savez_compressed(os.path.join(G, "%s_unet_z%d_b%d.npz" % (tag, z, B)),
                                seed=seed, t=t.numpy(), eps=e_ref.numpy())

        # ---- S4 kernels (two extreme layers), reference vs oracle(reference mode) vs exact
        ks = {}
        for name, mod in model.model.unet_model.named_modules():
            if name.endswith("s4_model.kernel.kernel"):
                L = int(mod.L)
                k_ref = mod(L=L)[0][0]
                p = "model.unet_model." + name
                k_orc = s4o.s4_kernel(sd, p, L, "reference")
                k_ex = s4o.s4_kernel(sd, p, L, "exact")
                ks[name] = (k_ref, k_orc, k_ex)
        d1 = max(maxdiff(a, b) for a, b, _ in ks.values())
        d2 = max(maxdiff(a, c_) for a, _, c_ in ks.values())
        km = max(a.abs().max().item() for a, _, _ in ks.values())
        note("%s z=%d s4 kernel |oracle-ref|max over %d layers" % (tag, z, len(ks)), d1)
        note("%s z=%d s4 kernel |exact-ref|max" % (tag, z), d2)
        note("%s z=%d s4 kernel |ref|max" % (tag, z), km)
        first, last = list(ks)[0], list(ks)[-1]
        np.savez_compressed(os.path.join(G, "%s_s4kernel_z%d.npz" % (tag, z)),
                            names=np.array([first, last]), k0=ks[first][0].numpy(), k1=ks[last][0].numpy(),
                            k0_exact=ks[first][2].numpy(), k1_exact=ks[last][2].numpy())

        # ---- VAE decode
        zlat = cases.randn(7 + z, 2, (2, 16, z))
        d_ref = model.model.decode(zlat)
        d_orc = nets.vae_decode(sd, case["vae"], zlat)
        note("%s vae z=%d |oracle-ref|max" % (tag, z), maxdiff(d_ref, d_orc))
        np.savez_compressed(os.path.join(G, "%s_vae_z%d.npz" % (tag, z)), seed=7 + z, logits=d_ref.numpy())

        # ---- wave encoder
        frames = z * case["audio_ratio"]
        mel = cases.mel_input(case, 11 + z, 1, frames)
        w_ref = model.model.wave_model(mel)
        w_orc = nets.wave_encode(sd, case["wave"], mel)
        note("%s wave frames=%d |oracle-ref|max" % (tag, frames), max(maxdiff(a, b) for a, b in zip(w_ref, w_orc)))
        nl = len(case["unet"]["channel_mult"])
        np.savez_compressed(os.path.join(G, "%s_wave_z%d.npz" % (tag, z)), seed=11 + z,
                            **{"w%d" % i: a.numpy() for i, a in enumerate(w_ref[-nl:])},
                            absmean=np.array([a.abs().mean().item() for a in w_ref]))

        # ---- end-to-end DDIM (+ decode), explicit x_T, eta=0
        for (S, B, scale) in ddim_cfg.get(z, []):
            seed = 500 + z + S
            fy = cases.feature_yaml()
            prompts = [{"sr": 4.0, "rank_status": "ranked"}, {"sr": 2.5, "ln_ratio": 0.4}][:B]
            ids = torch.tensor([host.feature_ids(pm, fy) for pm in prompts], dtype=torch.float32)
            c = model.model.cond_stage_model(ids)
            assert maxdiff(c, nets.cond_embed(sd, ids)) == 0.0
            uc = model.model.cond_stage_model(torch.tensor([host.feature_ids({}, fy)] * B, dtype=torch.float32))
            wl = [wi.repeat(B, 1, 1) for wi in w_ref]
            xT = cases.x_T(seed, B, z)
            model.z_length = z
            smp = DDIMSampler(model)
            t0 = time.time()
            lat_ref, _ = smp.sample(S=S, c=c, w=wl, batch_size=B, eta=0.0, verbose=False, x_T=xT,
                                    unconditional_guidance_scale=scale, unconditional_conditioning=uc,
                                    tqdm_class=lambda it, **k: it)
            t1 = time.time()
            lat_orc = sampler.ddim_sample(sd, case["unet"], S, c, wl, xT, 0.0, scale, uc)
            dec_ref = model.model.decode(lat_ref)
            dec_orc = nets.vae_decode(sd, case["vae"], lat_orc)
            key = "%s ddim z=%d S=%d B=%d cfg=%g" % (tag, z, S, B, scale)
            note(key + " latent |oracle-ref|max (ref %.1fs)" % (t1 - t0), maxdiff(lat_ref, lat_orc))
            note(key + " latent |ref|max", lat_ref.abs().max().item())
            gs_r, gh_r = host.note_grid(dec_ref.numpy())
            gs_o, gh_o = host.note_grid(dec_orc.numpy())
            flips = int((gs_r != gs_o).sum() + (gh_r != gh_o).sum())
            note(key + " grid flips oracle vs ref", flips)
            note(key + " notes on (start)", int(gs_r.sum()))
            np.savez_compressed(os.path.join(G, "%s_ddim_z%d_S%d_B%d_cfg%g.npz" % (tag, z, S, B, scale)),
                                seed=seed, ids=ids.numpy(), latent=lat_ref.numpy(), logits=dec_ref.numpy(),
                                mel_seed=11 + z)
    del model


def host_goldens():
    """Schedule arrays and prompt ids straight from the reference's own host functions."""
    from mug.diffusion.utils import make_ddim_timesteps, make_ddim_sampling_parameters, make_beta_schedule
    from mug.util import feature_dict_to_embedding_ids, count_beatmap_features
    with open(os.path.join(refimport.REF, "configs/mug/mania_beatmap_features.yaml")) as f:
        fy = yaml.safe_load(f)
    with open(os.path.join(G, "mania_beatmap_features.yaml"), "w") as f:
        yaml.safe_dump(fy, f, sort_keys=False)
    betas = make_beta_schedule("linear", 1000, 1e-4, 2e-2)
    ac = torch.tensor(np.cumprod(1.0 - betas), dtype=torch.float32)
    out = {}
    for S in (10, 50, 100, 200, 7):
        ts = make_ddim_timesteps("uniform", S, 1000, verbose=False)
        for eta in (0.0, 1.0):
            sig, a, ap = make_ddim_sampling_parameters(ac.cpu(), ts, eta, verbose=False)
            o_sig, o_a, o_ap, _ = sampler.ddim_parameters(ac.numpy(), sampler.ddim_timesteps(S), eta)
            assert np.array_equal(ts, sampler.ddim_timesteps(S))
            assert np.array_equal(np.asarray(a), o_a) and np.array_equal(np.asarray(ap), o_ap)
            # eta>0: the reference mixes torch-f32 / numpy-f64 operands; agreement is to f32 rounding, exact for eta=0
            assert np.allclose(np.asarray(sig), o_sig, rtol=0, atol=(0 if eta == 0 else 1e-7)), (S, eta)
            out["S%d_eta%g" % (S, eta)] = dict(ts=ts.tolist(), sigmas=np.asarray(sig, dtype=np.float64).tolist(),
                                               alphas=np.asarray(a, dtype=np.float64).tolist(),
                                               alphas_prev=np.asarray(ap, dtype=np.float64).tolist())
    prompts = [{}, {"sr": 4.0, "rank_status": "ranked"}, {"sr": 2.5, "ln_ratio": 0.4},
               {"sr": 100.0, "ln": True if any(x["name"] == "ln" for x in fy) else None}]
    for x in fy:   # one prompt exercising every feature at its mid/first value
        pass
    full = {}
    for x in fy:
        if x["type"] == "numeric":
            full[x["name"]] = (x["min"] + x["max"]) / 2
        elif x["type"] == "bool":
            full[x["name"]] = 1
        else:
            full[x["name"]] = x["category"][-1]
    prompts.append(full)
    idl = []
    for pm in prompts:
        pm = {k: v for k, v in pm.items() if v is not None}
        r = feature_dict_to_embedding_ids(pm, fy)
        assert r == host.feature_ids(pm, fy), pm
        idl.append(dict(prompt=pm, ids=r))
    assert count_beatmap_features(fy) == host.feature_table_rows(fy)
    with open(os.path.join(G, "host_golden.json"), "w") as f:
        json.dump(dict(ddim=out, prompts=idl, table_rows=count_beatmap_features(fy)), f)
    print("  host goldens: schedules + %d prompts OK" % len(idl))


def s4_host_goldens():
    """SSKernelNPLR fresh-init values and two lazy `_setup_C` growth steps (s4.py:557-584, 726-730)
    straight from the reference module: the pin for mug/model/s4.py (product host code)."""
    from mug.model.s4 import S4
    torch.manual_seed(3)
    m = S4(d_model=8, d_state=64)
    k = m.kernel.kernel
    names = ["C", "B", "P", "inv_w_real", "w_imag", "log_dt"]
    out = {"init_" + n: getattr(k, n).detach().clone().numpy() for n in names}
    out["init_L"] = np.array(int(k.L))
    k1 = k(L=24)[0].detach().clone()
    out["C_after24"] = k.C.detach().clone().numpy()
    out["L_after24"] = np.array(int(k.L))
    out["k_after24"] = k1.numpy()
    k2 = k(L=40)[0].detach().clone()
    out["C_after40"] = k.C.detach().clone().numpy()
    out["L_after40"] = np.array(int(k.L))
    out["k_after40"] = k2.numpy()
    np.savez_compressed(os.path.join(G, "s4_setup_C.npz"), **out)
    print("  s4 host goldens: L %d -> %d -> %d" % (out["init_L"], out["L_after24"], out["L_after40"]))


def encode_goldens():
    """AutoencoderKL.encode of the real reference (inpainting / partial regeneration row) -> *_vaeenc_*.npz."""
    rep = {}
    for case, z_list in ((cases.TINY, [32]), (cases.FULL, [96])):
        model = ref_model(case)
        man = weights.manifest_of(model.state_dict())
        for z in z_list:
            sd = weights.set_s4_lengths(weights.make_state_dict(man, seed=0), case["unet"], z)
            load(model, sd)
            up = 2 ** (len(case["vae"]["channel_mult"]) - 1)
            x = cases.randn(13 + z, 3, (2, case["vae"]["x_channels"], z * up))
            post = model.model.first_stage_model.encode(x)
            m_ref = post.parameters
            m_orc = nets.vae_encode(sd, case["vae"], x)
            d = maxdiff(m_ref, m_orc)
            print("  %s vae encode z=%d |oracle-ref|max %.3e  |ref|max %.3f" % (case["name"], z, d, m_ref.abs().max().item()))
            assert d == 0.0
            rep["%s vae encode z=%d |oracle-ref|max" % (case["name"], z)] = d
            np.savez_compressed(os.path.join(G, "%s_vaeenc_z%d.npz" % (case["name"], z)), seed=13 + z, moments=m_ref.numpy(),
                                mode=post.mode().numpy(), logvar=post.logvar.numpy())
    path = os.path.join(G, "oracle_vs_reference.json")
    with open(path) as f:
        full = json.load(f)
    full.update(rep)
    with open(path, "w") as f:
        json.dump(full, f, indent=1)


def postprocess_goldens():
    """`gridify` and `remove_intractable_mania_mini_jacks` of the real reference (mug/data/utils.py) on
    seeded synthetic charts -> tests/golden/postprocess_golden.json.gz.  The module is loaded by file path: it needs numpy and
    scikit-learn only.  Floats are stored as hex so the comparison is bit-for-bit."""
    import contextlib
    import importlib.util
    import io
    import sklearn
    from oracle import postprocess as pp
    spec = importlib.util.spec_from_file_location("ref_data_utils", os.path.join(refimport.REF, "mug/data/utils.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    specs = [dict(seed=0, beats=40), dict(seed=1, beats=160, bpm=187.0, offset=412.0),
             dict(seed=2, beats=120, bpm=222.22, offset=35.0, jitter=7.0, ln_p=0.4),
             dict(seed=3, beats=200, bpm=150.0, offset=1999.0, jitter=2.0, density=0.85, ln_p=0.05),
             dict(seed=4, beats=64, bpm=271.3, offset=0.0, jitter=12.0),
             dict(seed=5, beats=700, bpm=174.0, offset=733.0, jitter=3.0)]
    charts = [dict(spec=sp, objects=pp.synthetic_chart(**sp)) for sp in specs]
    charts.append(dict(spec="one note", objects=["64,192,1000,1,0,0:0:0:0:"]))
    charts.append(dict(spec="two notes", objects=["64,192,1000,1,0,0:0:0:0:", "192,192,1400,128,0,1900:0:0:0:0:"]))
    charts.append(dict(spec="same-time chord", objects=["%d,192,500,1,0,0:0:0:0:" % x for x in (64, 192, 320, 448)]))
    out = []
    for ch in charts:
        ho = ch["objects"]
        log = io.StringIO()
        with contextlib.redirect_stdout(log):
            snapped, bpm, offset = ref.gridify(ho, verbose=True)
        improvements = log.getvalue().count("[valid")
        rec = dict(spec=ch["spec"], objects=ho, snapped=snapped, bpm=float(bpm).hex(), offset=float(offset).hex(),
                   offset_dtype=type(offset).__name__, improvements=improvements, jacks={})
        # webui.py:401-407 order (snap, then jacks at the UI's interval) and scripts/mapping.py:496-498 order (jacks first)
        for name, src, interval in (("after_snap_90", snapped, 90), ("raw_90", ho, 90), ("raw_150", ho, 150), ("after_snap_40", snapped, 40)):
            rec["jacks"][name] = ref.remove_intractable_mania_mini_jacks(src, verbose=False, jack_interval=interval)
            assert pp.remove_mini_jacks(src, interval) == rec["jacks"][name], (ch["spec"], name)
        o_snapped, o_bpm, o_offset = pp.gridify(ho)
        assert o_snapped == snapped and float(o_bpm).hex() == rec["bpm"] and float(o_offset).hex() == rec["offset"], ch["spec"]
        assert type(o_offset).__name__ == rec["offset_dtype"]
        print("  postprocess %-40s %5d notes  bpm %.4f offset %.3f  %d refits  jacks -> %d notes"
              % (str(ch["spec"])[:40], len(ho), bpm, offset, improvements, len(rec["jacks"]["after_snap_90"])))
        out.append(rec)
    # whole files: note grid -> save_osu_file with the web UI's post-processing (webui.py:401-407,431-445), real reference
    spec = importlib.util.spec_from_file_location("ref_convertor_golden", os.path.join(refimport.REF, "mug/data/convertor.py"))
    conv = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = conv
    spec.loader.exec_module(conv)
    import tempfile
    files = []
    with tempfile.TemporaryDirectory() as tmp:
        osu = os.path.join(tmp, "template.osu")
        with open(osu, "w", encoding="utf8") as f:
            f.write(pp.TEMPLATE_OSU)
        open(os.path.join(tmp, "audio.mp3"), "wb").close()
        params = dict(frame_ms=128 / 22050 * 8 * 1000, max_frame=4096, from_logits=True)
        _, meta = conv.parse_osu_file(osu, dict(params))

        def ui_gridify(objs):
            snapped, bpm, offset = ref.gridify(objs, verbose=False)
            return bpm, offset, ref.remove_intractable_mania_mini_jacks(snapped, verbose=False, jack_interval=90)

        for seed, frames in ((21, 512), (22, 4096)):
            grid = pp.synthetic_note_grid(seed, frames)
            path = os.path.join(tmp, "out%d.osu" % seed)
            conv.save_osu_file(meta, grid, path=path, override={"Version": "AI v%d" % seed, "Creator": "golden"}, gridify=ui_gridify)
            with open(path, encoding="utf8") as f:
                text = f.read()
            files.append(dict(seed=seed, frames=frames, text=text))
            print("  chart file seed %d: %d frames -> %d lines" % (seed, frames, text.count("\n")))
    import gzip
    with gzip.open(os.path.join(G, "postprocess_golden.json.gz"), "wt", encoding="utf8") as f:
        json.dump(dict(numpy=np.__version__, sklearn=sklearn.__version__, charts=out, files=files), f)
    print("  postprocess goldens: %d charts, oracle restatement identical" % len(out))


def masked_goldens():
    """Inpainting / partial regeneration: DDIMSampler.sample(mask=, x0=) of the REAL reference
    (ddim.py:141-144: x = q_sample(x0, t) * mask + (1 - mask) * x before every step), seeded CPU generator.  The RNG
    consumption per step is q_sample's randn_like(x0) followed by p_sample_ddim's randn(shape) (drawn although eta = 0)."""
    from mug.diffusion.ddim import DDIMSampler
    for case, z, S, B in ((cases.TINY, 32, 4, 2), (cases.FULL, 96, 10, 1)):
        tag = case["name"]
        model = ref_model(case)
        man = weights.load_manifest(os.path.join(G, case["manifest"]))
        sd = weights.set_s4_lengths(weights.make_state_dict(man, seed=0), case["unet"], z)
        model.load_state_dict(sd)
        fy = cases.feature_yaml()
        prompts = [{"sr": 4.0, "rank_status": "ranked"}, {"sr": 2.5, "ln_ratio": 0.4}][:B]
        ids = torch.tensor([host.feature_ids(pm, fy) for pm in prompts], dtype=torch.float32)
        c = model.model.cond_stage_model(ids)
        mel = cases.mel_input(case, 11 + z, 1, z * case["audio_ratio"])
        wl = [wi.repeat(B, 1, 1) for wi in model.model.wave_model(mel)]
        seed = 900 + z
        xT = cases.x_T(seed, B, z)
        x0 = cases.randn(seed, 5, (B, 16, z))
        mask = torch.zeros(B, 16, z)
        mask[:, :, : z // 2] = 1.0                                   # keep the first half of the chart, regenerate the rest
        model.z_length = z
        torch.manual_seed(seed)
        lat, _ = DDIMSampler(model).sample(S=S, c=c, w=wl, batch_size=B, eta=0.0, verbose=False, x_T=xT, mask=mask, x0=x0,
                                           tqdm_class=lambda it, **k: it)
        dec = model.model.decode(lat)
        note("%s masked ddim z=%d S=%d latent |ref|max" % (tag, z, S), lat.abs().max().item())
        np.savez_compressed(os.path.join(G, "%s_ddim_masked_z%d_S%d_B%d.npz" % (tag, z, S, B)), seed=seed, ids=ids.numpy(),
                            latent=lat.numpy(), logits=dec.numpy(), mel_seed=11 + z)
        del model


# tensors whose full gradients the shipped-size fixture keeps (every block type, all three trainable networks; the rest are
# pinned through their L2 norms)
TRAIN_FULL_KEEP = [
    "model.cond_stage_model.embedding.weight",
    "model.unet_model.time_embed.0.weight", "model.unet_model.time_embed.2.bias",
    "model.unet_model.input_blocks.0.0.weight",
    "model.unet_model.input_blocks.2.0.in_layers.2.weight", "model.unet_model.input_blocks.2.0.emb_layers.1.weight",
    "model.unet_model.input_blocks.2.0.skip_connection.weight", "model.unet_model.input_blocks.2.0.out_layers.0.weight",
    "model.unet_model.input_blocks.2.1.s4_model.kernel.kernel.C", "model.unet_model.input_blocks.2.1.s4_model.kernel.kernel.log_dt",
    "model.unet_model.input_blocks.2.1.s4_model.kernel.kernel.inv_w_real", "model.unet_model.input_blocks.2.1.s4_model.D",
    "model.unet_model.input_blocks.4.0.conv.weight",
    "model.unet_model.input_blocks.6.1.transformer_blocks.0.attn1.to_q.weight",
    "model.unet_model.input_blocks.6.1.transformer_blocks.0.attn1.relative_position_embedding",
    "model.unet_model.input_blocks.6.1.transformer_blocks.0.attn2.to_k.weight",
    "model.unet_model.input_blocks.6.1.transformer_blocks.0.attn2.C_embedding",
    "model.unet_model.input_blocks.6.1.transformer_blocks.0.ff.net.0.proj.weight",
    "model.unet_model.input_blocks.6.1.transformer_blocks.0.norm2.weight",
    "model.unet_model.middle_block.1.proj_out.weight",
    "model.unet_model.output_blocks.3.2.conv.weight", "model.unet_model.output_blocks.15.0.out_layers.3.weight",
    "model.unet_model.out.0.weight", "model.unet_model.out.2.weight",
    "model.wave_model.conv_in.weight", "model.wave_model.down.6.block.0.conv1.weight", "model.wave_model.down.6.block.1.conv2.weight",
    "model.wave_model.down.7.block.0.nin_shortcut.weight", "model.wave_model.down.7.downsample.conv.weight",
    "model.wave_model.down.8.attn.0.transformer_blocks.0.attn2.to_v.weight", "model.wave_model.down.9.block.1.norm2.bias",
]


def train_goldens():
    """The training step of the REAL reference (BASELINE configs[4]): `DDPM.forward` (diffusion.py:408-414:
    frozen `first_stage_model.encode(batch['note'])` -> `.mode()`, t ~ randint(generator), then `p_losses` :356-406: noise ~
    randn(generator), q_sample, cond / wave encoders with grad, U-Net, smooth_l1(beta 0.02) + 0.01, mean over (C, T) then batch)
    followed by `loss.backward()`, in train mode, on the tiny and the shipped architecture.  Stored: the batch, the (t, noise)
    the generator produced, the loss, x_start, the L2 norm of EVERY parameter gradient and the full gradients of a spread of
    tensors (all of them for the tiny model).  The oracle's autograd restatement of the same step is checked against it here."""
    import torch.nn.functional as F
    rep = {}
    for case, z, B, keep in ((cases.TINY, 32, 2, None), (cases.FULL, 96, 2, TRAIN_FULL_KEEP)):
        tag = case["name"]
        model = ref_model(case)
        man = weights.manifest_of(model.state_dict())
        sd = weights.set_s4_lengths(weights.make_state_dict(man, seed=0), case["unet"], z)
        load(model, sd)
        model.train()                                             # the first stage stays in eval mode (disabled_train, diffusion.py:15-18,36-41)
        seed = 1300 + z
        note_t, mel, ids = cases.train_batch(case, seed, B, z, sd["model.cond_stage_model.embedding.weight"].shape[0])
        batch = {"note": note_t, "audio": mel, "feature": ids.to(torch.float32)}
        g = torch.Generator().manual_seed(seed)
        t0 = time.time()
        with torch.enable_grad():
            for p_ in model.parameters():
                p_.grad = None
            loss, loss_dict = model(batch, generator=g)
            loss.backward()
        t1 = time.time()
        # what the generator handed out inside forward / p_losses, replayed: randint for t, then randn for the noise
        g2 = torch.Generator().manual_seed(seed)
        t = torch.randint(0, model.num_timesteps, (B,), generator=g2).long()
        x_start = model.model.encode(batch).mode()
        noise = torch.randn(x_start.size(), generator=g2)
        grads = {k: p_.grad.detach().clone() for k, p_ in model.named_parameters() if p_.grad is not None}
        frozen = [k for k, p_ in model.named_parameters() if p_.grad is None]
        assert all(k.startswith("model.first_stage_model.") or k.startswith("model.wave_model.") for k in frozen), frozen[:5]
        print("  %s train z=%d B=%d: loss %.6f, %d gradient tensors, %d without gradient (frozen VAE / wave levels the U-Net does not read), %.1fs"
              % (tag, z, B, float(loss), len(grads), len(frozen), t1 - t0))
        # ---- the oracle's restatement of the same step through autograd
        trainable = [k for k in grads]
        st = {k: (v.clone().requires_grad_(True) if k in trainable else v) for k, v in sd.items()}
        with torch.enable_grad():
            x0_o = nets.vae_encode(sd, case["vae"], note_t)[:, :case["z_channels"]]         # mode() of the diagonal Gaussian = its mean
            assert maxdiff(x0_o, x_start) == 0.0
            xt = st["sqrt_alphas_cumprod"][t][:, None, None] * x0_o + st["sqrt_one_minus_alphas_cumprod"][t][:, None, None] * noise
            ctx = nets.cond_embed(st, ids)
            w = nets.wave_encode(st, case["wave"], mel)
            pred = nets.unet_forward(st, case["unet"], xt, t, ctx, w)
            lo = (F.smooth_l1_loss(noise, pred, beta=0.02, reduction="none") + 0.01).mean(dim=[1, 2]).mean()
            lo.backward()
        dl = abs(float(lo) - float(loss))
        worst = max(((st[k].grad - grads[k]).abs().max().item() / max(grads[k].abs().max().item(), 1e-12), k) for k in trainable)
        note("%s train z=%d loss |oracle-ref|" % (tag, z), dl)
        note("%s train z=%d worst relative gradient |oracle-ref| (%s)" % (tag, z, worst[1][-40:]), worst[0])
        rep["%s train z=%d loss |oracle-ref|" % (tag, z)] = dl
        rep["%s train z=%d worst relative gradient |oracle-ref|" % (tag, z)] = worst[0]
        names = sorted(grads)
        full = names if keep is None else keep
        for k in full:
            assert k in grads, k
        # tensors above 8192 elements are kept as a seeded sample of 8192 flat positions (cases.rng(seed, 9, position in `full_names`))
        kept = {}
        for i, k in enumerate(full):
            gk = grads[k].reshape(-1)
            kept["g%d" % i] = (gk if gk.numel() <= 8192 else gk[torch.from_numpy(cases.train_sample_index(seed, i, gk.numel()))]).numpy()
        # the batch itself is reproducible from the seed (cases.randn / mel_input / rng: numpy generators); t and the noise came out
        # of a torch generator and are stored
        np.savez_compressed(os.path.join(G, "%s_train_z%d_B%d.npz" % (tag, z, B)), seed=seed, t=t.numpy(), noise=noise.numpy(),
                            x_start=x_start.numpy(), loss=np.float64(float(loss)), loss_simple=np.float64(float(loss_dict["train/loss_simple"])),
                            names=np.array(names), norms=np.array([float(grads[k].double().norm()) for k in names]),
                            absmax=np.array([float(grads[k].abs().max()) for k in names]), full_names=np.array(full), **kept)
        del model
    path = os.path.join(G, "oracle_vs_reference.json")
    with open(path) as f:
        full_rep = json.load(f)
    full_rep.update(rep)
    with open(path, "w") as f:
        json.dump(full_rep, f, indent=1)


def main():
    torch.set_num_threads(os.cpu_count())
    if "--train-only" in sys.argv:
        train_goldens()
        return
    if "--masked-only" in sys.argv:
        masked_goldens()
        return
    if "--postprocess-only" in sys.argv:
        postprocess_goldens()
        return
    if "--encode-only" in sys.argv:
        encode_goldens()
        return
    if "--s4-only" in sys.argv:
        s4_host_goldens()
        return
    host_goldens()
    s4_host_goldens()
    postprocess_goldens()
    masked_goldens()
    train_goldens()
    run_case(cases.TINY, [32], {32: [(4, 2, 1.0), (4, 2, 5.0)]})
    if "--tiny-only" not in sys.argv:
        run_case(cases.FULL, [96, 512], {96: [(10, 1, 1.0), (10, 1, 5.0)]})
    with open(os.path.join(G, "oracle_vs_reference.json"), "w") as f:
        json.dump(REPORT, f, indent=1)


if __name__ == "__main__":
    main()
