"""The headless sampling job (mug/job.py; reference: scripts/mapping.py:444-505, length rule webui.py:349-367):
length rule against hand-computed cases, song packing (two audios in one batch-2N launch) against separate launches, and
the 2-rank path -- real `gloo` process group, each rank running the tiny model on the emulated build -- against the
single-rank result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import cases, weights
from mug import job

G = cases.GOLDEN


def test_length_rule_hand_computed():
    """webui.py:349-356 with the shipped max_audio_frame = 32768, z_length = 512 (64 mel frames per latent step):
    z = (int(frames / 64 / 32) + 1) * 32 -- an exact multiple of 2048 frames still moves to the NEXT multiple of 32."""
    rule = job.z_length_for
    assert rule(1) == 32
    assert rule(2047) == 32 and rule(2048) == 64 and rule(2049) == 64          # 2048 / 64 / 32 = 1.0 exactly -> (1 + 1) * 32
    assert rule(31008) == 512                                                  # 180 s at 22.05 kHz, hop 128: 1 + 3969000 // 128 frames
    assert rule(32767) == 512 and rule(32768) == 544                           # the configured maximum itself spills over
    assert rule(104448 - 1) == 1632 and rule(104448) == 1664                   # 10-minute audio (configs[3])
    assert rule(5168) == 96                                                    # 30 s (configs[0]): 1 + 661500 // 128 frames -> 80.75 latent steps -> 96
    assert rule(6144) == 128                                                   # 6144 / 64 = 96 exactly = 3.0 * 32 -> (3 + 1) * 32
    # other configurations of the same rule
    assert job.z_length_for(1000, max_audio_frame=16384, z_length=256) == 32
    # pad / truncate to z * ratio frames (webui.py:358-367)
    mel = torch.arange(2 * 100, dtype=torch.float32).reshape(2, 100)
    padded = job.fit_mel(mel, 32, ratio=4)
    assert padded.shape == (2, 128) and torch.equal(padded[:, :100], mel) and float(padded[:, 100:].abs().sum()) == 0.0
    assert torch.equal(job.fit_mel(mel, 16, ratio=4), mel[:, :64])


def _model_config(case):
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


def _tiny_on_emu(z_max=64):
    """The tiny drop-in model with the emulated library installed as the process-wide one (test infrastructure)."""
    from conftest import emu_lib
    import mug._native as N
    from mug.diffusion.ddim import DDIMSampler
    from mug.util import instantiate_from_config
    N._default = emu_lib()
    case = cases.TINY
    model = instantiate_from_config(_model_config(case)).eval()
    man = weights.load_manifest(os.path.join(G, case["manifest"]))
    model.load_state_dict(weights.set_s4_lengths(weights.make_state_dict(man, seed=0), case["unet"], z_max))
    return case, model, DDIMSampler(model)


def _mel_of(case, lengths):
    ratio = case["audio_ratio"]

    def mel_of(a):
        frames = lengths[a]
        return cases.mel_input(case, 40 + a, 1, frames)[0]
    return mel_of, ratio


# audio -> mel frames: with the tiny case's ratio (max_audio_frame / z_length_cfg) these give z = 32, 32, 64
def _job_args(case):
    ratio = case["audio_ratio"]
    lengths = {0: 20 * ratio, 1: 31 * ratio, 2: 40 * ratio}
    mel_of, _ = _mel_of(case, lengths)
    import yaml
    with open(os.path.join(G, "mania_beatmap_features.yaml")) as f:
        fy = yaml.safe_load(f)
    kw = dict(steps=4, batch=2, max_audio_frame=32 * ratio, z_length_cfg=32)
    return mel_of, fy, kw


def _units():
    prompts = [{"sr": 4.0}, {"sr": 2.5, "ln_ratio": 0.4}, {}]
    return job.make_units(3, 2, prompts=prompts, seed0=7)


def test_song_packing_matches_separate_launches():
    case, model, sampler = _tiny_on_emu()
    mel_of, fy, kw = _job_args(case)
    units = _units()
    a, sa = job.run_job(model, sampler, units, mel_of, fy, pack_songs=1, **kw)
    b, sb = job.run_job(model, sampler, units, mel_of, fy, pack_songs=2, **kw)
    assert sa["launches"] == 3 and sb["launches"] == 2 and sa["audios"] == 3           # audios 0 and 1 (z = 32) share a launch; audio 2 (z = 64) cannot
    up = 2 ** (len(case["vae"]["channel_mult"]) - 1)
    assert [g.shape[-1] for g in a] == [32 * up] * 4 + [64 * up] * 2          # z = 32, 32, 64 by the length rule
    for ga, gb in zip(a, b):
        assert ga.dtype == torch.bool and torch.equal(ga, gb)
    assert sum(int(g.sum()) for g in a) > 0


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case, model, sampler = _tiny_on_emu()
        mel_of, fy, kw = _job_args(case)
        grids, stats = job.run_job(model, sampler, _units(), mel_of, fy, pack_songs=2, **kw)
        q.put((rank, [g.numpy() for g in grids], stats))
    finally:
        dist.destroy_process_group()


def test_two_rank_job_equals_single_rank():
    """configs[2] in miniature: 6 (audio, seed) units over 2 ranks (gloo), each rank running the networks itself; the
    gathered note grids on every rank equal the single-process result (units of different lengths included)."""
    case, model, sampler = _tiny_on_emu()
    mel_of, fy, kw = _job_args(case)
    want, _ = job.run_job(model, sampler, _units(), mel_of, fy, pack_songs=2, **kw)
    world = 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, grids, stats in got:
        assert stats["world"] == 2 and stats["units"] == 3
        assert len(grids) == len(want)
        for g, w in zip(grids, want):
            assert g.shape == tuple(w.shape) and (g == w.numpy()).all(), rank


def test_wav_fallback_decoder(tmp_path):
    """mug.util._read_wav: the stdlib RIFF reader used when neither librosa nor soundfile is installed (libsndfile scaling)."""
    import struct
    import wave
    from mug.util import _read_wav
    x = (np.sin(np.arange(999) / 7.0) * 20000).astype(np.int16)
    p = str(tmp_path / "s16.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(44100)
        w.writeframes(np.stack([x, -x], 1).tobytes())
    y, sr = _read_wav(p)
    assert sr == 44100 and y.shape == (999, 2) and y.dtype == np.float32
    assert np.array_equal(y[:, 0], x.astype(np.float32) / 32768.0) and np.array_equal(y[:, 1], -x.astype(np.float32) / 32768.0)
    # 24-bit and float32 payloads written by hand
    v24 = np.array([0, 1, -1, 8388607, -8388608], dtype=np.int64)
    raw = b"".join(struct.pack("<i", int(v))[:3] for v in v24)
    hdr = lambda fmt, bits, payload: (b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, fmt, 1, 22050, 22050 * bits // 8, bits // 8, bits)
                                      + b"data" + struct.pack("<I", len(payload)) + payload)
    p24 = tmp_path / "s24.wav"
    p24.write_bytes(hdr(1, 24, raw))
    y, sr = _read_wav(str(p24))
    assert sr == 22050 and np.allclose(y[:, 0], v24 / 8388608.0, atol=0)
    f = np.array([0.0, 0.5, -1.0, 0.25], dtype="<f4")
    pf = tmp_path / "f32.wav"
    pf.write_bytes(hdr(3, 32, f.tobytes()))
    y, _ = _read_wav(str(pf))
    assert np.array_equal(y[:, 0], f)


@pytest.mark.gpu
def test_cli_end_to_end_on_the_gpu(tmp_path):
    """`python -m mug.job` (the reference's scripts/mapping.py flow): tiny architecture YAML + seeded synthetic weights, one 44.1 kHz WAV,
    two samples -> two .osu files next to the template's metadata, through decode (stdlib WAV) -> device resampler -> log-mel -> wave
    encoder -> DDIM -> VAE decode -> mini-jack pass -> gridify -> save_osu_file."""
    import subprocess
    import sys
    import wave
    import yaml
    from oracle import host, postprocess as pp
    case = cases.TINY
    cfg = dict(model=_model_config(case), data=dict(params=dict(common_params=dict(sr=22050, n_fft=512, n_mels=case["wave"]["n_freq"],
               max_audio_frame=32 * case["audio_ratio"], audio_note_window_ratio=2 ** (len(case["vae"]["channel_mult"]) - 1)))))
    cfg["model"]["params"]["z_length"] = 32
    (tmp_path / "model.yaml").write_text(yaml.safe_dump(cfg))
    (tmp_path / "template.osu").write_text(pp.TEMPLATE_OSU, encoding="utf8")
    (tmp_path / "audio.mp3").write_bytes(b"")
    pcm = (host.synth_audio(0.9, sr=44100, seed=5) * 32767).astype(np.int16)
    with wave.open(str(tmp_path / "song.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(44100)
        w.writeframes(pcm.tobytes())
    out = tmp_path / "out"
    env = dict(os.environ, PYTHONPATH=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mug-diffusion_amd"))
    r = subprocess.run([sys.executable, "-m", "mug.job", "--config", str(tmp_path / "model.yaml"), "--synthetic-seed", "0",
                        "--audio", str(tmp_path / "song.wav"), "--feature_yaml", os.path.join(G, "mania_beatmap_features.yaml"),
                        "--template_beatmap", str(tmp_path / "template.osu"), "--outdir", str(out), "--n_samples", "2", "--ddim_steps", "4"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    files = sorted(os.listdir(out))
    assert len(files) == 2 and all(f.endswith(".osu") for f in files), (files, r.stdout[-1000:])
    text = (out / files[0]).read_text(encoding="utf8")
    assert "[HitObjects]" in text and "AI v1" in text
