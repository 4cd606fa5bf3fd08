"""Headless sampling job: what scripts/mapping.py:444-505 of the reference does for one audio and `n_samples`
seeds, generalised to a work list of independent (audio, seed, prompt) UNITS spread over the GPUs of a node
(BASELINE configs[2]).

    units  -> shard.partition by rank (units of one audio stay adjacent: its mel + wave-encode run once per rank)
           -> per audio: length rule (webui.py:349-356), pad / truncate the mel (:358-367), wave encoder (batch 1)
           -> per launch: up to `batch` seeds; with `pack_songs` = 2, two audios of the same latent length share ONE
              batch-2N launch (rows interleaved so that row b reads audio map b % 2: ConvSeg::bmod) -- the DDIM loop
              is launch-latency bound at batch 4, so a batch-8 launch costs ~1.4x a batch-4 one
           -> DDIM (one native call) -> VAE decode -> note grid (logit > 0)
           -> optional: gridify / mini-jack pass / .osu files (mug.data, the reference's own order: mapping.py:496-499)
           -> one all_gather of the bit-packed note grids at the end (the job's only collective)

No CPU fallback anywhere: the networks are the libmugd-backed mirror classes.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import shard


def z_length_for(frames, max_audio_frame=32768, z_length=512):
    """webui.py:349-356: ratio = max_audio_frame // z_length (64 mel frames per latent step);
    z = (int(frames / ratio / 32) + 1) * 32 -- the next multiple of 32 STRICTLY above frames / ratio."""
    ratio = max_audio_frame // z_length
    return (int(frames / ratio / 32) + 1) * 32


def fit_mel(mel, z, ratio=64):
    """webui.py:358-367: zero-pad or truncate the (n_mels, frames) log-mel to z * ratio frames."""
    t, tgt = mel.shape[-1], z * ratio
    if t < tgt:
        return torch.nn.functional.pad(mel, (0, tgt - t))
    return mel[..., :tgt]


def note_grid(logits):
    """convertor.py:212-216: channels 0..3 = note starts, 8..11 = holds; a cell is on where its logit is > 0."""
    return torch.cat([logits[:, 0:4] > 0, logits[:, 8:12] > 0], dim=1)


class Unit(dict):
    """One chart to generate: audio (key into the job's audio table), seed, prompt (feature dict)."""


def make_units(n_audio, seeds_per_audio, prompts=None, seed0=0):
    prompts = prompts or [{}]
    return [Unit(audio=a, seed=seed0 + a * seeds_per_audio + s, prompt=prompts[(a * seeds_per_audio + s) % len(prompts)])
            for a in range(n_audio) for s in range(seeds_per_audio)]


def _x_T(seed, z, channels=16):
    return torch.randn((channels, z), generator=torch.Generator(device="cpu").manual_seed(int(seed)))


def run_job(model, sampler, units, mel_of, feature_yaml, steps=50, scale=1.0, eta=0.0, batch=4, pack_songs=1,
            max_audio_frame=32768, z_length_cfg=512, group=None, gather=True, on_chart=None, tqdm_class=None):
    """Runs this rank's share of `units` and returns (grids, stats).

    model / sampler : the drop-in DDPM and its DDIMSampler (mug.diffusion)
    mel_of(audio)   : -> (n_mels, frames) fp32 device tensor, the log-mel of that audio (mug.util.pcm_to_log_mel / file loader)
    grids           : list over ALL units (gather=True, every rank) or this rank's units: bool (8, 8 z) tensors
    on_chart(unit_index, unit, logits_row) : optional sink for post-processing / file writing (runs on the owning rank)
    """
    from .util import feature_dict_to_embedding_ids
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    mine = list(shard.partition(len(units), world, rank))
    dev = next(model.parameters()).device
    ratio = max_audio_frame // z_length_cfg
    # ---- per audio: mel -> z -> wave-encoder maps (batch 1), once per rank
    audio_of = {}
    for u in mine:
        a = units[u]["audio"]
        if a not in audio_of:
            mel = mel_of(a)
            z = z_length_for(mel.shape[-1], max_audio_frame, z_length_cfg)
            w = model.model.wave_model(fit_mel(mel, z, ratio)[None].to(dev))
            audio_of[a] = dict(z=z, w=w, units=[])
        audio_of[a]["units"].append(u)
    # ---- launches: chunks of <= batch seeds per audio; pairs of equal-length audios packed into one launch
    chunks = []
    for a, rec in audio_of.items():
        for i in range(0, len(rec["units"]), batch):
            chunks.append((a, rec["units"][i:i + batch]))
    launches, used = [], [False] * len(chunks)
    for i, (a, us) in enumerate(chunks):
        if used[i]:
            continue
        used[i] = True
        group_ = [(a, us)]
        if pack_songs > 1:
            for j in range(i + 1, len(chunks)):
                b, vs = chunks[j]
                if len(group_) < pack_songs and not used[j] and b != a and len(vs) == len(us) and audio_of[b]["z"] == audio_of[a]["z"] \
                        and all(b != g[0] for g in group_):
                    used[j] = True
                    group_.append((b, vs))
        launches.append(group_)
    grids = {}
    n_launch = 0
    # ---- everything a launch needs that does not depend on the GPU: batch order, prompt ids, x_T (one CPU generator per unit:
    # reproducible per seed wherever the unit runs).  Staged for ALL launches in one go and copied with the stream still short, so
    # no host-side copy waits behind a DDIM loop later
    staged = []
    for group_ in launches:
        z = audio_of[group_[0][0]]["z"]
        ns, per = len(group_), len(group_[0][1])
        # batch row r = seed (r // ns) of song (r % ns): the U-Net reads audio map row b % ns
        order = [group_[r % ns][1][r // ns] for r in range(ns * per)]
        ids = torch.tensor([feature_dict_to_embedding_ids(units[u]["prompt"], feature_yaml) for u in order], dtype=torch.float32)
        uids = torch.tensor([feature_dict_to_embedding_ids({}, feature_yaml)] * len(order), dtype=torch.float32) if scale != 1.0 else None
        x_T = torch.stack([_x_T(units[u]["seed"], z) for u in order])
        staged.append((order, ids.to(dev, non_blocking=True), None if uids is None else uids.to(dev, non_blocking=True), x_T.to(dev, non_blocking=True)))
    for group_, (order, ids, uids, x_T) in zip(launches, staged):
        z = audio_of[group_[0][0]]["z"]
        w = [torch.cat([audio_of[a]["w"][l] for a, _ in group_], dim=0) for l in range(len(audio_of[group_[0][0]]["w"]))] if len(group_) > 1 \
            else audio_of[group_[0][0]]["w"]
        c = model.model.cond_stage_model(ids)
        uc = model.model.cond_stage_model(uids) if uids is not None else None
        model.z_length = z
        lat, _ = sampler.sample(S=steps, c=c, w=w, batch_size=len(order), eta=eta, verbose=False, x_T=x_T,
                                unconditional_guidance_scale=scale, unconditional_conditioning=uc,
                                tqdm_class=tqdm_class or (lambda it, **k: it))
        logits = model.model.decode(lat)
        g = note_grid(logits)
        for r, u in enumerate(order):
            grids[u] = g[r]
            if on_chart is not None:
                on_chart(u, units[u], logits[r])
        n_launch += 1
    stats = dict(rank=rank, world=world, units=len(mine), audios=len(audio_of), launches=n_launch)
    if not gather or world == 1:
        keys = mine if world > 1 else list(range(len(units)))
        if keys and len(set(tuple(grids[u].shape) for u in keys)) == 1:      # equal lengths: ONE device-to-host copy (and one sync)
            host = torch.stack([grids[u] for u in keys]).cpu()
            return [host[i] for i in range(len(keys))], stats
        return [grids[u].cpu() for u in keys], stats
    # ---- end of job: every rank gets every chart's grid.  Lengths differ per audio: gather them first (one int per unit),
    # pad the bit-packed rows to the longest, gather once
    T_local = torch.tensor([grids[u].shape[-1] for u in mine], dtype=torch.int64)
    per = -(-len(units) // world)
    tb = torch.zeros(per, dtype=torch.int64)
    tb[: len(mine)] = T_local
    backend_dev = dev if dist.get_backend(group) == "nccl" else torch.device("cpu")
    tl = [torch.empty_like(tb, device=backend_dev) for _ in range(world)]
    dist.all_gather(tl, tb.to(backend_dev), group=group)
    T_all = torch.cat([tl[r][: len(shard.partition(len(units), world, r))].cpu() for r in range(world)])
    Tmax = int(T_all.max().item()) if len(units) else 0
    local = torch.zeros((len(mine), 8, Tmax), dtype=torch.bool)
    for i, u in enumerate(mine):
        local[i, :, : grids[u].shape[-1]] = grids[u].cpu()
    full = shard.gather_grids(local, len(units), group=group, device=backend_dev if backend_dev.type == "cuda" else None)
    return [full[u, :, : int(T_all[u])] for u in range(len(units))], stats


def chart_writer(outdir, template_osu, frame_ms, max_frame, audio_path=None, creator="MuG Diffusion", jack_interval=90, lib=None):
    """on_chart sink: logits -> hit objects -> mini-jack pass -> gridify -> .osu next to a copy of the audio, in the
    reference CLI's order (scripts/mapping.py:487-505).  Returns (callback, written_paths)."""
    from .data import convertor, utils as du
    os.makedirs(outdir, exist_ok=True)
    written = []

    def custom_gridify(hit_objects):
        hit_objects = du.remove_intractable_mania_mini_jacks(hit_objects, verbose=False, jack_interval=jack_interval, lib=lib)
        hit_objects, bpm, offset = du.gridify(hit_objects, verbose=False, lib=lib)
        return bpm, offset, hit_objects

    def on_chart(index, unit, logits_row):
        params = dict(frame_ms=frame_ms, max_frame=max_frame, from_logits=True)
        _, meta = convertor.parse_osu_file(template_osu, params)
        path = os.path.join(outdir, "chart_%04d_seed%d.osu" % (index, unit["seed"]))
        convertor.save_osu_file(meta, logits_row.detach().cpu().numpy(), path=path,
                                override={"Version": "AI v%d" % (index + 1), "Creator": creator}, gridify=custom_gridify)
        written.append(path)

    return on_chart, written


def load_model(config_path, ckpt_path=None, device="cuda", seed_synthetic=None):
    """webui.py:42-58 / scripts/mapping.py:429-431: instantiate `model` from a YAML (configs/mug/mug_diffusion.yaml layout) and
    load the checkpoint's state dict (strict=False, like the reference).  ckpt_path=None + seed_synthetic: seeded synthetic
    weights of that architecture (benchmarks / smoke runs; no checkpoint ships offline)."""
    import yaml
    from .util import instantiate_from_config
    with open(config_path) as f:
        cfg = yaml.safe_load(f)
    model = instantiate_from_config(cfg["model"] if "model" in cfg else cfg).eval()
    if ckpt_path is not None:
        sd = torch.load(ckpt_path, map_location="cpu")
        sd = sd.get("state_dict", sd)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        if missing or unexpected:
            print("load_model: %d missing, %d unexpected keys" % (len(missing), len(unexpected)))
    elif seed_synthetic is not None:
        from .model.paramtree import seed_all_parameters
        unet = model.model.unet_model
        z_cfg = int(cfg.get("model", cfg)["params"].get("z_length", 512))
        seed_all_parameters(model, seed=seed_synthetic, s4_length_of=lambda k: unet.s4_length_of(k[len("model.unet_model."):], z_cfg))
    return model.to(device), cfg


def main(argv=None):
    """`python -m mug.job`: the reference's headless entry (scripts/mapping.py:308-522) on the drop-in, for one audio or a list.

        python -m mug.job --config models/ckpt/model.yaml --ckpt models/ckpt/model.ckpt --audio song.wav \\
               --template_beatmap data/template.osu --outdir outputs/beatmaps --n_samples 4 --ddim_steps 200 --scale 1.0

    Under `python -m torch.distributed.run --nproc-per-node N -m mug.job ... --audio a.wav b.wav ...` the (audio, sample) units are
    partitioned over the N GPUs; every rank writes the charts it generated."""
    import argparse
    import yaml
    from .diffusion.ddim import DDIMSampler
    from .util import load_audio_without_cache
    ap = argparse.ArgumentParser(prog="python -m mug.job", description=main.__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", required=True, help="model YAML (configs/mug/mug_diffusion.yaml / models/ckpt/model.yaml)")
    ap.add_argument("--ckpt", default=None, help="checkpoint with a state_dict in the reference's key layout; omit with --synthetic-seed")
    ap.add_argument("--synthetic-seed", type=int, default=None, help="seeded synthetic weights instead of a checkpoint")
    ap.add_argument("--audio", nargs="+", required=True)
    ap.add_argument("--feature_yaml", default="configs/mug/mania_beatmap_features.yaml")
    ap.add_argument("--prompt_dir", default=None, help="directory with feature_<i>.yaml per sample (scripts/mapping.py:422-425)")
    ap.add_argument("--template_beatmap", default=None, help=".osu template; without it only the note grids are produced")
    ap.add_argument("--outdir", default="outputs/beatmaps")
    ap.add_argument("--n_samples", type=int, default=4)
    ap.add_argument("--ddim_steps", type=int, default=200)
    ap.add_argument("--ddim_eta", type=float, default=0.0)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--pack_songs", type=int, default=1)
    a = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("mug.job needs an MI355X: libmugd has no CPU path")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    model, cfg = load_model(a.config, a.ckpt, device=torch.device("cuda", local), seed_synthetic=a.synthetic_seed)
    common = (cfg.get("data", {}).get("params", {}).get("common_params", {})) if isinstance(cfg, dict) else {}
    sr, n_fft, n_mels = int(common.get("sr", 22050)), int(common.get("n_fft", 512)), int(common.get("n_mels", 128))
    max_audio_frame = int(common.get("max_audio_frame", 32768))
    ratio_note = int(common.get("audio_note_window_ratio", 8))
    z_cfg = int(cfg.get("model", cfg)["params"].get("z_length", 512))
    with open(a.feature_yaml) as f:
        fy = yaml.safe_load(f)
    prompts = [{}]
    if a.prompt_dir:
        prompts = [yaml.safe_load(open(os.path.join(a.prompt_dir, "feature_%d.yaml" % (i + 1)))) for i in range(a.n_samples)]
    units = [Unit(audio=ai, seed=a.seed + ai * a.n_samples + s, prompt=prompts[s % len(prompts)])
             for ai in range(len(a.audio)) for s in range(a.n_samples)]
    hop = n_fft // 4

    def mel_of(ai):
        mel = load_audio_without_cache(a.audio[ai], n_mels, hop, n_fft, sr, None)       # fp16 (n_mels, frames), like the reference
        return torch.from_numpy(mel.astype(np.float32))

    on_chart, written = None, []
    if a.template_beatmap:
        frame_ms = hop / sr * ratio_note * 1000
        on_chart, written = chart_writer(a.outdir, a.template_beatmap, frame_ms, 1 << 30)
    grids, stats = run_job(model, DDIMSampler(model), units, mel_of, fy, steps=a.ddim_steps, scale=a.scale, eta=a.ddim_eta,
                           batch=a.n_samples, pack_songs=a.pack_songs, max_audio_frame=max_audio_frame, z_length_cfg=z_cfg,
                           on_chart=on_chart, gather=world > 1)
    print("rank %d: %d units, %d launches, %d charts written to %s" % (stats["rank"], stats["units"], stats["launches"], len(written), a.outdir))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return grids


if __name__ == "__main__":
    main()
