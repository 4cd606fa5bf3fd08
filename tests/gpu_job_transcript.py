"""GPU transcript (development tool, run through gpurun): the stand-in for the reference's never-executed scripts/mapping.py --
`python -m mug.job` end to end on synthetic weights: a 30 s WAV in, .osu charts out.  Writes the model YAML, a WAV (stdlib `wave`),
the .osu template, runs the CLI as a subprocess and prints what it wrote (first lines of one chart)."""
import os
import subprocess
import sys
import wave

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "mug-diffusion_amd"))

import numpy as np  # noqa: E402
import yaml  # noqa: E402

import bench  # noqa: E402
from oracle import postprocess as pp  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/mug_job_demo"
    os.makedirs(out, exist_ok=True)
    cfg = {"model": bench.model_config(), "data": {"params": {"common_params": dict(sr=22050, n_fft=512, n_mels=128, max_audio_frame=32768,
                                                                                         audio_note_window_ratio=8)}}}
    with open(os.path.join(out, "model.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    pcm = bench.synth_audio(30.0, 44100, seed=7)
    with wave.open(os.path.join(out, "song.wav"), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(44100)
        w.writeframes((np.clip(pcm, -1, 1) * 32767).astype("<i2").tobytes())
    with open(os.path.join(out, "template.osu"), "w", encoding="utf8") as f:
        f.write(pp.TEMPLATE_OSU)
    cmd = [sys.executable, "-m", "mug.job", "--config", os.path.join(out, "model.yaml"), "--synthetic-seed", "0", "--audio", os.path.join(out, "song.wav"),
           "--feature_yaml", bench.FEATURE_YAML, "--template_beatmap", os.path.join(out, "template.osu"), "--outdir", os.path.join(out, "beatmaps"),
           "--n_samples", "4", "--ddim_steps", "50", "--scale", "1.0", "--seed", "11"]
    print("$ " + " ".join(cmd), flush=True)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "mug-diffusion_amd") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=out)
    print(r.stdout[-3000:])
    print("exit status", r.returncode)
    bm = os.path.join(out, "beatmaps")
    files = sorted(os.listdir(bm)) if os.path.isdir(bm) else []
    print("written:", files)
    for fn in files[:1]:
        with open(os.path.join(bm, fn), encoding="utf8") as f:
            lines = f.read().splitlines()
        print("---- %s (%d lines), head and first hit objects:" % (fn, len(lines)))
        i = lines.index("[HitObjects]") if "[HitObjects]" in lines else 0
        print("\n".join(lines[:12] + ["..."] + lines[i:i + 10]))
    for fn in files:
        with open(os.path.join(bm, fn), encoding="utf8") as f:
            lines = f.read().splitlines()
        i = lines.index("[HitObjects]") if "[HitObjects]" in lines else len(lines)
        print("%s: %d hit objects" % (fn, len(lines) - i - 1))


if __name__ == "__main__":
    main()
