"""TEST INFRASTRUCTURE (see oracle/__init__.py): times the UNMODIFIED reference (imported from /root/reference through
oracle/refimport.py) on the stages of BASELINE configs[1] -- the honest CPU figure next to bench.py's `cpu_baseline` (which times
the oracle PORT on the GPU box, where /root/reference does not exist).

    python -m oracle.time_reference [threads]        ->  profiles/r2_reference_cpu_timing.json

Stages (same shapes as bench.py: 180 s audio -> z = 512, batch 4): log-mel (the librosa-algorithm restatement: librosa itself is
not installed), wave encoder B=1 @ 32768 frames, 3 U-Net evaluations B=4 (S4 kernels REGENERATED in every call, as the reference
does: s4.py:706-832), VAE decode B=4; extrapolated to 50 DDIM steps exactly like the bench's port figure.
"""
import json
import os
import sys
import time

from . import refimport

refimport.activate()

import numpy as np  # noqa: E402
import torch  # noqa: E402

from . import cases, host, weights  # noqa: E402
from .gen_golden import ref_model  # noqa: E402


def main():
    nthr = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    torch.set_num_threads(nthr)
    torch.set_grad_enabled(False)
    case, z, B, S = cases.FULL, 512, 4, 50
    model = ref_model(case)
    man = weights.load_manifest(os.path.join(cases.GOLDEN, case["manifest"]))
    model.load_state_dict(weights.set_s4_lengths(weights.make_state_dict(man, seed=0), case["unet"], z), strict=False)
    t0 = time.perf_counter()
    y = host.synth_audio(180.0 - 0.5)
    mel = host.pad_or_trunc_mel(host.log_mel(y).astype(np.float32), z * 64)
    t_mel = time.perf_counter() - t0
    mel_t = torch.from_numpy(mel)[None]
    t0 = time.perf_counter()
    w = model.model.wave_model(mel_t)
    t_wave_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    w = model.model.wave_model(mel_t)
    t_wave = time.perf_counter() - t0
    w = [m.repeat(B, 1, 1) for m in w]
    x, c = cases.x_T(1, B, z), cases.context(case, 1, B)
    model.model.unet_model(x, torch.full((B,), 981), c, *w)         # warm-up
    reps = 3
    t0 = time.perf_counter()
    for i in range(reps):
        model.model.unet_model(x, torch.full((B,), 981 - 20 * i), c, *w)
    t_unet = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    model.model.decode(x)
    t_dec = time.perf_counter() - t0
    per_batch = t_mel + t_wave + S * t_unet + t_dec
    out = {"what": "UNMODIFIED reference (Keytoyze/Mug-Diffusion, PyTorch-CPU fp32) in the authoring container, configs[1] stages",
           "threads": nthr, "host_cpus": os.cpu_count(), "torch": torch.__version__,
           "log_mel_s": t_mel, "wave_encoder_first_call_s": t_wave_first, "wave_encoder_s": t_wave, "unet_eval_B4_z512_s": t_unet,
           "vae_decode_B4_s": t_dec, "charts_per_s_extrapolated_50_steps": B / per_batch,
           "unet_sample_steps_per_s": B / t_unet,
           "note": "log-mel is the oracle's restatement of librosa's algorithm (librosa is not installed); everything else is the reference's own "
                   "modules with their own S4 kernel regeneration per call"}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r2_reference_cpu_timing.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
