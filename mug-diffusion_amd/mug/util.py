"""mug/util.py of the reference, re-implemented: the plugin loader (the drop-in boundary), prompt ids,
and the audio front-end whose STFT->mel arithmetic runs in libmugd (k_mel.hip)."""
import importlib
import math
from inspect import isfunction

import numpy as np
import torch


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if isfunction(d) else d


def mean_flat(tensor):
    return tensor.mean(dim=list(range(1, len(tensor.shape))))


def count_params(model, verbose=False):
    n = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{model.__class__.__name__} has {n * 1.e-6:.2f} M params.")
    return n


def count_beatmap_features_embedding(x):
    """mug/util.py:51-60."""
    if x['type'] == 'numeric':
        return int(math.ceil((x['max'] - x['min']) / x['interval'])) + 1
    if x['type'] == 'category':
        return len(x['category']) + 1
    if x['type'] == 'bool':
        return 3
    raise ValueError(str(x))


def feature_dict_to_embedding_ids(feature_dict, feature_yaml):
    """mug/util.py:62-84: 0 = missing, else 1 + bucket, offset by the rows of the preceding features."""
    emb_ids = []
    base = 0
    for x in feature_yaml:
        value = feature_dict.get(x['name'], None)
        if value is None:
            k = 0
        else:
            if x['type'] == 'numeric':
                value = max(x['min'], min(x['max'], value))
                k = int((value - x['min']) / x['interval'])
            elif x['type'] == 'bool':
                k = value
            else:
                try:
                    k = x['category'].index(value)
                except IndexError:
                    k = -1
            k += 1
        for _ in range(x.get("count", 1)):
            emb_ids.append(k + base)
            base += count_beatmap_features_embedding(x)
    return emb_ids


def count_beatmap_features(feature_yaml):
    return sum(count_beatmap_features_embedding(x) * x.get('count', 1) for x in feature_yaml)


def instantiate_from_config(config):
    """mug/util.py:93-108 -- the plugin seam: `target` dotted path + `params` kwargs."""
    if "target" not in config:
        if config in ('__is_first_stage__', '__is_unconditional__'):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**config.get("params", dict()))


def get_obj_from_str(string, reload=False):
    module, cls = string.rsplit(".", 1)
    if reload:
        importlib.reload(importlib.import_module(module))
    return getattr(importlib.import_module(module, package=None), cls)


def load_dict_from_batch(dict_data, i):
    return {k: (v[i].item() if isinstance(v, torch.Tensor) else v[i]) for k, v in dict_data.items()}


def _decode_audio(audio_path, sr, max_duration):
    """Decoding stays on the host, as in the reference (librosa.load through audioread/soundfile,
    mug/util.py:120-131), with whichever decoder is installed.  With librosa the call is the reference's own
    (its resampler included); without it the file is decoded at its native rate and converted to `sr` on the
    device (mugd_resample_poly = librosa's res_type="polyphase"), and the result is a device tensor."""
    try:
        import librosa
        y, _ = librosa.load(audio_path, sr=sr, duration=max_duration)
        return np.asarray(y, dtype=np.float32)
    except ImportError:
        pass
    try:
        import soundfile
        y, file_sr = soundfile.read(audio_path, dtype="float32", always_2d=True)
    except ImportError as e:
        raise RuntimeError("no audio decoder available (install librosa or soundfile)") from e
    y = y.mean(axis=1)
    if max_duration is not None:
        y = y[: int(max_duration * file_sr)]
    if file_sr != sr:
        from mug._native import get_lib
        return get_lib().resample_poly(torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32)), int(sr), int(file_sr))
    return y


def pcm_to_log_mel(y, n_mels, audio_hop_length, n_fft, sr):
    """The arithmetic of load_audio_without_cache after decoding (mug/util.py:138-143), on the GPU:
    log1p(mel power spectrogram) rounded to float16.  Returns a numpy float16 (n_mels, frames) array
    like the reference."""
    from mug._native import get_lib
    lib = get_lib()
    pcm = y if isinstance(y, torch.Tensor) else torch.as_tensor(np.asarray(y, dtype=np.float32))
    mel = lib.log_mel(pcm, sr=sr, n_fft=n_fft, hop=audio_hop_length, n_mels=n_mels)
    return mel.cpu().numpy().astype(np.float16)


def load_audio_without_cache(audio_path, n_mels, audio_hop_length, n_fft, sr, max_duration):
    """mug/util.py:133-144."""
    y = _decode_audio(audio_path, sr, max_duration)
    return pcm_to_log_mel(y, n_mels, audio_hop_length, n_fft, sr)


def load_audio(cache_dir, audio_path, n_mels, audio_hop_length, n_fft, sr, max_duration):
    import os
    audio_path = audio_path.strip()
    if cache_dir is None:
        return load_audio_without_cache(audio_path, n_mels, audio_hop_length, n_fft, sr, max_duration)
    cache_name = f"{os.path.basename(os.path.dirname(audio_path))}-{os.path.basename(audio_path)}.npz"
    cache_path = os.path.join(cache_dir, cache_name)
    if os.path.isfile(cache_path):
        return np.load(cache_path)['y']
    y = load_audio_without_cache(audio_path, n_mels, audio_hop_length, n_fft, sr, max_duration)
    np.savez_compressed(cache_path, y=y)
    return y
