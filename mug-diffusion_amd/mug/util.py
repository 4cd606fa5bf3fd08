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
        if str(audio_path).lower().endswith((".wav", ".wave")):
            y, file_sr = _read_wav(audio_path)          # RIFF PCM needs no third-party decoder
        else:
            raise RuntimeError("no audio decoder available for %s (install librosa or soundfile; .wav files are read natively)" % audio_path) from e
    y = y.mean(axis=1)
    if max_duration is not None:
        y = y[: int(max_duration * file_sr)]
    if file_sr != sr:
        from mug._native import get_lib
        return get_lib().resample_poly(torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32)), int(sr), int(file_sr))
    return y


def _read_wav(path):
    """RIFF/WAVE PCM (8 / 16 / 24 / 32-bit integer or 32-bit float) -> (float32 (frames, channels) in [-1, 1), sample rate),
    with the standard library only: the fallback decoder when neither librosa nor soundfile is installed.  Scaling follows
    libsndfile's float conversion (integer / 2^(bits-1)), which is what soundfile / librosa would return."""
    import struct
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise RuntimeError("%s is not a RIFF/WAVE file" % path)
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
            if fmt[0] == 0xFFFE and len(body) >= 26:          # WAVE_FORMAT_EXTENSIBLE: the real format tag is in the sub-format GUID
                fmt = (struct.unpack("<H", body[24:26])[0],) + fmt[1:]
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise RuntimeError("%s: missing fmt / data chunk" % path)
    tag, ch, rate, _, _, bits = fmt
    if tag == 3 and bits == 32:
        y = np.frombuffer(pcm, dtype="<f4").astype(np.float32)
    elif tag == 1 and bits == 16:
        y = np.frombuffer(pcm, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 8:
        y = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif tag == 1 and bits == 32:
        y = (np.frombuffer(pcm, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif tag == 1 and bits == 24:
        b = np.frombuffer(pcm[: len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        y = (v.astype(np.float64) / 8388608.0).astype(np.float32)
    else:
        raise RuntimeError("%s: unsupported WAVE format (tag %d, %d bits)" % (path, tag, bits))
    n = len(y) // ch
    return y[: n * ch].reshape(n, ch), int(rate)


def pcm_to_log_mel(y, n_mels, audio_hop_length, n_fft, sr, pad_mode=None):
    """The arithmetic of load_audio_without_cache after decoding (mug/util.py:138-143), on the GPU:
    log1p(mel power spectrogram) rounded to float16.  Returns a numpy float16 (n_mels, frames) array
    like the reference.  pad_mode: how the centred STFT frames are padded -- 'constant' (librosa >= 0.10) or 'reflect'
    (librosa <= 0.9; the reference leaves librosa unpinned, requirements.txt:8); None = the library context's setting
    (Lib.set_mel_pad_mode / MUGD_MEL_PAD, default 'constant')."""
    from mug._native import get_lib
    lib = get_lib()
    pcm = y if isinstance(y, torch.Tensor) else torch.as_tensor(np.asarray(y, dtype=np.float32))
    mel = lib.log_mel(pcm, sr=sr, n_fft=n_fft, hop=audio_hop_length, n_mels=n_mels, pad_mode=pad_mode)
    return mel.cpu().numpy().astype(np.float16)


def load_audio_without_cache(audio_path, n_mels, audio_hop_length, n_fft, sr, max_duration, pad_mode=None):
    """mug/util.py:133-144 (pad_mode: see pcm_to_log_mel; the reference's positional signature is unchanged)."""
    y = _decode_audio(audio_path, sr, max_duration)
    return pcm_to_log_mel(y, n_mels, audio_hop_length, n_fft, sr, pad_mode=pad_mode)


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
