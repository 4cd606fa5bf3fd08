"""TEST INFRASTRUCTURE (see oracle/__init__.py).

Host-side pieces of the path, numpy only.

* mel front-end: the reference calls the THIRD-PARTY `librosa.feature.melspectrogram`
  (mug/util.py:133-144; librosa is unpinned in requirements.txt:8 and absent from
  /root/reference and from this image).  This is a restatement of librosa's
  published algorithm (librosa >= 0.10 defaults): centred STFT with zero
  ("constant") padding of n_fft//2, periodic Hann window of n_fft, hop n_fft//4,
  power 2, Slaney mel filterbank (htk=False) with Slaney area normalisation,
  fmin=0, fmax=sr/2; then log1p and a cast to float16.  PARITY UNPINNED: no
  fixture in the reference fixes pad mode or any golden mel; the restatement is
  cross-checked against torch.stft in tests.
* length rule, prompt ids, note threshold: restated from webui.py:349-367,
  mug/util.py:51-84, mug/data/convertor.py:212-264.
"""
import math

import numpy as np


# ---------------------------------------------------------------- mel ---------

def hz_to_mel(f):
    """Slaney mel scale (librosa.hz_to_mel, htk=False)."""
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr=22050, n_fft=512, n_mels=128, fmin=0.0, fmax=None):
    """librosa.filters.mel (norm='slaney', htk=False) -> float32 (n_mels, 1+n_fft//2)."""
    if fmax is None:
        fmax = sr / 2.0
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32)


LIBROSA_TARGET = "librosa 0.10.x defaults (stft: center=True, pad_mode='constant', window='hann', dtype complex64; melspectrogram: power=2.0, " \
                 "norm='slaney', htk=False)"     # the ONE place the restated third-party version is named (requirements.txt:8 does not pin it)


def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True): float64, periodic."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def n_frames(n_samples, hop=128):
    return 1 + n_samples // hop


def log_mel(y, sr=22050, n_fft=512, hop=128, n_mels=128, out_dtype=np.float16, pad_mode="constant"):
    """load_audio_without_cache's arithmetic (mug/util.py:138-143) on mono float32 PCM already at `sr`:
    (n_mels, 1 + len(y)//hop), log1p(power mel) cast to float16.  PARITY UNPINNED against the reference (librosa is not
    installed and the reference ships no fixture); restates LIBROSA_TARGET with librosa's own precision path:
      stft      float64 window x float32 frame -> float64, rfft in double, result CAST to complex64   (librosa/core/spectrum.py: stft)
      power     np.abs(complex64) ** 2.0 in float32                                                      (_spectrogram)
      mel       float32 filterbank (filters.mel, Slaney) times float32 power                             (melspectrogram, einsum)
    pad_mode: librosa >= 0.10 pads the centred frames with zeros ('constant'); 0.9 and older reflected ('reflect')."""
    y = np.asarray(y, dtype=np.float32)
    pad = n_fft // 2
    yp = np.pad(y, (pad, pad), mode=pad_mode)
    nf = n_frames(len(y), hop)
    win = hann_periodic(n_fft)                                             # float64
    idx = np.arange(n_fft)[None, :] + hop * np.arange(nf)[:, None]
    frames = yp[idx].astype(np.float64) * win[None, :]
    spec = np.fft.rfft(frames, axis=1).astype(np.complex64)
    power = np.abs(spec) ** 2.0                                            # float32
    mel = mel_filterbank(sr, n_fft, n_mels) @ power.T                      # (n_mels, nf) float32
    return np.log1p(mel).astype(out_dtype)


def synth_audio(seconds, sr=22050, seed=0):
    """SURVEY.md 8(d) synthetic signal: 0.5 sin(2pi 440 n/sr) + 0.25 sin(2pi (110 + n/N 3000) n/sr)
    + N(0, 0.01) noise, float32 at `sr`."""
    N = int(round(seconds * sr))
    n = np.arange(N, dtype=np.float64)
    y = 0.5 * np.sin(2 * np.pi * 440.0 * n / sr) + 0.25 * np.sin(2 * np.pi * (110.0 + n / N * 3000.0) * n / sr)
    y = y + np.random.default_rng(seed).normal(0.0, 0.01, N)
    return y.astype(np.float32)


# ------------------------------------------------------- sample-rate conversion ----------

def resample_taps(up, down):
    """The low-pass of scipy.signal.resample_poly (signaltools: `firwin(2 * half_len + 1, 1 / max(up, down),
    window=('kaiser', 5.0))`, half_len = 10 max(up, down)) written out: windowed sinc with unit DC gain, rounded to
    float32 as scipy does for float32 input, times `up`."""
    m = max(up, down)
    half = 10 * m
    n = 2 * half + 1
    t = np.arange(n) - half
    h = (1.0 / m) * np.sinc(t / m) * np.kaiser(n, 5.0)
    h = (h / h.sum()).astype(np.float32)
    return h * np.float32(up), half


def resample_poly(x, up, down):
    """librosa.resample(res_type='polyphase') == scipy.signal.resample_poly(x, up, down) (the
    reference resamples inside librosa.load(sr=22050), mug/util.py:126, with an unpinned backend -> PARITY UNPINNED against
    the reference itself; pinned against scipy's function in tests/test_resample.py):
        y[m] = sum_i x[i] h[m down + half - i up],  m < ceil(n up / down),
    products accumulated in float64 and rounded to float32 once (scipy accumulates in float32: agreement ~1e-6)."""
    g = math.gcd(int(up), int(down))
    up, down = int(up) // g, int(down) // g
    x = np.asarray(x, dtype=np.float32)
    if up == down == 1:
        return x.copy()
    h, half = resample_taps(up, down)
    n_out = -(-len(x) * up // down)
    stuffed = np.zeros(len(x) * up, dtype=np.float64)
    stuffed[::up] = x
    full = np.convolve(stuffed, h.astype(np.float64))          # full[k] = sum_i x[i] h[k - i up]
    idx = np.arange(n_out) * down + half
    return full[idx].astype(np.float32)


# ------------------------------------------------------- length rule ----------

def z_length_for(n_mel_frames, max_audio_frame=32768, z_length=512):
    """webui.py:349-356: ratio = max_audio_frame // z_length (=64);
    z = (int(T_a / ratio / 32) + 1) * 32; audio padded/truncated to z * ratio."""
    ratio = max_audio_frame // z_length
    z = (int(n_mel_frames / ratio / 32) + 1) * 32
    return z, z * ratio


def pad_or_trunc_mel(mel, target):
    """webui.py:358-367: zero-pad (exact 0.0) or truncate along time to `target` frames."""
    t = mel.shape[1]
    if t < target:
        return np.concatenate([mel, np.zeros((mel.shape[0], target - t), dtype=np.float32)], axis=1)
    return mel[:, :target]


# --------------------------------------------------------- prompt ids ---------

def _feat_count(x):
    """mug/util.py:51-60."""
    if x["type"] == "numeric":
        return int(math.ceil((x["max"] - x["min"]) / x["interval"])) + 1
    if x["type"] == "category":
        return len(x["category"]) + 1
    if x["type"] == "bool":
        return 3
    raise ValueError(str(x))


def feature_ids(feature_dict, feature_yaml):
    """mug/util.py:62-84 (feature_dict_to_embedding_ids)."""
    ids = []
    base = 0
    for x in feature_yaml:
        v = feature_dict.get(x["name"], None)
        if v is None:
            k = 0
        else:
            if x["type"] == "numeric":
                v = max(x["min"], min(x["max"], v))
                k = int((v - x["min"]) / x["interval"])
            elif x["type"] == "bool":
                k = v
            else:
                k = x["category"].index(v)   # reference raises ValueError on unknown categories too
            k += 1
        for _ in range(x.get("count", 1)):
            ids.append(k + base)
            base += _feat_count(x)
    return ids


def feature_table_rows(feature_yaml):
    """mug/util.py:86-90."""
    return sum(_feat_count(x) * x.get("count", 1) for x in feature_yaml)


# ------------------------------------------------------ note threshold --------

def note_grid(logits, key_count=4):
    """convertor.py:212-216 + :232-264: with from_logits=True a cell is on iff logit > 0
    (strict).  Returns (is_start, is_holding) boolean arrays (key_count, T) for
    logits (16, T) laid out [is_start x4 | offset_start x4 | is_holding x4 | offset_end x4]."""
    logits = np.asarray(logits)
    return logits[..., 0:key_count, :] > 0, logits[..., 2 * key_count:3 * key_count, :] > 0


def array_to_objects(note_array, frame_ms, key_count=4):
    """convertor.py:232-264, restated (hit-object strings for 4K mania)."""
    a = np.asarray(note_array).transpose()
    col_w = int(512 / key_count)
    out = []
    n = len(a)
    for col in range(key_count):
        for s in np.where(a[:, col] > 0)[0]:
            so = np.clip(a[s, col + key_count], 0, 1)
            start = int(round((s + so) * frame_ms))
            end = -1
            if s != n - 1:
                i = s + 1
                while i < n and a[i, col + 2 * key_count] > 0 and not a[i, col] > 0:
                    i += 1
                e = i - 1
                if e != s:
                    eo = np.clip(a[e, col + 3 * key_count], 0, 1)
                    end = int(round((e + eo) * frame_ms))
            x = int(round((col + 0.5) * col_w))
            line = "%d,192,%d,1,0,0:0:0:0:" % (x, start) if end == -1 else "%d,192,%d,128,0,%d:0:0:0:0:" % (x, start, end)
            out.append((line, start))
    out.sort(key=lambda r: r[1])
    return [r[0] for r in out]
