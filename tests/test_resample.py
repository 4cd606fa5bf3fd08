"""Device sample-rate conversion in front of the log-mel kernel: `mugd_resample_poly` against the
oracle's restatement and against its specification, scipy.signal.resample_poly (= librosa's res_type="polyphase"; the
reference's own resampler inside librosa.load is unpinned, so this stage is parity-unpinned against the reference).

Tolerances: product vs oracle -- both accumulate exact float32 products in float64 and round once, only the order and
the last bits of the Kaiser / sinc taps (C++ series vs NumPy's i0) can differ: 1 float32 ulp of the signal's peak.  Oracle
and product vs scipy -- scipy accumulates ~40 terms in float32: 3e-6 of the peak."""
import numpy as np
import pytest
import torch
from scipy import signal

from oracle import host

RATIOS = [(1, 2), (147, 320), (2, 1), (3, 2), (441, 960), (2, 4)]


def noise(seed, n):
    g = np.random.default_rng(seed)
    t = np.arange(n)
    return (0.6 * np.sin(2 * np.pi * 0.01 * t) + 0.3 * g.standard_normal(n)).astype(np.float32)


@pytest.mark.parametrize("up,down", RATIOS)
def test_oracle_follows_scipy(up, down):
    for n in (1, 7, 1000, 4097):
        x = noise(n, n)
        want = signal.resample_poly(x, up, down)
        got = host.resample_poly(x, up, down)
        assert got.dtype == np.float32 and got.shape == want.shape, (n, got.shape, want.shape)
        assert np.abs(got - want).max() <= 3e-6 * max(1.0, np.abs(x).max())
    g = int(np.gcd(up, down))
    taps, half = host.resample_taps(up // g, down // g)
    ref_taps = signal.firwin(2 * half + 1, 1.0 / max(up // g, down // g), window=("kaiser", 5.0)).astype(np.float32) * (up // g)
    assert np.array_equal(taps, ref_taps)


@pytest.mark.parametrize("up,down", RATIOS)
def test_device_resampler_matches_oracle_and_scipy(lib, up, down):
    for n in (1, 5, 300, 2000, 5001):
        x = noise(100 + n, n)
        got = lib.resample_poly(torch.from_numpy(x), up, down).cpu().numpy()
        want = host.resample_poly(x, up, down)
        peak = max(1.0, float(np.abs(x).max()))
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 1.2e-7 * peak, (n, np.abs(got - want).max())
        assert np.abs(got - signal.resample_poly(x, up, down)).max() <= 3e-6 * peak


def test_identity_and_errors(lib):
    x = torch.from_numpy(noise(1, 100))
    assert torch.equal(lib.resample_poly(x, 3, 3).cpu(), x)
    with pytest.raises(Exception):
        lib.resample_poly(x, 0, 2)
    with pytest.raises(Exception):
        lib.resample_poly(x, 1, 5000)           # a ratio whose input window per block exceeds what the kernel stages


@pytest.mark.gpu
def test_three_minutes_of_44k1_audio(gpu_lib):
    """configs[1]'s ingest: 180 s at 44.1 kHz -> 22.05 kHz on the device.  Size-independent properties: a 1 kHz tone
    keeps its amplitude, a 15 kHz tone (above the new Nyquist) is rejected, and the result agrees with the oracle on
    a slice from the middle (computed from the matching input slice)."""
    sr, n = 44100, 180 * 44100
    t = np.arange(n, dtype=np.float64) / sr
    low = (0.5 * np.sin(2 * np.pi * 1000 * t)).astype(np.float32)
    high = (0.5 * np.sin(2 * np.pi * 15000 * t)).astype(np.float32)
    y_low = gpu_lib.resample_poly(torch.from_numpy(low), 1, 2).cpu().numpy()
    y_high = gpu_lib.resample_poly(torch.from_numpy(high), 1, 2).cpu().numpy()
    assert len(y_low) == n // 2
    mid = slice(1000, -1000)
    assert abs(np.abs(y_low[mid]).max() - 0.5) < 2e-3
    assert np.abs(y_high[mid]).max() < 0.5 * 10 ** (-40 / 20)
    a, b = 2_000_000, 2_010_000                    # output samples [a, b) depend on inputs [2a - 20, 2b + 20)
    seg = host.resample_poly(low[2 * a - 40:2 * b + 40], 1, 2)
    assert np.abs(seg[20:20 + (b - a)] - y_low[a:b]).max() <= 1.2e-7
