"""The discriminator's spectrogram (torchaudio.transforms.Spectrogram(power=None, normalized=True, center=False) in the reference,
models/encodec.py:74-76 — torchaudio is not in the reference tree and not installed here) pinned against an INDEPENDENT source:

  * the closed-form spectrum of bin-centred sinusoids under the periodic Hann window: for x[n] = A cos(2 pi k0 n / N + phi), frame m
    (start m * hop) has X_m[k0] = A N e^{i theta_m} / 4, X_m[k0 +- 1] = -A N e^{i theta_m} / 8, zero elsewhere (2 <= k0 <= N/2 - 2),
    theta_m = phi + 2 pi k0 m hop / N; the transform divides by ||w|| = sqrt(3 N / 8);
  * a direct float64 evaluation of the definition sum_n w[n] x[n + m hop] e^{-2 pi i k n / N} / ||w|| for an arbitrary signal.

Checked for the CPU oracle (oracle/disc_oracle.spectrogram) and for the native kernel (csrc/stft.hip sat_spec_fwd): simulator at the
tiny scales, GPU at the five configured scales (n_fft 2048 .. 128, hop n_fft / 4)."""
import math

import numpy as np
import pytest
import torch

import disc_oracle


def _closed_form(n_fft, hop, frames, comps):
    """comps: [(A, k0, phi)] -> complex (frames, n_fft/2+1) float64."""
    z = np.zeros((frames, n_fft // 2 + 1), dtype=np.complex128)
    m = np.arange(frames)
    for a, k0, phi in comps:
        assert 2 <= k0 <= n_fft // 2 - 2
        e = np.exp(1j * (phi + 2 * math.pi * k0 * m * hop / n_fft))
        z[:, k0] += a * n_fft / 4 * e
        z[:, k0 - 1] -= a * n_fft / 8 * e
        z[:, k0 + 1] -= a * n_fft / 8 * e
    return z / math.sqrt(3 * n_fft / 8)


def _signal(n_fft, t, comps):
    n = np.arange(t)
    return sum(a * np.cos(2 * math.pi * k0 * n / n_fft + phi) for a, k0, phi in comps)


def _definition(x, n_fft, hop, frames_idx):
    """Direct O(N^2) DFT of the windowed frames (float64): rows = the requested frames."""
    n = np.arange(n_fft)
    w = 0.5 - 0.5 * np.cos(2 * math.pi * n / n_fft)                       # periodic Hann (torch.hann_window default)
    k = np.arange(n_fft // 2 + 1)
    basis = np.exp(-2j * math.pi * np.outer(k, n) / n_fft)
    return np.stack([basis @ (w * x[m * hop:m * hop + n_fft]) for m in frames_idx]) / math.sqrt((w ** 2).sum())


def _comps(n_fft):
    return [(0.7, 2, 0.3), (0.25, n_fft // 4 + 1, -1.1), (0.1, n_fft // 2 - 2, 2.0)]


def _native(ops, x, n_fft, hop, device):
    """native planes (NI, 2C, frames, bins) -> complex (C, frames, bins)"""
    z = ops.spec_fwd(torch.from_numpy(x).float().to(device).contiguous(), n_fft, hop).cpu().double().numpy()[0]
    c = z.shape[0] // 2
    return z[:c] + 1j * z[c:]


@pytest.mark.parametrize("n_fft", [32, 128, 2048])
def test_oracle_spectrogram_closed_form_and_definition(n_fft):
    hop, t = n_fft // 4, n_fft * 6 + 5
    frames = (t - n_fft) // hop + 1
    x = _signal(n_fft, t, _comps(n_fft))
    z = disc_oracle.spectrogram(torch.from_numpy(x)[None, None], n_fft, hop, n_fft)[0, 0].numpy().T       # (frames, bins)
    ref = _closed_form(n_fft, hop, frames, _comps(n_fft))
    assert z.shape == ref.shape
    assert np.abs(z - ref).max() < 1e-10 * np.abs(ref).max()
    rng = np.random.default_rng(3)
    y = rng.standard_normal(t)
    zy = disc_oracle.spectrogram(torch.from_numpy(y)[None, None], n_fft, hop, n_fft)[0, 0].numpy().T
    idx = [0, frames // 2, frames - 1]
    d = _definition(y, n_fft, hop, idx)
    assert np.abs(zy[idx] - d).max() < 1e-10 * np.abs(d).max()


def _native_case(ops, device, n_fft, t):
    hop = n_fft // 4
    frames = (t - n_fft) // hop + 1
    comps = _comps(n_fft)
    x = np.stack([_signal(n_fft, t, comps), _signal(n_fft, t, [(0.5, 3, 1.0)])])[None]      # stereo: second channel another tone
    z = _native(ops, x, n_fft, hop, device)
    ref0 = _closed_form(n_fft, hop, frames, comps)
    ref1 = _closed_form(n_fft, hop, frames, [(0.5, 3, 1.0)])
    for got, ref in ((z[0], ref0), (z[1], ref1)):
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() < 2e-6 * np.abs(ref).max()
    rng = np.random.default_rng(5)
    y = rng.standard_normal((1, 2, t)) * 0.3
    zy = _native(ops, y, n_fft, hop, device)
    idx = [0, frames // 3, frames - 1]
    for c in range(2):
        d = _definition(y[0, c], n_fft, hop, idx)
        assert np.abs(zy[c][idx] - d).max() < 2e-6 * np.abs(d).max()


@pytest.mark.parametrize("n_fft", [32, 128])
def test_native_spectrogram_closed_form_simulator(emu, n_fft):
    _native_case(emu, "cpu", n_fft, n_fft * 5 + 3)


@pytest.mark.gpu
@pytest.mark.parametrize("n_fft", [2048, 1024, 512, 256, 128])
def test_native_spectrogram_closed_form_gpu(hip, n_fft):
    _native_case(hip, "cuda", n_fft, 65536)
