"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product package).

CPU restatement of the reference's multi-resolution STFT loss
(/root/reference/stable_audio_tools/training/losses/auraloss.py, vendored auraloss freq.py) and of
the way the autoencoder training wrapper assembles it (training/autoencoders.py:142-146, :186-194,
training/losses/losses.py:107-113).  The STFT is written out explicitly (reflect pad, framing,
periodic Hann, rfft) rather than calling torch.stft, so it is an independent check.
Pinned against golden vectors generated from the reference (oracle/gen_golden.py).

Every function is dtype-generic: fed float64 tensors it is the "truth" used to measure the fp32
noise floor of the loss GRADIENT.  (The log-magnitude term divides by STFT magnitudes that the
A-weighting filter drives down to the 1e-4 clamp; the reference's own fp32 gradient is 2e-3 (L2) /
7e-4 (max-abs) away from the float64 result on the golden signal — tests/test_oracle_golden.py.)
"""
import numpy as np
import scipy.signal
import torch
import torch.nn.functional as F


def aweighting_fir_taps(fs, ntaps=101):
    """auraloss.py:117-149 (filter_type == "aw"): analog A-weighting (IEC/CD 1672) -> bilinear ->
    512-point response -> 101-tap least-squares FIR, stored as float32."""
    f1, f2, f3, f4 = 20.598997, 107.65265, 737.86223, 12194.217
    a1000 = 1.9997
    nums = [(2 * np.pi * f4) ** 2 * (10 ** (a1000 / 20)), 0, 0, 0, 0]
    dens = np.polymul([1, 4 * np.pi * f4, (2 * np.pi * f4) ** 2], [1, 4 * np.pi * f1, (2 * np.pi * f1) ** 2])
    dens = np.polymul(np.polymul(dens, [1, 2 * np.pi * f3]), [1, 2 * np.pi * f2])
    b, a = scipy.signal.bilinear(nums, dens, fs=fs)
    w_iir, h_iir = scipy.signal.freqz(b, a, worN=512, fs=fs)
    taps = scipy.signal.firls(ntaps, w_iir, abs(h_iir), fs=fs)
    return torch.tensor(taps.astype("float32"))


def fir_filter(x, taps):
    """auraloss.py:163-169: F.conv1d (cross-correlation) with zero padding ntaps//2.  x: (N, T)."""
    n = taps.numel()
    return F.conv1d(x.unsqueeze(1), taps.to(x.dtype).view(1, 1, -1), padding=n // 2).squeeze(1)


def hann_periodic(n, dtype=torch.float32):
    """torch.hann_window(n) default periodic=True (auraloss.py:23-41 get_window -> getattr(torch, 'hann_window'))."""
    i = torch.arange(n, dtype=torch.float64)
    return (0.5 - 0.5 * torch.cos(2.0 * np.pi * i / n)).to(dtype)


def stft_mag(x, n_fft, hop, win_length, eps=1e-8):
    """auraloss.py:368-395: torch.stft(center=True, pad_mode='reflect', onesided, unnormalised, window zero-
    padded to n_fft if shorter) then sqrt(clamp(re^2 + im^2, min=eps)).  x: (N, T) -> (N, n_fft/2+1, frames)."""
    win = hann_periodic(win_length, x.dtype)
    if win_length < n_fft:
        left = (n_fft - win_length) // 2
        win = F.pad(win, (left, n_fft - win_length - left))
    xp = F.pad(x.unsqueeze(1), (n_fft // 2, n_fft // 2), mode="reflect").squeeze(1)
    frames = xp.unfold(-1, n_fft, hop)                    # (N, frames, n_fft)
    spec = torch.fft.rfft(frames * win, dim=-1)           # (N, frames, n_fft/2+1)
    power = spec.real ** 2 + spec.imag ** 2
    return torch.sqrt(torch.clamp(power, min=eps)).transpose(1, 2)


def stft_loss(inp, tgt, n_fft, hop, win_length, taps=None):
    """STFTLoss.forward (auraloss.py:397-449) with the defaults the AE config uses: w_sc = w_log_mag = 1,
    w_lin_mag = w_phs = 0, reduction 'mean'.  inp/tgt: (B, C, T).
      sc  = ||y_mag - x_mag||_F / ||y_mag||_F   per (b, c) item          (auraloss.py:180-181)
      log = mean |log x_mag - log y_mag|        over everything          (auraloss.py:219-223)
      loss = mean_items(sc) + log                                        (auraloss.py:437-443)"""
    b, c, t = inp.shape
    x = inp.reshape(b * c, t)
    y = tgt.reshape(b * c, t)
    if taps is not None:  # perceptual_weighting (auraloss.py:400-411)
        x = fir_filter(x, taps)
        y = fir_filter(y, taps)
    x_mag = stft_mag(x, n_fft, hop, win_length)
    y_mag = stft_mag(y, n_fft, hop, win_length)
    sc = torch.linalg.matrix_norm(y_mag - x_mag) / torch.linalg.matrix_norm(y_mag)  # (B*C,)
    log = (torch.log(x_mag) - torch.log(y_mag)).abs().mean()
    return (sc + log).mean()


def mrstft_loss(inp, tgt, fft_sizes, hop_sizes, win_lengths, taps=None):
    """MultiResolutionSTFTLoss.forward (auraloss.py:519-539): mean over resolutions."""
    total = 0.0
    for n, h, w in zip(fft_sizes, hop_sizes, win_lengths):
        total = total + stft_loss(inp, tgt, n, h, w, taps)
    return total / len(fft_sizes)


def sum_and_difference_loss(inp, tgt, fft_sizes, hop_sizes, win_lengths, taps=None):
    """SumAndDifferenceSTFTLoss.forward (auraloss.py:588-615), w_sum = w_diff = 1."""
    def sd(x):
        return (x[:, 0:1] + x[:, 1:2]), (x[:, 0:1] - x[:, 1:2])
    isum, idiff = sd(inp)
    tsum, tdiff = sd(tgt)
    return (mrstft_loss(isum, tsum, fft_sizes, hop_sizes, win_lengths, taps)
            + mrstft_loss(idiff, tdiff, fft_sizes, hop_sizes, win_lengths, taps)) / 2


def autoencoder_spectral_loss(reals, decoded, stft_cfg, sample_rate, weight=1.0):
    """Stereo AE reconstruction loss as the training wrapper assembles it
    (training/autoencoders.py:142-146, :186-194, :423-427) through AuralossLoss, which passes
    (target, input) — i.e. x = reals, y = decoded (training/losses/losses.py:111):
        weight * sdstft(reals, decoded) + weight/2 * lrstft(reals_L, decoded_L) + weight/2 * lrstft(reals_R, decoded_R)
    Mono: weight * mrstft(reals, decoded)."""
    ffts, hops, wins = stft_cfg["fft_sizes"], stft_cfg["hop_sizes"], stft_cfg["win_lengths"]
    taps = aweighting_fir_taps(sample_rate) if stft_cfg.get("perceptual_weighting", False) else None
    if reals.shape[1] == 2:
        sd = sum_and_difference_loss(reals, decoded, ffts, hops, wins, taps)
        left = mrstft_loss(reals[:, 0:1], decoded[:, 0:1], ffts, hops, wins, taps)
        right = mrstft_loss(reals[:, 1:2], decoded[:, 1:2], ffts, hops, wins, taps)
        return weight * sd + (weight / 2) * left + (weight / 2) * right
    return weight * mrstft_loss(reals, decoded, ffts, hops, wins, taps)
