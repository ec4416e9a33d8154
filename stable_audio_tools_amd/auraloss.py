"""MI355X-native multi-resolution STFT loss — same class names / constructor surface / call
convention as the reference's vendored auraloss
(stable_audio_tools/training/losses/auraloss.py: STFTLoss :226, MultiResolutionSTFTLoss :451,
SumAndDifferenceSTFTLoss :542; FIRFilter "aw" :117-149).

    loss_module(input, target) -> scalar          (input/target: (B, C, T))

Only the configuration the autoencoder training wrapper uses is implemented on the HIP path
(w_sc = w_log_mag = 1, w_lin_mag = w_phs = 0, scale=None, reduction='mean', output='loss',
hann window with win_length == fft_size, optional A-weighting); anything else raises.
Forward and backward both run in csrc/stft.hip — no spectrogram is ever written to HBM.

`AutoencoderSpectralLoss` is the fused form of what the training wrapper assembles for stereo
models (training/autoencoders.py:186-194): sum/difference + left + right evaluated from ONE pass of
the A-weighting filter per signal and one kernel launch per resolution for all four views.
"""
import numpy as np
import scipy.signal
import torch

from . import ops as _ops_mod
from . import functional as _fn


def _aweighting_taps(fs, ntaps=101):
    """Filter design exactly as FIRFilter(filter_type='aw') does it at construction time
    (auraloss.py:117-149): IEC/CD 1672 analog prototype -> bilinear -> freqz(512) -> firls(101) -> float32.
    Host-side, once per module (scipy, like the reference)."""
    f1, f2, f3, f4, a1000 = 20.598997, 107.65265, 737.86223, 12194.217, 1.9997
    nums = [(2 * np.pi * f4) ** 2 * (10 ** (a1000 / 20)), 0, 0, 0, 0]
    dens = np.polymul([1, 4 * np.pi * f4, (2 * np.pi * f4) ** 2], [1, 4 * np.pi * f1, (2 * np.pi * f1) ** 2])
    dens = np.polymul(np.polymul(dens, [1, 2 * np.pi * f3]), [1, 2 * np.pi * f2])
    b, a = scipy.signal.bilinear(nums, dens, fs=fs)
    w_iir, h_iir = scipy.signal.freqz(b, a, worN=512, fs=fs)
    taps = scipy.signal.firls(ntaps, w_iir, abs(h_iir), fs=fs)
    return torch.tensor(taps.astype("float32"))


_VIEW_W = {}


def _view_weights(view_w, device):
    """The per-view weights as a device tensor, built once per (weights, device): creating it from the host list at every call is a
    pageable host->device copy — a synchronisation inside every step, and not capturable into a HIP graph (training.GraphedTrainStep)."""
    key = (tuple(float(v) for v in view_w), str(device))
    t = _VIEW_W.get(key)
    if t is None:
        with torch.inference_mode(False):
            t = torch.tensor(key[0], dtype=torch.float32, device=device)
        _VIEW_W[key] = t
    return t


class _MRSTFTFn(torch.autograd.Function):
    """total = sum_v w_v * mean_r [ mean_i sqrt(S1/S2) + sum_i S3 / (NI * bins * frames) ]
    x, y: (NI, C, T); views (NV, 2); view_w (NV,) python floats."""

    @staticmethod
    def forward(ctx, x, y, views, view_w, taps, fft_sizes, hop_sizes, ops):
        ops = _fn._ops(ops)
        ni, c, t = x.shape
        x = x.contiguous()
        y = y.contiguous()
        if taps is not None:
            xf = ops.fir(x.view(ni * c, t), taps).view(ni, c, t)
            yf = ops.fir(y.view(ni * c, t), taps).view(ni, c, t)
        else:
            xf, yf = x, y
        nres = len(fft_sizes)
        vw = _view_weights(view_w, x.device)
        # all resolutions' kernels first, then ONE pass of scalar arithmetic over the stacked sums (round 6: ten tiny launches per
        # resolution sat between the STFT kernels before — 140 of the generator step's 233 torch launches with the backward's)
        sums_all = torch.stack([ops.stft_sums(xf, yf, views, n, h) for n, h in zip(fft_sizes, hop_sizes)])      # (R, NI, NV, 3)
        inv_cnt = _view_weights([1.0 / float(ni * (n // 2 + 1) * (1 + t // h)) for n, h in zip(fft_sizes, hop_sizes)], x.device).view(-1, 1)
        sc = torch.sqrt(sums_all[..., 0] / sums_all[..., 1])        # (R, NI, NV)
        per_view = sc.mean(1) + sums_all[..., 2].sum(1) * inv_cnt   # (R, NV)
        total = (per_view * vw).sum() / nres
        ctx.ops = ops
        ctx.meta = (fft_sizes, hop_sizes, view_w, taps is not None)
        ctx.save_for_backward(xf, yf, views, taps, sums_all)
        return total

    @staticmethod
    def backward(ctx, g):
        ops = ctx.ops
        fft_sizes, hop_sizes, view_w, has_taps = ctx.meta
        xf, yf, views, taps, sums_all = ctx.saved_tensors
        ni, c, t = xf.shape
        nres = len(fft_sizes)
        vw = _view_weights(view_w, xf.device)
        grads = [None, None]
        # the three per-(item, view) coefficients of every resolution in one pass: c1 = scale / (NI sc S2), c2 = sc^2, c3 = scale / count
        inv_cnt = _view_weights([1.0 / float(ni * (n // 2 + 1) * (1 + t // h)) for n, h in zip(fft_sizes, hop_sizes)], xf.device).view(-1, 1, 1)
        sc = torch.sqrt(sums_all[..., 0] / sums_all[..., 1])        # (R, NI, NV)
        scale = (g * vw / nres).view(1, 1, -1)                      # (1, 1, NV)
        coef_all = torch.stack([scale / (ni * sc * sums_all[..., 1]), sc * sc, (scale * inv_cnt).expand_as(sc)], dim=-1).contiguous()
        for which in (0, 1):          # 0: d/dx (first argument), 1: d/dy (second argument)
            if not ctx.needs_input_grad[which]:
                continue
            # one set of 4 write-once planes per resolution (csrc/stft.hip: no atomics), summed in a fixed order: the
            # gradient is bit-reproducible run to run
            planes = torch.zeros((nres, 4) + tuple(yf.shape), dtype=yf.dtype, device=yf.device)
            for ri, (n, h) in enumerate(zip(fft_sizes, hop_sizes)):
                ops.stft_backward(xf, yf, views, coef_all[ri], planes[ri], n, h, wrt_x=(which == 0))
            acc = planes.sum(dim=(0, 1))
            if has_taps:
                acc = ops.fir(acc.view(ni * c, t), taps, adjoint=True).view(ni, c, t)
            grads[which] = acc
        return grads[0], grads[1], None, None, None, None, None, None


def _check_supported(fft_size, hop_size, win_length, window, w_sc, w_log_mag, w_lin_mag, w_phs, scale, scale_invariance,
                     output, reduction, mag_distance, kwargs):
    bad = []
    if window != "hann_window":
        bad.append(f"window={window!r}")
    if win_length != fft_size:
        bad.append("win_length != fft_size")
    if (w_sc, w_log_mag, w_lin_mag, w_phs) != (1.0, 1.0, 0.0, 0.0):
        bad.append("loss weights other than w_sc=w_log_mag=1, w_lin_mag=w_phs=0")
    if scale is not None or scale_invariance:
        bad.append("mel/chroma scale or scale_invariance")
    if output != "loss" or reduction != "mean" or mag_distance != "L1":
        bad.append("output/reduction/mag_distance other than the defaults")
    if kwargs:
        bad.append(f"extra kwargs {sorted(kwargs)}")
    if bad:
        raise NotImplementedError("stable_audio_tools_amd.auraloss: not on the HIP path: " + "; ".join(bad))


class STFTLoss(torch.nn.Module):
    def __init__(self, fft_size=1024, hop_size=256, win_length=1024, window="hann_window", w_sc=1.0, w_log_mag=1.0,
                 w_lin_mag=0.0, w_phs=0.0, sample_rate=None, scale=None, n_bins=None, perceptual_weighting=False,
                 scale_invariance=False, eps=1e-8, output="loss", reduction="mean", mag_distance="L1", device=None,
                 retain_batch_dim=False, **kwargs):
        super().__init__()
        _check_supported(fft_size, hop_size, win_length, window, float(w_sc), float(w_log_mag), float(w_lin_mag), float(w_phs),
                         scale, scale_invariance, output, reduction, mag_distance, kwargs)
        if eps != 1e-8 or retain_batch_dim:
            raise NotImplementedError("eps != 1e-8 / retain_batch_dim are not on the HIP path")
        self.fft_size, self.hop_size, self.win_length = fft_size, hop_size, win_length
        self.sample_rate = sample_rate
        self.perceptual_weighting = perceptual_weighting
        if perceptual_weighting:
            if sample_rate is None:
                raise ValueError("`sample_rate` must be supplied when `perceptual_weighting = True`.")
            self.register_buffer("aw_taps", _aweighting_taps(sample_rate), persistent=False)
        else:
            self.aw_taps = None
        self.register_buffer("views", torch.tensor([[1.0, 0.0]]), persistent=False)

    def forward(self, input, target):
        bs, chs, t = input.shape
        # the reference folds channels into the batch for both the filter and the STFT (auraloss.py:400-417)
        x = input.reshape(bs * chs, 1, t)
        y = target.reshape(bs * chs, 1, t)
        taps = self.aw_taps.to(x.device) if self.aw_taps is not None else None
        return _MRSTFTFn.apply(x, y, self.views.to(x.device), (1.0,), taps, (self.fft_size,), (self.hop_size,), None)


class MultiResolutionSTFTLoss(torch.nn.Module):
    def __init__(self, fft_sizes=[1024, 2048, 512], hop_sizes=[120, 240, 50], win_lengths=[600, 1200, 240],
                 window="hann_window", w_sc=1.0, w_log_mag=1.0, w_lin_mag=0.0, w_phs=0.0, sample_rate=None, scale=None,
                 n_bins=None, perceptual_weighting=False, scale_invariance=False, **kwargs):
        super().__init__()
        assert len(fft_sizes) == len(hop_sizes) == len(win_lengths)
        for fs, ss, wl in zip(fft_sizes, hop_sizes, win_lengths):
            _check_supported(fs, ss, wl, window, float(w_sc), float(w_log_mag), float(w_lin_mag), float(w_phs), scale,
                             scale_invariance, "loss", "mean", "L1", kwargs)
        self.fft_sizes, self.hop_sizes, self.win_lengths = tuple(fft_sizes), tuple(hop_sizes), tuple(win_lengths)
        self.sample_rate = sample_rate
        if perceptual_weighting:
            if sample_rate is None:
                raise ValueError("`sample_rate` must be supplied when `perceptual_weighting = True`.")
            self.register_buffer("aw_taps", _aweighting_taps(sample_rate), persistent=False)
        else:
            self.aw_taps = None
        self.register_buffer("views", torch.tensor([[1.0, 0.0]]), persistent=False)

    def forward(self, x, y):
        bs, chs, t = x.shape
        xi = x.reshape(bs * chs, 1, t)
        yi = y.reshape(bs * chs, 1, t)
        taps = self.aw_taps.to(x.device) if self.aw_taps is not None else None
        return _MRSTFTFn.apply(xi, yi, self.views.to(x.device), (1.0,), taps, self.fft_sizes, self.hop_sizes, None)


class SumAndDifferenceSTFTLoss(torch.nn.Module):
    def __init__(self, fft_sizes, hop_sizes, win_lengths, window="hann_window", w_sum=1.0, w_diff=1.0, output="loss",
                 **kwargs):
        super().__init__()
        if output != "loss":
            raise NotImplementedError("output='full' is not on the HIP path")
        self.w_sum, self.w_diff = float(w_sum), float(w_diff)
        self.mrstft = MultiResolutionSTFTLoss(fft_sizes, hop_sizes, win_lengths, window, **kwargs)
        self.register_buffer("views", torch.tensor([[1.0, 1.0], [1.0, -1.0]]), persistent=False)

    def forward(self, input, target):
        assert input.shape == target.shape
        if input.size(1) != 2:
            raise ValueError(f"Input must be stereo: {input.size(1)} channel(s).")
        m = self.mrstft
        taps = m.aw_taps.to(input.device) if m.aw_taps is not None else None
        # loss = (w_sum * mrstft(sum) + w_diff * mrstft(diff)) / 2      (auraloss.py:608-610)
        return _MRSTFTFn.apply(input, target, self.views.to(input.device), (self.w_sum / 2, self.w_diff / 2), taps,
                               m.fft_sizes, m.hop_sizes, None)


class AutoencoderSpectralLoss(torch.nn.Module):
    """Fused stereo reconstruction loss of the AE training wrapper:
        weight * sdstft(reals, decoded) + weight/2 * lrstft(L) + weight/2 * lrstft(R)
    (training/autoencoders.py:186-194; argument order (reals, decoded) as AuralossLoss passes it,
    training/losses/losses.py:111).  Mono models: weight * mrstft(reals, decoded)."""

    def __init__(self, sample_rate, fft_sizes, hop_sizes, win_lengths, perceptual_weighting=False, weight=1.0, **kwargs):
        super().__init__()
        self.mrstft = MultiResolutionSTFTLoss(fft_sizes, hop_sizes, win_lengths, sample_rate=sample_rate,
                                              perceptual_weighting=perceptual_weighting, **kwargs)
        self.weight = float(weight)
        self.register_buffer("views2", torch.tensor([[1.0, 1.0], [1.0, -1.0], [1.0, 0.0], [0.0, 1.0]]), persistent=False)

    def forward(self, reals, decoded):
        m = self.mrstft
        taps = m.aw_taps.to(reals.device) if m.aw_taps is not None else None
        w = self.weight
        if reals.shape[1] == 2:
            return _MRSTFTFn.apply(reals, decoded, self.views2.to(reals.device), (w / 2, w / 2, w / 2, w / 2), taps,
                                   m.fft_sizes, m.hop_sizes, None)
        if reals.shape[1] == 1:
            return _MRSTFTFn.apply(reals, decoded, m.views.to(reals.device), (w,), taps, m.fft_sizes, m.hop_sizes, None)
        raise ValueError("AutoencoderSpectralLoss expects mono or stereo audio")
