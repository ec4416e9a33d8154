"""Parity of the HIP multi-resolution STFT loss (csrc/stft.hip via stable_audio_tools_amd.auraloss)
against golden values produced by the reference's auraloss and against the oracle.

Loss VALUES are held to the 1e-3 bar of BASELINE.json (they agree to ~1e-5).  The loss GRADIENT
w.r.t. the decoded signal is ill-conditioned in fp32 (log of A-weighted magnitudes near the 1e-4
clamp): the reference's own fp32 gradient sits 2e-3 (relative L2) from the float64 result on the
golden signal (tests/test_oracle_golden.py), so gradient parity is measured against the float64
oracle and must be no further from it than 2x the reference's own distance.
"""
import pytest
import torch

import seeded
import stft_oracle
from golden_util import load_golden, rel_err

CFG = seeded.STFT_CFG
SR = 44100


def _signals(device):
    reals = torch.from_numpy(seeded.seeded_array((2, 2, 6000), 500, scale=0.1)).to(device)
    decoded = (reals + torch.from_numpy(seeded.seeded_array((2, 2, 6000), 501, scale=0.01)).to(device))
    return reals, decoded


def _values(device):
    from stable_audio_tools_amd import auraloss as al
    g = load_golden("mrstft")
    reals, decoded = _signals(device)
    mx, my = reals[:, 0:1].contiguous(), decoded[:, 0:1].contiguous()
    for n, h, w in zip(CFG["fft_sizes"], CFG["hop_sizes"], CFG["win_lengths"]):
        assert rel_err(al.STFTLoss(n, h, w)(mx, my), g[f"stft_plain_{n}"]) < 1e-3, n
        assert rel_err(al.STFTLoss(n, h, w, perceptual_weighting=True, sample_rate=SR).to(device)(mx, my), g[f"stft_aw_{n}"]) < 1e-3, n
    mr = al.MultiResolutionSTFTLoss(sample_rate=SR, **CFG).to(device)
    sd = al.SumAndDifferenceSTFTLoss(sample_rate=SR, **CFG).to(device)
    assert rel_err(mr(mx, my), g["loss_mono"]) < 1e-3
    assert rel_err(sd(reals, decoded), g["loss_sd"]) < 1e-3
    assert rel_err(mr(reals[:, 0:1].contiguous(), decoded[:, 0:1].contiguous()), g["loss_left"]) < 1e-3
    assert rel_err(mr(reals[:, 1:2].contiguous(), decoded[:, 1:2].contiguous()), g["loss_right"]) < 1e-3
    fused = al.AutoencoderSpectralLoss(SR, **CFG).to(device)
    assert rel_err(fused(reals, decoded), g["total"]) < 1e-3


def _gradients(device):
    from stable_audio_tools_amd import auraloss as al
    g = load_golden("mrstft")
    reals, decoded = _signals(device)
    d = decoded.clone().requires_grad_(True)
    fused = al.AutoencoderSpectralLoss(SR, **CFG).to(device)
    (gr,) = torch.autograd.grad(fused(reals, d), d)
    # three separate reference-style modules must give the same gradient as the fused form
    sd = al.SumAndDifferenceSTFTLoss(sample_rate=SR, **CFG).to(device)
    mr = al.MultiResolutionSTFTLoss(sample_rate=SR, **CFG).to(device)
    d2 = decoded.clone().requires_grad_(True)
    tot = sd(reals, d2) + 0.5 * mr(reals[:, 0:1], d2[:, 0:1]) + 0.5 * mr(reals[:, 1:2], d2[:, 1:2])
    (gr2,) = torch.autograd.grad(tot, d2)
    assert float((gr - gr2).norm() / gr.norm()) < 2e-3
    # float64 truth from the oracle
    r64 = reals.detach().cpu().double()
    d64 = decoded.detach().cpu().double().requires_grad_(True)
    (g64,) = torch.autograd.grad(stft_oracle.autoencoder_spectral_loss(r64, d64, CFG, SR), d64)
    ref = torch.from_numpy(g["grad_decoded"]).double()
    floor_ref = float((ref - g64).norm() / g64.norm())
    err = float((gr.detach().cpu().double() - g64).norm() / g64.norm())
    assert err < max(2 * floor_ref, 5e-3), (err, floor_ref)
    # swapped order: gradient w.r.t. the FIRST argument
    x = decoded.clone().requires_grad_(True)
    (gx,) = torch.autograd.grad(sd(x, reals), x)
    x64 = decoded.detach().cpu().double().requires_grad_(True)
    (gx64,) = torch.autograd.grad(stft_oracle.sum_and_difference_loss(
        x64, r64, CFG["fft_sizes"], CFG["hop_sizes"], CFG["win_lengths"], stft_oracle.aweighting_fir_taps(SR)), x64)
    assert float((gx.detach().cpu().double() - gx64).norm() / gx64.norm()) < 1e-2


def test_fir_and_adjoint_simulator(emu):
    taps = stft_oracle.aweighting_fir_taps(SR)
    x = torch.from_numpy(seeded.seeded_array((3, 2500), 7))
    y = emu.fir(x, taps)
    assert rel_err(y, stft_oracle.fir_filter(x, taps)) < 1e-5
    # <FIR x, z> == <x, FIR^T z>
    z = torch.from_numpy(seeded.seeded_array((3, 2500), 8))
    lhs = float((y.double() * z.double()).sum())
    rhs = float((x.double() * emu.fir(z, taps, adjoint=True).double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * abs(lhs)


def test_stft_loss_values_simulator(emu_modules):
    _values("cpu")


def test_stft_loss_gradients_simulator(emu_modules):
    _gradients("cpu")


def _views_consistency(ops, device):
    """The kernels transform each CHANNEL once and form the views from the channels' bins (csrc/stft.hip, round 6): six views in one call
    (two forward workgroup chunks of four) must give the sums of six one-view calls, and the backward with all views at once must equal the
    sum of the one-view backwards (linearity of the gradient in the per-view coefficients); mono input takes the single-channel path."""
    x = torch.from_numpy(seeded.seeded_array((2, 2, 3000), 510, scale=0.1)).to(device)
    y = x + torch.from_numpy(seeded.seeded_array((2, 2, 3000), 511, scale=0.01)).to(device)
    views = torch.tensor([[1.0, 1.0], [1.0, -1.0], [1.0, 0.0], [0.0, 1.0], [0.5, 0.25], [-0.3, 0.9]], device=device)
    for n, h in ((512, 128), (64, 16), (1024, 256)):
        s_all = ops.stft_sums(x, y, views, n, h)
        s_one = torch.cat([ops.stft_sums(x, y, views[i:i + 1].contiguous(), n, h) for i in range(6)], dim=1)
        assert rel_err(s_all, s_one) < 1e-5, n
        coef = torch.from_numpy(seeded.seeded_array((2, 6, 3), 512 + n, scale=1e-2)).abs().to(device)
        g_all = torch.zeros(4, 2, 2, 3000, device=device)
        ops.stft_backward(x, y, views, coef, g_all, n, h)
        g_sum = torch.zeros(2, 2, 3000, device=device)
        for i in range(6):
            g = torch.zeros(4, 2, 2, 3000, device=device)
            ops.stft_backward(x, y, views[i:i + 1].contiguous(), coef[:, i:i + 1].contiguous(), g, n, h)
            g_sum += g.sum(0)
        assert rel_err(g_all.sum(0), g_sum) < 1e-4, n
    # mono: one channel, one view
    xm, ym = x[:, :1].contiguous(), y[:, :1].contiguous()
    one = torch.tensor([[1.0, 0.0]], device=device)
    assert rel_err(ops.stft_sums(xm, ym, one, 256, 64), ops.stft_sums(x, y, torch.tensor([[1.0, 0.0]], device=device), 256, 64)) < 1e-5


def test_stft_views_from_channel_spectra_simulator(emu):
    _views_consistency(emu, "cpu")


@pytest.mark.gpu
def test_stft_views_from_channel_spectra_gpu(hip):
    _views_consistency(hip, "cuda")


def test_backward_refuses_hops_outside_the_two_plane_invariant(emu):
    """csrc/stft.hip's write-out lets a sample be touched by two NEIGHBOURING workgroups only: n_fft <= (frames per workgroup + 1) * hop.
    A smaller hop used to be accepted and would have produced a wrong gradient; it is an error now (the forward has no such limit)."""
    x = torch.from_numpy(seeded.seeded_array((1, 1, 4000), 520, scale=0.1))
    views = torch.tensor([[1.0, 0.0]])
    emu.stft_sums(x, x, views, 512, 32)
    with pytest.raises(RuntimeError, match="hop too small"):
        emu.stft_backward(x, x, views, torch.ones(1, 1, 3), torch.zeros(4, 1, 1, 4000), 512, 32)
    emu.stft_backward(x, x, views, torch.ones(1, 1, 3), torch.zeros(4, 1, 1, 4000), 512, 128)     # hop = n / 4: always served


def test_unsupported_configurations_raise():
    from stable_audio_tools_amd import auraloss as al
    with pytest.raises(NotImplementedError):
        al.STFTLoss(1024, 256, 600)
    with pytest.raises(NotImplementedError):
        al.MultiResolutionSTFTLoss(w_lin_mag=1.0, fft_sizes=[64], hop_sizes=[16], win_lengths=[64])
    with pytest.raises(ValueError):
        al.STFTLoss(64, 16, 64, perceptual_weighting=True)


@pytest.mark.gpu
def test_stft_loss_values_gpu(hip):
    _values("cuda")


@pytest.mark.gpu
def test_stft_loss_gradients_gpu(hip):
    _gradients("cuda")


@pytest.mark.gpu
def test_stft_linearity_property_full_size_gpu(hip):
    """Size-independent property at BASELINE.json's full length (T = 2097152): the sums are homogeneous —
    scaling both signals by a scales S1, S2 by a^2 and leaves S3 unchanged."""
    t = 2097152
    x = torch.randn(1, 2, t, device="cuda") * 0.1
    y = x + 0.01 * torch.randn_like(x)
    views = torch.tensor([[1.0, 1.0], [1.0, -1.0], [1.0, 0.0], [0.0, 1.0]], device="cuda")
    for n, h in ((2048, 512), (1024, 256), (32, 8)):      # one resolution per kernel instance (LDS footprints 2048 / 1024 / 512)
        s = hip.stft_sums(x, y, views, n, h)
        s2 = hip.stft_sums(2 * x, 2 * y, views, n, h)
        assert rel_err(s2[..., 0], 4 * s[..., 0]) < 1e-4
        assert rel_err(s2[..., 1], 4 * s[..., 1]) < 1e-4
        assert rel_err(s2[..., 2], s[..., 2]) < 1e-3


def _deterministic(device):
    """The loss gradient is bit-reproducible: csrc/stft.hip's backward uses no atomics (LDS overlap-add gathered per sample,
    global write-out as four write-once planes summed in a fixed order)."""
    from stable_audio_tools_amd.auraloss import AutoencoderSpectralLoss
    loss = AutoencoderSpectralLoss(44100, weight=1.0, **seeded.STFT_CFG).to(device)
    reals = torch.from_numpy(seeded.seeded_array((2, 2, 6000), 500, scale=0.1)).to(device)
    dec0 = (reals + torch.from_numpy(seeded.seeded_array((2, 2, 6000), 501, scale=0.01)).to(device))
    grads = []
    for _ in range(3):
        d = dec0.clone().requires_grad_(True)
        (g,) = torch.autograd.grad(loss(reals, d), d)
        grads.append(g)
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])


def test_stft_gradient_is_bit_reproducible_simulator(emu_modules):
    _deterministic("cpu")


@pytest.mark.gpu
def test_stft_gradient_is_bit_reproducible_gpu(hip):
    _deterministic("cuda")
