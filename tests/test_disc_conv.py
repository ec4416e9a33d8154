"""csrc/disc_conv.hip — the discriminator's Conv2d layers (models/encodec.py:37-106) on the pitched-rows layout: planes pass,
weight packing, conv (forward and data-gradient) and weight-gradient kernels against torch's conv2d in float64.  The simulator runs
the same kernel sources on the host (tests/emu); the GPU tests run the gfx950 library on larger shapes."""
import pytest
import torch
import torch.nn.functional as F

from golden_util import rel_err


def _unpitch(ops, y, frames, w):
    P = ops.disc_geom(frames, w)[0]
    return y.view(y.shape[0], y.shape[1], frames, P)[..., 4:4 + w]


def _pads_zero(ops, y, frames, w):
    P = ops.disc_geom(frames, w)[0]
    v = y.view(y.shape[0], y.shape[1], frames, P)
    return float(v[..., :4].abs().max()) == 0.0 and float(v[..., 4 + w:].abs().max()) == 0.0


def _case(ops, device, b, cin, cout, frames, w, kh, kw, dil, slope, seed=0):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(b, cin, frames, w, generator=gen).to(device)
    wt = (torch.randn(cout, cin, kh, kw, generator=gen) * 0.2).to(device)
    bias = torch.randn(cout, generator=gen).to(device)
    pad = (dil * (kh - 1) // 2, (kw - 1) // 2)
    xd = x.double().cpu().requires_grad_(True)
    wd = wt.double().cpu().requires_grad_(True)
    bd = bias.double().cpu().requires_grad_(True)
    pre = F.conv2d(xd, wd, bd, dilation=(dil, 1), padding=pad)
    ref = F.leaky_relu(pre, slope)
    gy = torch.randn(ref.shape, generator=gen).double()

    # forward: planes of the input, conv with emission, emitted planes == planes of the output
    xs, xp = ops.disc_planes(x, frames, w, want_dst=True, slot=0)
    assert _pads_zero(ops, xs, frames, w) and rel_err(_unpitch(ops, xs, frames, w), x) == 0.0
    y, em = ops.disc_conv(xp, ops.disc_pack(wt, 0), bias, b, cin, cout, frames, w, kh, kw, dil, slope, emit_slot=1)
    assert _pads_zero(ops, y, frames, w)
    assert rel_err(_unpitch(ops, y, frames, w), ref.detach().float()) < 2e-5
    _, yp = ops.disc_planes(y, frames, w, slot=0)
    assert torch.equal(yp[0], em[0]) and torch.equal(yp[1], em[1])

    # backward: dL/d(pre-activation) = gy * LeakyReLU'(y) (planes + pitched fp32), data-gradient, weight-gradient
    P, L = ops.disc_geom(frames, w)[:2]
    gyp = torch.zeros(b, cout, frames, P, device=device)
    gyp[..., 4:4 + w] = gy.float().to(device)
    gyp[..., :4] = 7.0                                                 # garbage at pad positions must be masked out
    dpre, dpl = ops.disc_planes(gyp.view(b, cout, L), frames, w, out=y, slope=slope, want_dst=True, slot=0)
    # (LeakyReLU' from the NATIVE output's sign: pre-activations within rounding of zero may land on the other side of it; the
    # reference gradients below are those of the same dpre)
    dpre_ref = gy * torch.where(_unpitch(ops, y, frames, w).cpu().double() > 0, 1.0, slope)
    gx, gw, gb = torch.autograd.grad(pre, [xd, wd, bd], dpre_ref)
    assert _pads_zero(ops, dpre, frames, w)
    assert rel_err(_unpitch(ops, dpre, frames, w), dpre_ref.float()) < 1e-6
    dx, _ = ops.disc_conv(dpl, ops.disc_pack(wt, 1), None, b, cout, cin, frames, w, kh, kw, dil, 1.0)
    assert rel_err(_unpitch(ops, dx, frames, w), gx.float()) < 2e-5
    dw = ops.disc_wgrad(dpre, xs, frames, w, kh, kw, dil)
    assert tuple(dw.shape) == tuple(wt.shape)
    assert rel_err(dw, gw.float()) < 2e-5
    assert rel_err(ops.rowsum(dpre), gb.float()) < 2e-5


SIM_CASES = [
    # b, cin, cout, frames, w, kh, kw, dil, slope
    (2, 4, 8, 5, 19, 3, 9, 1, 0.2),        # the first layer's shape (4 spectrogram channels), two batch items
    (1, 16, 12, 9, 33, 3, 9, 2, 0.2),      # dilated frame taps, two 8-channel groups, ragged Cout
    (1, 8, 1, 6, 21, 3, 3, 1, 1.0),        # conv_post: one output channel, 3 x 3, no activation
    (1, 24, 70, 4, 150, 3, 9, 4, 0.2),     # two co tiles, three position tiles (L = 632), dilation 4
    (1, 2, 8, 6, 17, 3, 9, 1, 0.2),        # mono input: two spectrogram channels
]


@pytest.mark.parametrize("case", SIM_CASES)
def test_disc_conv_simulator(emu, case):
    _case(emu, "cpu", *case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", SIM_CASES + [(2, 64, 64, 37, 257, 3, 9, 2, 0.2), (1, 64, 64, 30, 1025, 3, 3, 1, 0.2),
                                              (1, 4, 64, 64, 65, 3, 9, 1, 0.2), (2, 64, 1, 21, 129, 3, 3, 1, 1.0)])
def test_disc_conv_gpu(hip, case):
    _case(hip, "cuda", *case)
