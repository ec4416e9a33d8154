"""Kernel-level parity of the conv building blocks (forward + all gradients) against torch fp32 conv ops — the
reference's own building blocks (autoencoders.py:58-83, :245-247, :266-268 use nn.Conv1d / nn.ConvTranspose1d).
CPU leg: the kernel sources under the host simulator (tests/emu), shapes chosen to walk every K-chunk pipeline
length (1..5 chunks: prologue, steady state, tail) of every bf16x3 plan, partial chunks and partial tiles.
GPU leg: the same cases through the gfx950 library."""
import math

import pytest
import torch
import torch.nn.functional as F

from stable_audio_tools_amd import functional as Fn

TOL = 2e-4   # relative to the reference tensor's max-abs; bf16x3 products are ~2^-16 accurate


def snake(x, la, lb):
    a = la.exp()[None, :, None]
    b = lb.exp()[None, :, None]
    return x + (1.0 / (b + 1e-9)) * torch.sin(x * a) ** 2


def _leaf(gen, dev, *shape, s=1.0):
    return (torch.randn(*shape, generator=gen) * s).to(dev).requires_grad_(True)


FLOOR = [1e-3]   # tensors whose max-abs is below this are compared against it (tiny, cancellation-dominated gradients)


def _compare(outs, refs, inputs, gen, tol=TOL):
    gy = [torch.randn(r.shape, generator=gen).to(r.device) for r in refs]
    g1 = torch.autograd.grad(outs, inputs, gy, allow_unused=True)
    g2 = torch.autograd.grad(refs, inputs, gy, allow_unused=True)
    for a, b in list(zip(outs, refs)) + [(a, b) for a, b in zip(g1, g2) if b is not None]:
        err = (a - b).abs().max().item()
        assert err <= tol * max(b.abs().max().item(), FLOOR[0]), (err, b.abs().max().item())


# (B, Cin, Cout, T, K, dil): k7 chunks of 8 channels, k1 chunks of 32, k3 chunks of 8
S1_CASES = [(1, 8, 8, 300, 7, 1), (1, 16, 70, 280, 7, 3), (2, 24, 8, 200, 7, 9), (1, 32, 130, 150, 7, 9), (1, 40, 6, 520, 7, 3),
            (1, 6, 40, 150, 7, 9), (1, 32, 8, 140, 1, 1), (1, 64, 130, 200, 1, 1), (1, 96, 8, 130, 1, 1), (2, 160, 20, 100, 1, 1),
            (1, 70, 20, 150, 1, 1), (1, 12, 4, 100, 3, 1), (1, 9, 5, 77, 2, 1), (1, 20, 5, 90, 5, 2),
            (1, 64, 40, 200, 7, 3), (1, 72, 130, 150, 7, 9), (2, 128, 8, 140, 7, 1),
            (1, 64, 130, 152, 7, 1)]   # >= 64 in-channels: the pipelined wgrad kernel (two co tiles)
DOWN_CASES = [(1, 8, 16, 256, 2), (2, 12, 20, 333, 4), (1, 6, 130, 1100, 8), (1, 40, 6, 300, 2), (1, 8, 8, 520, 4),
              # output lengths that are multiples of 4: the vector-staged weight-gradient path (round 4), S = 2 / 4 / 8, partial and second tiles
              (2, 70, 6, 264, 2), (1, 40, 24, 1040, 4), (1, 20, 130, 2112, 8),
              # three and more time tiles: INTERIOR tiles take the vector staging loads (forward) / the vector depth-to-space epilogue (data-gradient)
              (1, 12, 10, 800, 2), (1, 5, 9, 1560, 4)]
UP_CASES = [(1, 16, 8, 40, 2), (2, 12, 20, 33, 4), (1, 6, 130, 70, 8), (1, 70, 6, 150, 2), (1, 48, 8, 131, 4),
            (2, 6, 70, 260, 2), (1, 24, 40, 132, 4), (1, 130, 12, 68, 8),
            (1, 8, 12, 300, 4), (1, 10, 12, 300, 8), (1, 7, 5, 390, 2)]      # interior tiles (see DOWN_CASES), odd channel counts


def _run_s1(ops, dev, case, use_x3):
    B, Cin, Cout, T, K, dil = case
    gen = torch.Generator().manual_seed(hash(case) % 2 ** 31)
    ops.use_bf16x3 = use_x3
    try:
        x = _leaf(gen, dev, B, Cin, T)
        la, lb = _leaf(gen, dev, Cin, s=.3), _leaf(gen, dev, Cin, s=.3)
        w, bias = _leaf(gen, dev, Cout, Cin, K, s=.2), _leaf(gen, dev, Cout)
        pad = dil * (K - 1) // 2
        tout = T + 2 * pad - dil * (K - 1)
        res = _leaf(gen, dev, B, Cout, tout)
        y1 = Fn.SnakeConv1dFn.apply(x, la, lb, w, bias, res, 1, dil, pad, False, ops)
        y2 = F.conv1d(snake(x, la, lb), w, bias, padding=pad, dilation=dil) + res
        _compare([y1], [y2], [x, la, lb, w, bias, res], gen)
    finally:
        ops.use_bf16x3 = True


def _run_down(ops, dev, case, use_x3):
    B, Cin, Cout, T, S = case
    gen = torch.Generator().manual_seed(hash(case) % 2 ** 31)
    ops.use_bf16x3 = use_x3
    try:
        K, pad = 2 * S, math.ceil(S / 2)
        x = _leaf(gen, dev, B, Cin, T)
        la, lb = _leaf(gen, dev, Cin, s=.3), _leaf(gen, dev, Cin, s=.3)
        w, bias = _leaf(gen, dev, Cout, Cin, K, s=.2), _leaf(gen, dev, Cout)
        y1 = Fn.SnakeConv1dFn.apply(x, la, lb, w, bias, None, S, 1, pad, False, ops)
        y2 = F.conv1d(snake(x, la, lb), w, bias, stride=S, padding=pad)
        _compare([y1], [y2], [x, la, lb, w, bias], gen)
    finally:
        ops.use_bf16x3 = True


def _run_up(ops, dev, case, use_x3):
    B, Cin, Cout, T, S = case
    gen = torch.Generator().manual_seed(hash(case) % 2 ** 31)
    ops.use_bf16x3 = use_x3
    try:
        K, pad = 2 * S, math.ceil(S / 2)
        x = _leaf(gen, dev, B, Cin, T)
        la, lb = _leaf(gen, dev, Cin, s=.3), _leaf(gen, dev, Cin, s=.3)
        w, bias = _leaf(gen, dev, Cin, Cout, K, s=.2), _leaf(gen, dev, Cout)
        y1 = Fn.SnakeConvTr1dFn.apply(x, la, lb, w, bias, S, pad, ops)
        y2 = F.conv_transpose1d(snake(x, la, lb), w, bias, stride=S, padding=pad)
        _compare([y1], [y2], [x, la, lb, w, bias], gen)
    finally:
        ops.use_bf16x3 = True


def _run_nosnake_tanh(ops, dev):
    gen = torch.Generator().manual_seed(7)
    x = _leaf(gen, dev, 1, 16, 190)
    w = _leaf(gen, dev, 2, 16, 7, s=.2)
    y1 = Fn.SnakeConv1dFn.apply(x, None, None, w, None, None, 1, 1, 3, True, ops)
    y2 = torch.tanh(F.conv1d(x, w, None, padding=3))
    _compare([y1], [y2], [x, w], gen)


# the two-channel ends of the stack (csrc/edge_conv.hip, round 6): encoder-first-conv form (Cin <= 2, no activation, bias), decoder-last-conv
# form (SnakeBeta -> Cout <= 2, with and without bias / tanh), every gradient (the decoder form's data-gradient is the narrow-input
# kernel with the dsnake epilogue on the flipped weight; both weight-gradient forms; the bias gradient fused into the narrow-input one);
# lengths around the 1024-step tile and not multiples of 4 (the scalar edge paths); the same cases on the matrix kernels
# (ops.edge_convs = False) — two implementations, one torch reference
EDGE_CASES = [  # (B, Cin, Cout, T, K, snake, bias, tanh)
    (1, 2, 24, 1500, 7, False, True, False), (2, 2, 40, 1024, 7, False, True, False), (1, 1, 16, 333, 7, False, True, False),
    (1, 24, 2, 1500, 7, True, False, False), (2, 40, 2, 2050, 7, True, True, True), (1, 16, 1, 1027, 7, True, False, False),
    (1, 2, 8, 4100, 5, False, True, False), (1, 12, 2, 700, 3, True, False, False), (1, 2, 130, 2048, 7, False, True, False),
]


def _compare64(native, ref, inputs, gen, tol=TOL):
    """native(*inputs) on the device against ref(*float64 CPU copies): outputs and every gradient.  The truth is float64: a same-device
    torch conv is itself an fp32 kernel with its own summation order (MIOpen on the GPU: 2e-4 off on a one-channel weight-norm
    magnitude summed over 1027 x 16 x 7 terms — as far from float64 as the kernels under test)."""
    outs = native(*inputs)
    in64 = [t.detach().double().cpu().requires_grad_(True) for t in inputs]
    refs = ref(*in64)
    gy = [torch.randn(r.shape, generator=gen, dtype=torch.float64) for r in refs]
    g1 = torch.autograd.grad(outs, inputs, [g.float().to(o.device) for g, o in zip(gy, outs)], allow_unused=True)
    g2 = torch.autograd.grad(refs, in64, gy, allow_unused=True)
    for a, b in list(zip(outs, refs)) + [(a, b) for a, b in zip(g1, g2) if b is not None]:
        err = (a.detach().double().cpu() - b.detach()).abs().max().item()
        assert err <= tol * max(b.abs().max().item(), FLOOR[0]), (err, b.abs().max().item())


def _run_edge(ops, dev, case, edge):
    B, Cin, Cout, T, K, use_snake, use_bias, tanh_out = case
    gen = torch.Generator().manual_seed(hash(case) % 2 ** 31)
    saved = ops.edge_convs
    ops.edge_convs = edge
    try:
        pad = (K - 1) // 2
        assert ops.edge_ok(Cin, Cout, K, 1, 1, pad) == edge
        x = _leaf(gen, dev, B, Cin, T)
        la, lb = (_leaf(gen, dev, Cin, s=.3), _leaf(gen, dev, Cin, s=.3)) if use_snake else (None, None)
        w = _leaf(gen, dev, Cout, Cin, K, s=.2)
        bias = _leaf(gen, dev, Cout) if use_bias else None
        def unpack(ts):
            it = iter(ts)
            x_ = next(it)
            la_, lb_ = (next(it), next(it)) if use_snake else (None, None)
            return x_, la_, lb_, it

        def ref_conv(x_, la_, lb_, w_, b_):
            y = F.conv1d(snake(x_, la_, lb_) if use_snake else x_, w_, b_, padding=pad)
            return [torch.tanh(y) if tanh_out else y]

        def native_plain(*ts):
            x_, la_, lb_, it = unpack(ts)
            w_ = next(it)
            return [Fn.SnakeConv1dFn.apply(x_, la_, lb_, w_, next(it, None), None, 1, 1, pad, tanh_out, ops)]

        def ref_plain(*ts):
            x_, la_, lb_, it = unpack(ts)
            w_ = next(it)
            return ref_conv(x_, la_, lb_, w_, next(it, None))
        _compare64(native_plain, ref_plain, [t for t in (x, la, lb, w, bias) if t is not None], gen)
        # weight-normed form (what the model runs): the unit folds (v, g) itself and takes the edge kernels' slabs to sat_wn_grad_splits
        v, g = _leaf(gen, dev, Cout, Cin, K, s=.2), _leaf(gen, dev, Cout, 1, 1)

        def native_wn(*ts):
            x_, la_, lb_, it = unpack(ts)
            v_, g_ = next(it), next(it)
            return [Fn.SnakeConv1dFn.apply(x_, la_, lb_, v_, next(it, None), None, 1, 1, pad, tanh_out, ops, None, None, g_)]

        def ref_wn(*ts):
            x_, la_, lb_, it = unpack(ts)
            v_, g_ = next(it), next(it)
            return ref_conv(x_, la_, lb_, g_ * v_ / v_.flatten(1).norm(dim=1).view(-1, 1, 1), next(it, None))
        _compare64(native_wn, ref_wn, [t for t in (x, la, lb, v, g, bias) if t is not None], gen)
    finally:
        ops.edge_convs = saved


def _edge_emit_case(ops, dev):
    """Plane emission of the narrow-input edge conv (the encoder's first conv writes the first ResidualUnit's k7 planes): hi + lo of the
    emitted planes == snake(y) to 2^-16, rows around the sequence zero, channels past Cout zero (Cout = 20: a partial group of 8)."""
    gen = torch.Generator().manual_seed(11)
    B, Cin, Cout, T = 2, 2, 20, 1300
    x = (torch.randn(B, Cin, T, generator=gen)).to(dev)
    w = (torch.randn(Cout, Cin, 7, generator=gen) * .3).to(dev)
    bias = torch.randn(Cout, generator=gen).to(dev)
    la, lb = (torch.randn(Cout, generator=gen) * .3).to(dev), (torch.randn(Cout, generator=gen) * .3).to(dev)
    for esnake in ((la, lb), None):
        y = ops.edge_conv(x, w, 3, bias=bias, emit={"snake": esnake})
        ref = F.conv1d(x.cpu(), w.cpu(), bias.cpu(), padding=3)
        assert (y.cpu() - ref).abs().max() <= 1e-5 * ref.abs().max()
        em = ops._take_emitted(y, esnake)
        assert em is not None
        rows = em["rows"]
        hi = em["hi"].view(torch.bfloat16).float().view(B, 3, rows, 8).cpu()
        lo = em["lo"].view(torch.bfloat16).float().view(B, 3, rows, 8).cpu()
        got = (hi + lo).permute(0, 1, 3, 2).reshape(B, 24, rows)           # (B, channel, plane row)
        want = snake(ref, la.cpu(), lb.cpu()) if esnake is not None else ref
        assert (got[:, :Cout, 32:32 + T] - want).abs().max() <= 2.0 ** -15 * want.abs().max()
        assert got[:, Cout:, :].abs().max() == 0 and got[:, :, :32].abs().max() == 0 and got[:, :, 32 + T:].abs().max() == 0


def test_edge_conv_emission_sim(emu):
    _edge_emit_case(emu, "cpu")


@pytest.mark.gpu
def test_edge_conv_emission_gpu(hip):
    _edge_emit_case(hip, "cuda")


@pytest.mark.parametrize("case", EDGE_CASES)
def test_edge_conv_sim(emu, case):
    _run_edge(emu, "cpu", case, True)
    if case[3] <= 1500:
        _run_edge(emu, "cpu", case, False)


@pytest.mark.gpu
def test_edge_conv_gpu(hip):
    for case in EDGE_CASES + [(1, 2, 128, 262144, 7, False, True, False), (1, 128, 2, 262144, 7, True, False, False)]:
        _run_edge(hip, "cuda", case, True)
        _run_edge(hip, "cuda", case, False)


@pytest.mark.parametrize("case", S1_CASES)
def test_conv_stride1_sim(emu, case):
    _run_s1(emu, "cpu", case, True)


@pytest.mark.parametrize("case", DOWN_CASES)
def test_conv_down_sim(emu, case):
    _run_down(emu, "cpu", case, True)


@pytest.mark.parametrize("case", UP_CASES)
def test_conv_up_sim(emu, case):
    _run_up(emu, "cpu", case, True)


def _run_planes(ops, dev, cases):
    """The k = 7 convs fed from pre-split activation planes (conv1d_planes.h: sat_conv1d_k7_planes; conv1d_bf16x3_k7q.h:
    sat_conv1d_bf16x3_planesq) — the path every C >= 64 level takes — forced on for every channel count: forward and all gradients vs
    torch, and outputs equal to the direct kernel's up to the accumulation order (16-channel chunks, one tap per k-step)."""
    keepq = (ops.k7q, ops.k7q_min_cin, ops.k7q_min_cout, ops.k7q_persist)
    try:
        for case in cases:
            ops.k7q, ops.k7q_min_cin, ops.k7q_min_cout = True, 1, 1
            # the PERSISTENT launch (round 6: a workgroup walks several tiles, the next tile's first chunk requested before the epilogue) forced on
            # these small shapes (three tiles per workgroup: first / middle / last-and-partial), then the one-workgroup-per-tile launch
            ops.k7q_persist = "force"
            _run_s1(ops, dev, case, True)
            ops.k7q_persist = False
            _run_s1(ops, dev, case, True)               # autograd units through the planes kernel
            B, Cin, Cout, T, K, dil = case
            gen = torch.Generator().manual_seed(7)
            x = torch.randn(B, Cin, T, generator=gen).to(dev)
            w = (torch.randn(Cout, Cin, K, generator=gen) * .2).to(dev)
            la, lb = (torch.randn(Cin, generator=gen) * .3).to(dev), (torch.randn(Cin, generator=gen) * .3).to(dev)
            x2 = torch.randn(B, Cout, T, generator=gen).to(dev)
            a2, b2 = (torch.randn(Cout, generator=gen) * .3).to(dev), (torch.randn(Cout, generator=gen) * .3).to(dev)
            wp = ops.pack_bf16x3(w, 0, 1)
            pad = dil * (K - 1) // 2
            direct = (ops.conv1d_bf16x3(x, wp, Cout, K, 1, dil, pad, snake=(la, lb)),
                      *ops.conv1d_bf16x3(x, wp, Cout, K, 1, dil, pad, dsnake=(x2, a2, b2)))
            if 5 <= K <= 7:
                wq = ops.pack_bf16x3(w, 0, 1, q=True)
                outq = (ops.conv1d_bf16x3(x, wq, Cout, K, 1, dil, pad, snake=(la, lb)),
                        *ops.conv1d_bf16x3(x, wq, Cout, K, 1, dil, pad, dsnake=(x2, a2, b2)))
                for a, b in zip(outq, direct):
                    assert (a - b).abs().max().item() <= 2e-5 * max(b.abs().max().item(), 1e-3), (case, (a - b).abs().max().item())
                ops.k7q_persist = "force"
                outp = (ops.conv1d_bf16x3(x, wq, Cout, K, 1, dil, pad, snake=(la, lb)),
                        *ops.conv1d_bf16x3(x, wq, Cout, K, 1, dil, pad, dsnake=(x2, a2, b2)))
                for a, b in zip(outp, outq):
                    assert torch.equal(a, b), case        # same arithmetic in the same order: the persistent launch is bit-identical
                # the kernel variant that issues the next chunk's LDS-DMA inside the MFMA sections (round 6): same arithmetic, other schedule
                keepv = ops.k7q_dma_in_mfma
                try:
                    ops.k7q_dma_in_mfma = not keepv
                    for mode in ("force", False, True):
                        ops.k7q_persist = mode
                        outv = (ops.conv1d_bf16x3(x, wq, Cout, K, 1, dil, pad, snake=(la, lb)),
                                *ops.conv1d_bf16x3(x, wq, Cout, K, 1, dil, pad, dsnake=(x2, a2, b2)))
                        for a, b in zip(outv, outq):
                            assert torch.equal(a, b), (case, mode)
                finally:
                    ops.k7q_dma_in_mfma = keepv
    finally:
        ops.k7q, ops.k7q_min_cin, ops.k7q_min_cout, ops.k7q_persist = keepq


def test_conv_k7_planes_sim(emu):
    _run_planes(emu, "cpu", [c for c in S1_CASES if c[4] == 7][:6] + [(1, 20, 5, 90, 5, 2), (2, 16, 130, 517, 7, 3), (1, 16, 130, 1300, 7, 9)])


def test_conv_k7_wide_input_few_outputs_sim(emu):
    """The data-gradient of the decoder's first conv (2048 -> 64 channels, 1024 steps; here 520 -> 40): fewer output channels than
    k7q_min_cout still take the planes kernel when the input is wide (ops.k7q_wide_cin) — round 2's k7p kernel served this plan."""
    keep = emu.k7q_wide_cin
    try:
        emu.k7q_wide_cin = 512
        assert emu.k7q_applicable(520, 7, 1, 1, 3, 40) and not emu.k7q_applicable(128, 7, 1, 1, 3, 40)
        _run_s1(emu, "cpu", (1, 520, 40, 200, 7, 1), True)
    finally:
        emu.k7q_wide_cin = keep


@pytest.mark.gpu
def test_conv_k7_planes_gpu(hip):
    _run_planes(hip, "cuda", [c for c in S1_CASES if c[4] == 7] + [(1, 128, 128, 8192, 7, 9), (1, 1024, 1024, 512, 7, 3), (2, 512, 512, 1000, 7, 1),
                              (1, 128, 128, 300000, 7, 3), (1, 2048, 64, 1024, 7, 1)])


def _run_ru(ops, dev, cases):
    """The fused ResidualUnit forward (csrc/conv1d_bf16x3_k7q.h, FUSED: one launch) vs torch's conv1d chain: y, the kept intermediate
    through every gradient, with and without the plane emission for a following unit (whose k7 conv then consumes those planes)."""
    keep = (ops.k7q, ops.k7q_min_cin, ops.k7q_min_cout, ops.ru_fused, ops.k7_emit)
    calls = {"n": 0}
    orig = ops.lib.sat_residual_unit_fwd

    def counted(*a):
        calls["n"] += 1
        return orig(*a)
    ops.lib.sat_residual_unit_fwd = counted
    try:
        ops.k7q, ops.k7q_min_cin, ops.k7q_min_cout, ops.ru_fused, ops.k7_emit = True, 1, 1, True, True
        for (B, C, T, dil) in cases:
            gen = torch.Generator().manual_seed(C * 1000 + T + dil)
            x = _leaf(gen, dev, B, C, T)
            ps = [_leaf(gen, dev, C, s=.3) for _ in range(4)]                  # a1, b1, a2, b2
            w1, bias1 = _leaf(gen, dev, C, C, 7, s=.2 / math.sqrt(C / 8)), _leaf(gen, dev, C)
            w2, bias2 = _leaf(gen, dev, C, C, 1, s=.5 / math.sqrt(C / 8)), _leaf(gen, dev, C)
            na, nb = _leaf(gen, dev, C, s=.3), _leaf(gen, dev, C, s=.3)        # the next unit's first activation
            w3 = _leaf(gen, dev, C, C, 7, s=.2 / math.sqrt(C / 8))
            n0 = calls["n"]
            y1 = Fn.ResidualUnitFn.apply(x, ps[0], ps[1], w1, bias1, ps[2], ps[3], w2, bias2, dil, ops, False, None, (na, nb, 1), True)
            z1 = Fn.SnakeConv1dFn.apply(y1, na, nb, w3, None, None, 1, 1, 3, False, ops)     # consumes the emitted planes
            assert calls["n"] == n0 + 1
            h = F.conv1d(snake(x, ps[0], ps[1]), w1, bias1, padding=3 * dil, dilation=dil)
            y2 = x + F.conv1d(snake(h, ps[2], ps[3]), w2, bias2)
            z2 = F.conv1d(snake(y2, na, nb), w3, None, padding=3)
            _compare([y1, z1], [y2, z2], [x, *ps, w1, bias1, w2, bias2, na, nb, w3], gen)
            with torch.no_grad():                                               # inference form: the intermediate is never stored
                y3 = Fn.ResidualUnitFn.apply(x, ps[0], ps[1], w1, bias1, ps[2], ps[3], w2, bias2, dil, ops, False, None, None, "nokeep")
            assert calls["n"] == n0 + 2 and torch.equal(y3, y1.detach())
    finally:
        ops.k7q, ops.k7q_min_cin, ops.k7q_min_cout, ops.ru_fused, ops.k7_emit = keep
        ops.lib.sat_residual_unit_fwd = orig


def _ru_k1_bwd_case(ops, dev, B, C, T, seed):
    """csrc/ru_k1_bwd.hip (the whole backward of a unit's 1x1 conv in one pass over dy and h) against float64 autograd of
    conv1d(snake(h), W2) + b2, and its emitted planes against the planes pre-pass of the same dh."""
    gen = torch.Generator().manual_seed(seed)
    dy = torch.randn(B, C, T, generator=gen).to(dev)
    h = torch.randn(B, C, T, generator=gen).to(dev)
    w2 = (torch.randn(C, C, 1, generator=gen) * (.5 / math.sqrt(C / 8))).to(dev)
    a2, b2 = (torch.randn(C, generator=gen) * .3).to(dev), (torch.randn(C, generator=gen) * .3).to(dev)
    assert ops.ru_k1_bwd_ok(B, C, T)
    dh, da, db, dw, dbias2, dbias1 = ops.ru_k1_bwd(dy, h, w2, (a2, b2), emit=True)
    em = ops._take_emitted(dh, None)
    assert em is not None
    hd, wd, ad, bd = (t.detach().double().cpu().requires_grad_(True) for t in (h, w2, a2, b2))
    bias = torch.zeros(C, dtype=torch.float64, requires_grad=True)
    y = F.conv1d(snake(hd, ad, bd), wd, bias)
    y.backward(dy.double().cpu())
    for got, ref, name in ((dh, hd.grad, "dh"), (da, ad.grad, "dalpha"), (db, bd.grad, "dbeta"), (dw, wd.grad, "dW2"), (dbias2, bias.grad, "dbias2"),
                           (dbias1, hd.grad.sum(dim=(0, 2)), "dbias1")):
        err = (got.double().cpu() - ref).abs().max().item()
        assert err <= 2e-5 * max(ref.abs().max().item(), 1e-3) + 1e-6 * math.sqrt(B * T), (name, (B, C, T), err, ref.abs().max().item())
    # the emitted planes hold dh (hi + lo) at rows 32 + t of every 8-channel group
    rows = em["rows"]
    hi = em["hi"].view(B, C // 8, rows, 8)[:, :, 32:32 + T].permute(0, 1, 3, 2).reshape(B, C, T)
    lo = em["lo"].view(B, C // 8, rows, 8)[:, :, 32:32 + T].permute(0, 1, 3, 2).reshape(B, C, T)
    rec = (hi.view(torch.bfloat16).float() + lo.view(torch.bfloat16).float()).cpu()
    assert (rec - dh.cpu()).abs().max().item() <= 2.0 ** -15 * dh.abs().max().item()


def test_ru_k1_bwd_sim(emu):
    _ru_k1_bwd_case(emu, "cpu", 2, 128, 96, 41)          # two items, three tiles each
    assert not emu.ru_k1_bwd_ok(1, 256, 64) and not emu.ru_k1_bwd_ok(1, 128, 48)      # other widths / ragged lengths: the separate kernels


@pytest.mark.gpu
def test_ru_k1_bwd_gpu(hip):
    _ru_k1_bwd_case(hip, "cuda", 2, 128, 96, 41)
    _ru_k1_bwd_case(hip, "cuda", 1, 128, 65536, 43)      # 2048 tiles: eight per workgroup, every workgroup of the grid busy
    _ru_k1_bwd_case(hip, "cuda", 3, 128, 8224, 44)       # 771 tiles on 256 workgroups: ranges cross the batch boundaries, the last one is short


def test_residual_unit_fused_sim(emu):
    _run_ru(emu, "cpu", [(1, 16, 300, 1), (2, 24, 520, 3), (1, 72, 260, 9), (1, 128, 256, 3)])


@pytest.mark.gpu
def test_residual_unit_fused_gpu(hip):
    _run_ru(hip, "cuda", [(1, 16, 300, 1), (2, 24, 520, 3), (1, 72, 260, 9), (1, 128, 8192, 3), (2, 128, 70000, 9), (1, 128, 4096, 1)])


def test_conv_fp32_kernels_sim(emu):
    _run_s1(emu, "cpu", S1_CASES[1], False)
    _run_s1(emu, "cpu", S1_CASES[7], False)
    _run_down(emu, "cpu", DOWN_CASES[1], False)
    _run_up(emu, "cpu", UP_CASES[1], False)
    _run_nosnake_tanh(emu, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("use_x3", [True, False])
def test_conv_kernels_gpu(hip, use_x3):
    for case in S1_CASES:
        _run_s1(hip, "cuda", case, use_x3)
    for case in DOWN_CASES:
        _run_down(hip, "cuda", case, use_x3)
    for case in UP_CASES:
        _run_up(hip, "cuda", case, use_x3)
    _run_nosnake_tanh(hip, "cuda")


@pytest.mark.gpu
def test_conv_kernels_gpu_large(hip):
    """Full-width tiles and long K loops (the shapes the bench runs), against torch on the same device."""
    for case in [(1, 128, 128, 8192, 7, 9), (1, 256, 256, 4096, 1, 1), (1, 1024, 1024, 512, 7, 3)]:
        _run_s1(hip, "cuda", case, True)
    _run_down(hip, "cuda", (1, 128, 256, 8192, 2), True)
    _run_down(hip, "cuda", (1, 512, 1024, 4096, 8), True)
    _run_up(hip, "cuda", (1, 256, 128, 2048, 4), True)
    _run_up(hip, "cuda", (1, 1024, 512, 512, 8), True)


def _random_cases(seed, n):
    """Seeded random shapes around the tile edges: T below / at / just past one tile and not a multiple of 4, channel
    counts that are not multiples of the 8-channel K-chunks or the 32/64/128-wide tiles, batch > 1."""
    import random
    rnd = random.Random(seed)
    out = []
    for _ in range(n):
        kind = rnd.choice(["s1", "s1", "down", "up"])
        b = rnd.choice([1, 1, 2])
        cin, cout = rnd.choice([1, 2, 5, 8, 17, 33, 64, 70]), rnd.choice([1, 2, 7, 32, 40, 65, 129])
        if kind == "s1":
            k = rnd.choice([1, 1, 3, 7, 7, 7])
            dil = rnd.choice([1, 3, 9]) if k == 7 else 1
            t = rnd.choice([5 * dil + 9, 63, 127, 128, 130, 257, 300])
            out.append(("s1", (b, cin, cout, max(t, 3 * dil + 2), k, dil)))
        else:
            s = rnd.choice([2, 4, 8])
            t = rnd.choice([4 * s, 8 * s + 3, 130, 259]) if kind == "down" else rnd.choice([3, 17, 64, 130])
            out.append((kind, (b, cin, cout, t, s)))
    return out


@pytest.mark.parametrize("kind,case", _random_cases(20260921, 24))
def test_conv_random_shapes_sim(emu, kind, case):
    # one-channel / few-sample shapes produce gradients that are sums of O(1) products cancelling to ~1e-2: the bf16x3
    # products are accurate to ~2^-16 of the PRODUCTS, so the floor for these cases is 0.1
    FLOOR[0] = 0.1
    try:
        {"s1": _run_s1, "down": _run_down, "up": _run_up}[kind](emu, "cpu", case, True)
    finally:
        FLOOR[0] = 1e-3


def _wgrad7_bias_case(ops, dev):
    """conv_wgrad7_bf16x3(dy_rowsum=True): dW and the bias gradient, on a shape the pipelined kernel takes (row sums by sat_rowsum) and on
    one the four-wave kernel takes (row sums fused)."""
    gen = torch.Generator().manual_seed(5)
    for (m, n, t) in ((130, 64, 152), (70, 24, 152)):
        dy = torch.randn(2, m, t, generator=gen).to(dev)
        x = torch.randn(2, n, t, generator=gen).to(dev)
        ref_w = ops.conv_wgrad7_bf16x3(dy, x, 1, 3)
        dw, db = ops.conv_wgrad7_bf16x3(dy, x, 1, 3, dy_rowsum=True)
        assert torch.equal(dw, ref_w)
        assert (db.cpu() - dy.sum(dim=(0, 2)).cpu()).abs().max().item() <= 1e-4 * dy.abs().sum(dim=(0, 2)).max().item()


def test_wgrad7_bias_gradient_sim(emu):
    _wgrad7_bias_case(emu, "cpu")


@pytest.mark.gpu
def test_wgrad7_bias_gradient_gpu(hip):
    _wgrad7_bias_case(hip, "cuda")


def _wn_splits_case(ops, dev):
    """sat_wn_grad_splits (weight-norm gradient straight from a weight-gradient kernel's split slabs) against sat_reduce_splits + layout
    change + sat_wn_grad, and against autograd of g * v / ||v|| in float64: both slab layouts (torch order, tap-major), one and many
    slabs, a row longer than the LDS staging buffer (summed twice instead)."""
    from stable_audio_tools_amd.ops import WgradSlabs
    gen = torch.Generator().manual_seed(11)
    for (m, n, k, ns, tap_major, pad) in ((5, 12, 7, 3, True, 8), (130, 64, 7, 1, True, 0), (3, 9, 1, 6, False, 8), (40, 24, 16, 5, False, 8),
                                          (4, 6, 7, 9, True, 8), (6, 12, 7, 20, True, 7), (16, 32, 4, 11, False, 3), (128, 128, 7, 37, True, 0),
                                          (2, 2400, 7, 2, True, 8), (2, 2400, 7, 3, False, 8)):
        # (16-byte loads need aligned slabs: the odd paddings take the 4-byte path; 2400 x 7 > the LDS staging buffer: summed twice)
        v = torch.randn(m, n, k, generator=gen).to(dev)
        g = (torch.rand(m, 1, 1, generator=gen) + .5).to(dev)
        partial = torch.randn(ns, m * n * k + pad, generator=gen).to(dev)
        strides = (n, 1, m * n) if tap_major else (n * k, k, 1)
        slabs = WgradSlabs(partial, ns, (m, n, k), strides)
        summed = partial[:, :m * n * k].sum(0)
        dw = summed.view(k, m, n).permute(1, 2, 0).contiguous() if tap_major else summed.view(m, n, k)
        w, norm = ops.wn_fold(v, g.view(-1))
        bp = torch.randn(m, 1 + 37 * ns, generator=gen).to(dev)             # per-split sums of dy: the bias gradient rides along
        dv, dg, dbias = ops.wn_grad_splits(slabs, v, g.view(-1), norm, bias_partial=bp)
        assert (dbias.double().cpu() - bp.double().cpu().sum(1)).abs().max().item() <= 1e-5 * bp.abs().sum(1).max().item()
        dv1, dg1 = ops.wn_grad_splits(slabs, v, g.view(-1), norm)
        assert torch.equal(dv, dv1) and torch.equal(dg, dg1)
        dv0, dg0 = ops.wn_grad(v, g.view(-1), norm, dw)
        vd, gd = v.double().cpu().requires_grad_(True), g.double().cpu().requires_grad_(True)
        (gd * vd / vd.flatten(1).norm(dim=1).view(-1, 1, 1) * dw.double().cpu()).sum().backward()
        for got, ref in ((dv, vd.grad), (dg, gd.grad.view(-1)), (dv0, vd.grad), (dg0, gd.grad.view(-1))):
            assert (got.double().cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item(), (m, n, k, ns, tap_major)
    with pytest.raises(ValueError):
        ops.wn_grad_splits(slabs, v[:, :5], g.view(-1), norm)


def test_wn_grad_from_slabs_sim(emu):
    _wn_splits_case(emu, "cpu")


@pytest.mark.gpu
def test_wn_grad_from_slabs_gpu(hip):
    _wn_splits_case(hip, "cuda")


def test_wn_grad_from_slabs_random_shapes_sim(emu):
    """Randomised shapes through sat_wn_grad_splits on the simulator: degenerate dims (one row, one input channel, one tap, one slab),
    every combination of layout, slab alignment and z-part count the kernel's dispatch distinguishes, against the separate kernels."""
    import random
    from stable_audio_tools_amd.ops import WgradSlabs
    rnd = random.Random(5)
    gen = torch.Generator().manual_seed(5)
    for case in range(40):
        m, n = rnd.randint(1, 9), rnd.choice([1, 2, 3, 4, 8, 12, 17, 40, 64])
        k, ns = rnd.choice([1, 2, 3, 4, 7, 16]), rnd.choice([1, 2, 3, 5, 8, 12, 33])
        tap_major, pad = rnd.random() < 0.5, rnd.choice([0, 1, 4, 8])
        v = torch.randn(m, n, k, generator=gen)
        g = torch.rand(m, generator=gen) + .5
        partial = torch.randn(ns, m * n * k + pad, generator=gen)
        slabs = WgradSlabs(partial, ns, (m, n, k), (n, 1, m * n) if tap_major else (n * k, k, 1))
        summed = partial[:, :m * n * k].double().sum(0).float()
        dw = summed.view(k, m, n).permute(1, 2, 0).contiguous() if tap_major else summed.view(m, n, k)
        _, norm = emu.wn_fold(v, g)
        bp = torch.randn(m, rnd.choice([1, 3, 64, 300]), generator=gen)
        dv, dg, dbias = emu.wn_grad_splits(slabs, v, g, norm, bias_partial=bp)
        dv0, dg0 = emu.wn_grad(v, g, norm, dw)
        scale = max(dw.abs().max().item(), 1e-6)
        assert (dv - dv0).abs().max().item() <= 2e-5 * max(dv0.abs().max().item(), scale), (case, m, n, k, ns, tap_major, pad)
        assert (dg - dg0).abs().max().item() <= 2e-5 * max(dg0.abs().max().item(), scale), (case, m, n, k, ns, tap_major, pad)
        assert (dbias - bp.sum(1)).abs().max().item() <= 1e-5 * max(bp.abs().sum(1).max().item(), 1e-6)
        assert torch.equal(slabs.reduce(emu), dw) or (slabs.reduce(emu) - dw).abs().max().item() <= 1e-5 * scale
