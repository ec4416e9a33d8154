"""Kernel-level tests of csrc/gemm.hip (the DiT projections: transformer.py:263,308,356-364,481-507) against plain torch
fp32 matmuls of the same bf16-rounded operands: every epilogue, every shipped tile, ragged M / N / K (partial tiles, K tails
that are not a multiple of the 64-wide K-step), split-K, the fused head-split / rotary / plane-layout epilogue, and the
operand preparation kernels (cast, transpose, bf16x3 split).  CPU: the simulator; `-m gpu`: the gfx950 library, plus the
BASELINE shapes (M = 2050 tokens, d = 1536, FF 12288) at full size."""
import pytest
import torch

import dit_oracle
from golden_util import rel_err

SHAPES = [(130, 136, 72), (257, 128, 64), (70, 264, 200), (33, 64, 8), (290, 520, 328), (165, 80, 456)]


def _gemm_cases(ops, dev, shapes, tiles=(0, 4, 7, 8)):
    torch.manual_seed(0)
    for (m, n, k) in shapes:
        a = torch.randn(m, k).bfloat16().to(dev)
        b = torch.randn(n, k).bfloat16().to(dev)
        bias = torch.randn(n).to(dev)
        res = torch.randn(m, n).bfloat16().to(dev)
        ref = (a.float() @ b.float().t()).cpu()
        for tile in tiles:
            ops.gemm_tile = tile
            try:
                assert rel_err(ops.gemm_bf16(a, b, out_dtype=torch.float32), ref) < 1e-5
                assert rel_err(ops.gemm_bf16(a, b, bias=bias).float(), ref + bias.cpu()) < 6e-3            # bf16 output rounding
                assert rel_err(ops.gemm_bf16(a, b, bias=bias, res=res, epilogue=ops.EPI_RES).float(), ref + bias.cpu() + res.float().cpu()) < 6e-3
                nb = 2 if m % 2 == 0 else 1
                gate = torch.randn(nb, n).to(dev)
                c = ops.gemm_bf16(a, b, res=res.float(), gate=gate, rows_per_gate=m // nb, epilogue=ops.EPI_GATE_RES, out_dtype=torch.float32)
                g = torch.sigmoid(1 - gate.cpu()).repeat_interleave(m // nb, 0)
                assert rel_err(c, ref * g + res.float().cpu()) < 1e-5
                if n % 16 == 0:
                    c, pre = ops.gemm_bf16(a, b, bias=bias, epilogue=ops.EPI_SWIGLU, out_dtype=torch.float32, want_pre=True)
                    full = ref + bias.cpu()
                    assert rel_err(pre, full) < 1e-5
                    assert rel_err(c, full[:, :n // 2] * torch.nn.functional.silu(full[:, n // 2:])) < 1e-5
                assert rel_err(ops.gemm_bf16(a, b, out_dtype=torch.float32, splits=2), ref) < 1e-5
                # split-K with the bias / residual applied by sat_splitk_epilogue (the few-tile / long-K FF2 projection)
                if n % 4 == 0:
                    for s_ in (2, 3):
                        assert rel_err(ops.gemm_bf16_splitk(a, b, s_, bias=bias, res=res.float(), out_dtype=torch.float32),
                                       ref + bias.cpu() + res.float().cpu()) < 1e-5
                        assert rel_err(ops.gemm_bf16_splitk(a, b, s_, bias=bias, res=res).float(), ref + bias.cpu() + res.float().cpu()) < 6e-3
            finally:
                ops.gemm_tile = None


def _heads_case(ops, dev, nb, ntok, heads, k, tiles=(0, 4, 7, 8)):
    torch.manual_seed(1)
    x = torch.randn(nb * ntok, k).bfloat16().to(dev)
    w = (torch.randn(3 * heads * 64, k) / k ** 0.5).bfloat16().to(dev)
    inv = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
    cs = ops.rope_tables(inv.to(dev), ntok + 3)          # longer table: freqs[-seq_len:] offset (transformer.py:162)
    qkv = (x.float().cpu() @ w.float().cpu().t()).view(nb, ntok, 3, heads, 64)
    q, kk, v = [qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
    freqs = dit_oracle.rotary_freqs(inv, ntok + 3)[-ntok:]
    qr, kr = dit_oracle.apply_rotary(q, freqs), dit_oracle.apply_rotary(kk, freqs)
    for tile in tiles:
        ops.gemm_tile = tile
        try:
            pl = ops.gemm_heads_bf16(x, w, cs, heads, nb, ntok, 0, 3)
            qp, kp, vp = [pl[n].view(torch.bfloat16).float().cpu() for n in ("q", "k", "v_tr")]
            assert rel_err(qp[:, :, :ntok], qr) < 6e-3 and rel_err(kp[:, :, :ntok], kr) < 6e-3
            assert rel_err(vp[:, :, :, :ntok], v.transpose(2, 3)) < 6e-3
            assert float(qp[:, :, ntok:].abs().max()) == 0.0 and float(vp[:, :, :, ntok:].abs().max()) == 0.0
            # cross-attention projections: q only (no rotary), k/v from a context of another length
            pq = ops.gemm_heads_bf16(x, w[:heads * 64], None, heads, nb, ntok, 0, 1)
            assert rel_err(pq["q"].view(torch.bfloat16).float().cpu()[:, :, :ntok], q) < 6e-3
            pkv = ops.gemm_heads_bf16(x, w[heads * 64:], None, heads, nb, ntok, 1, 2)
            assert rel_err(pkv["k"].view(torch.bfloat16).float().cpu()[:, :, :ntok], kk) < 6e-3
            assert rel_err(pkv["v_tr"].view(torch.bfloat16).float().cpu()[:, :, :, :ntok], v.transpose(2, 3)) < 6e-3
            # the planes feed the attention kernel directly
            o = ops.attention_planes(pl["q"], pl["k"], pl["v_tr"], ntok, ntok, 0.125)
            ref = torch.nn.functional.scaled_dot_product_attention(qr, kr, v).permute(0, 2, 1, 3).reshape(nb, ntok, heads * 64)
            assert rel_err(o.float(), ref) < 2e-2
        finally:
            ops.gemm_tile = None


def _prep_case(ops, dev):
    torch.manual_seed(2)
    a = torch.randn(70, 136).to(dev)
    assert rel_err(ops.cast_bf16(a).float(), a.bfloat16().float()) == 0.0
    t = ops.cast_bf16(a, transpose=True, row_pad=64)
    assert t.shape == (136, 128) and rel_err(t[:, :70].float(), a.t().bfloat16().float()) == 0.0 and float(t[:, 70:].abs().max()) == 0.0
    ab = a.bfloat16()
    assert rel_err(ops.cast_bf16(ab, transpose=True).float(), ab.t().float()) == 0.0
    # both copies of a weight in one pass (sat_cast_bf16_dual): bit-equal to the two separate casts; shapes outside its 16-byte path decline
    w = torch.randn(200, 136).to(dev)
    plain, tr = ops.cast_bf16_dual(w)
    assert torch.equal(plain, ops.cast_bf16(w)) and torch.equal(tr, ops.cast_bf16(w, transpose=True, row_pad=8))
    plain, tr = ops.cast_bf16_dual(w[:70])                                   # R = 70: two zero columns pad the transposed copy to 72
    assert tr.shape == (136, 72) and torch.equal(plain, ops.cast_bf16(w[:70])) and torch.equal(tr, ops.cast_bf16(w[:70], transpose=True, row_pad=8))
    assert ops.cast_bf16_dual(torch.randn(16, 20).to(dev)) is None
    # two transposing casts in one launch (sat_cast_bf16_tpair): bit-equal to the separate ones, fp32 and bf16 sources, different shapes
    xa, xb = torch.randn(70, 200).to(dev).bfloat16(), torch.randn(70, 136).to(dev)
    ta, tb = ops.cast_bf16_tpair(xa, xb)
    assert torch.equal(ta, ops.cast_bf16(xa, transpose=True, row_pad=8)) and torch.equal(tb, ops.cast_bf16(xb, transpose=True, row_pad=8))
    buf = torch.full((144, 72), 7.0, dtype=torch.bfloat16, device=dev)
    ops.cast_bf16_tpair(xb, xa[:, :131], out_b=buf[:131])
    assert torch.equal(buf[:131], ops.cast_bf16(xa[:, :131], transpose=True, row_pad=8)) and float(buf[131:].float().min()) == 7.0
    sa, sb = ops.split_bf16x3(a, 0).float(), ops.split_bf16x3(a, 1).float()
    c = a.shape[1]
    assert rel_err(sa[:, :c] + sa[:, 2 * c:], a) < 1e-5 and torch.equal(sa[:, :c], sa[:, c:2 * c])
    assert rel_err(sb[:, :c] + sb[:, c:2 * c], a) < 1e-5 and torch.equal(sb[:, :c], sb[:, 2 * c:])
    b = torch.randn(56, 136).to(dev)
    cc = ops.gemm_bf16(ops.split_bf16x3(a, 0), ops.split_bf16x3(b, 1), out_dtype=torch.float32)
    assert rel_err(cc, a.double().cpu() @ b.double().cpu().t()) < 2e-5        # fp32-class GEMM through the bf16x3 split


def test_gemm_epilogues_simulator(emu):
    _gemm_cases(emu, "cpu", SHAPES)


# The K loops of the eight-wave kernels are specialised at compile time on (wave group, all row blocks active) and split into staged steps
# and the <= LOOK tail steps (round 5: round 4's "lean" arm is the kernel): shapes with more K-steps than ring stages, with fewer K-steps
# than LOOK (the tail-only path), with an M tail inside a wave's rows (the partial-block specialisation) and with whole waves off.
RING_SHAPES = [(130, 136, 72), (290, 520, 328), (136, 264, 448), (40, 520, 136)]


def test_gemm_ring_k_loop_simulator(emu):
    _gemm_cases(emu, "cpu", RING_SHAPES, tiles=(4, 7, 8))
    _heads_case(emu, "cpu", 2, 70, 2, 136, tiles=(4, 7, 8))
    _fp8_case(emu, "cpu", tiles=(7, 8), shapes=((330, 272, 400),))


@pytest.mark.gpu
def test_gemm_ring_k_loop_gpu(hip):
    _gemm_cases(hip, "cuda", RING_SHAPES + [(2050, 1536, 1536), (2050, 1536, 6144), (4100, 4608, 1536)], tiles=(4, 7, 8))
    _heads_case(hip, "cuda", 2, 1025, 24, 1536, tiles=(4, 7, 8))
    _fp8_case(hip, "cuda", tiles=(7, 8))


def test_gemm_heads_epilogue_simulator(emu):
    _heads_case(emu, "cpu", 2, 71, 3, 72)
    _heads_case(emu, "cpu", 3, 70, 2, 64)      # token counts of every residue mod 4: each V^T store phase of the epilogue
    _heads_case(emu, "cpu", 2, 69, 2, 64)


def test_gemm_operand_preparation_simulator(emu):
    _prep_case(emu, "cpu")


@pytest.mark.gpu
def test_gemm_epilogues_gpu(hip):
    _gemm_cases(hip, "cuda", SHAPES + [(2050, 1536, 1536), (2050, 4608, 1536), (260, 1536, 768)])


@pytest.mark.gpu
def test_gemm_ff_shapes_gpu(hip):
    """The feed-forward pair at full size: SwiGLU projection 1536 -> 2 x 6144 and the 6144 -> 1536 output projection."""
    _gemm_cases(hip, "cuda", [(2050, 12288, 1536), (2050, 1536, 6144)], tiles=(0, 4, 7, 8))


@pytest.mark.gpu
def test_gemm_heads_epilogue_gpu(hip):
    _heads_case(hip, "cuda", 2, 71, 3, 72)
    _heads_case(hip, "cuda", 3, 70, 2, 64)
    _heads_case(hip, "cuda", 2, 69, 2, 64)
    _heads_case(hip, "cuda", 2, 1025, 24, 1536)


@pytest.mark.gpu
def test_gemm_operand_preparation_gpu(hip):
    _prep_case(hip, "cuda")


def _fp8_case(ops, dev, tiles=(None, 0, 4, 7, 8), shapes=((130, 144, 144), (330, 272, 400))):
    """fp8 e4m3 projections (BASELINE.json configs[4]): the quantiser is bit-exact against torch's float8_e4m3fn cast, the GEMM
    matches the de-quantised operands' fp32 product to 1e-4 (measured 3e-5 on gfx950: the MX MFMA's internal accumulation is not
    a plain fp32 fma chain; the simulator is exact), every epilogue included; against the un-quantised fp32
    product the distance is the format's: ~4 % relative L2 for unit-variance operands (3 mantissa bits each side)."""
    torch.manual_seed(4)
    for (m, n, k) in shapes:
        x, w = torch.randn(m, k).to(dev), (torch.randn(n, k) / 12).to(dev)
        qx, sx = ops.quant_fp8(x)
        qw, sw = ops.quant_fp8(w.bfloat16())
        assert torch.equal(qx.cpu().view(torch.float8_e4m3fn).float(), (x.cpu() / sx.cpu()).to(torch.float8_e4m3fn).float())
        xd, wd = qx.cpu().view(torch.float8_e4m3fn).float() * sx.cpu(), qw.cpu().view(torch.float8_e4m3fn).float() * sw.cpu()
        ref = xd @ wd.t()
        bias = torch.randn(n).to(dev)
        res = torch.randn(m, n).to(dev)
        full = ref + bias.cpu()
        for tile in tiles:                        # None: the shape's own pick; the four fp8 instances of csrc/gemm.hip
            ops.gemm_fp8_tile = tile
            try:
                assert rel_err(ops.gemm_fp8(qx, qw, sx * sw, out_dtype=torch.float32), ref) < 1e-4
                assert rel_err(ops.gemm_fp8(qx, qw, sx * sw, bias=bias, res=res, epilogue=ops.EPI_RES, out_dtype=torch.float32), ref + bias.cpu() + res.cpu()) < 1e-4
                c = ops.gemm_fp8(qx, qw, sx * sw, bias=bias, epilogue=ops.EPI_SWIGLU, out_dtype=torch.float32)
                assert rel_err(c, full[:, :n // 2] * torch.nn.functional.silu(full[:, n // 2:])) < 1e-4
            finally:
                ops.gemm_fp8_tile = None
        full32 = x.cpu() @ w.cpu().t()
        assert float((ref - full32).norm() / full32.norm()) < 8e-2
    # per-ROW activation scales (round 4: sat_quant_fp8_rows + row_alpha): the quantiser against torch's cast of the row-scaled tensor
    # (same fp32 arithmetic: x * (448 / row max)), the GEMM against the de-quantised operands, every epilogue, every tile; ragged row count
    for (m, n, k) in ((130, 144, 144), (203, 272, 400)):
        x, w = (torch.randn(m, k) * torch.rand(m, 1) * 4).to(dev), (torch.randn(n, k) / 12).to(dev)
        qx, rs = ops.quant_fp8_rows(x.bfloat16())
        xb = x.bfloat16().float().cpu()
        am = xb.abs().amax(dim=1).clamp_min(1e-12)
        assert torch.equal(rs.cpu(), am / 448.0)
        assert torch.equal(qx.cpu().view(torch.float8_e4m3fn).float(), (xb * (torch.full_like(am, 448.0) / am)[:, None]).to(torch.float8_e4m3fn).float())   # (tensor / tensor: an IEEE division, as the kernel's; `448.0 / am` is a reciprocal + multiply in torch)
        qx32, rs32 = ops.quant_fp8_rows(x)
        am32 = x.cpu().abs().amax(dim=1).clamp_min(1e-12)
        assert torch.equal(qx32.cpu().view(torch.float8_e4m3fn).float(), (x.cpu() * (torch.full_like(am32, 448.0) / am32)[:, None]).to(torch.float8_e4m3fn).float())
        qw, sw = ops.quant_fp8(w.bfloat16())
        xd, wd = qx.cpu().view(torch.float8_e4m3fn).float() * rs.cpu()[:, None], qw.cpu().view(torch.float8_e4m3fn).float() * sw.cpu()
        ref = xd @ wd.t()
        bias = torch.randn(n).to(dev)
        res = torch.randn(m, n).to(dev)
        full = ref + bias.cpu()
        for tile in (None, 0, 4, 7, 8):
            ops.gemm_fp8_tile = tile
            try:
                assert rel_err(ops.gemm_fp8(qx, qw, sw, out_dtype=torch.float32, row_alpha=rs), ref) < 1e-4
                assert rel_err(ops.gemm_fp8(qx, qw, sw, bias=bias, res=res, epilogue=ops.EPI_RES, out_dtype=torch.float32, row_alpha=rs),
                               ref + bias.cpu() + res.cpu()) < 1e-4
                c = ops.gemm_fp8(qx, qw, sw, bias=bias, epilogue=ops.EPI_SWIGLU, out_dtype=torch.float32, row_alpha=rs)
                assert rel_err(c, full[:, :n // 2] * torch.nn.functional.silu(full[:, n // 2:])) < 1e-4
            finally:
                ops.gemm_fp8_tile = None
        full32 = xb @ w.bfloat16().float().cpu().t()
        assert float((ref - full32).norm() / full32.norm()) < 8e-2
        # per-OUTPUT-CHANNEL weight scales (round 6: the weight's rows quantised one by one, col_alpha of the epilogues) with per-row
        # activation scales (alpha None) and with a per-tensor activation scale (alpha = the activation's); the weight rows are given
        # very different magnitudes, which is where one scale per tensor loses: the per-channel product is closer to the fp32 one
        wv = (w * (torch.rand(n, 1) * 8 + 0.05).to(dev)).bfloat16()
        qwc, swc = ops.quant_fp8_rows(wv)
        qwt, swt = ops.quant_fp8(wv)
        wdc = qwc.cpu().view(torch.float8_e4m3fn).float() * swc.cpu()[:, None]
        refc = xd @ wdc.t()
        fullc = refc + bias.cpu()
        qxt, sxt = ops.quant_fp8(x.bfloat16())
        reft = (qxt.cpu().view(torch.float8_e4m3fn).float() * sxt.cpu()) @ wdc.t()
        for tile in (None, 0, 4, 7, 8):
            ops.gemm_fp8_tile = tile
            try:
                assert rel_err(ops.gemm_fp8(qx, qwc, None, out_dtype=torch.float32, row_alpha=rs, col_alpha=swc), refc) < 1e-4
                assert rel_err(ops.gemm_fp8(qxt, qwc, sxt, out_dtype=torch.float32, col_alpha=swc), reft) < 1e-4
                assert rel_err(ops.gemm_fp8(qx, qwc, None, bias=bias, res=res, epilogue=ops.EPI_RES, out_dtype=torch.float32, row_alpha=rs, col_alpha=swc),
                               refc + bias.cpu() + res.cpu()) < 1e-4
                c = ops.gemm_fp8(qx, qwc, None, bias=bias, epilogue=ops.EPI_SWIGLU, out_dtype=torch.float32, row_alpha=rs, col_alpha=swc)
                assert rel_err(c, fullc[:, :n // 2] * torch.nn.functional.silu(fullc[:, n // 2:])) < 1e-4
            finally:
                ops.gemm_fp8_tile = None
        exact = xd @ wv.float().cpu().t()
        e_chan = float((refc - exact).norm() / exact.norm())
        e_tens = float((xd @ (qwt.cpu().view(torch.float8_e4m3fn).float() * swt.cpu()).t() - exact).norm() / exact.norm())
        assert e_chan < e_tens, (e_chan, e_tens)
    # the head-split / plane-layout epilogue on fp8 operands (no rotary: the cross-attention to_q; with: to_qkv), every fp8 tile
    nb, ntok, heads, k = 2, 71, 3, 80
    x = torch.randn(nb * ntok, k).to(dev)
    w = (torch.randn(3 * heads * 64, k) / k ** 0.5).to(dev)
    qx, sx = ops.quant_fp8(x)
    qw, sw = ops.quant_fp8(w)
    xd, wd = qx.cpu().view(torch.float8_e4m3fn).float() * sx.cpu(), qw.cpu().view(torch.float8_e4m3fn).float() * sw.cpu()
    qkv = (xd @ wd.t()).view(nb, ntok, 3, heads, 64)
    q, kk, v = [qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
    for tile in (0, 4, 7, 8):
        ops.gemm_fp8_tile = tile
        try:
            pl = ops.gemm_heads_fp8(qx, qw, sx * sw, None, heads, nb, ntok, 0, 3)
            qp, kp, vp = [pl[nm].view(torch.bfloat16).float().cpu() for nm in ("q", "k", "v_tr")]
            assert rel_err(qp[:, :, :ntok], q) < 6e-3 and rel_err(kp[:, :, :ntok], kk) < 6e-3
            assert rel_err(vp[:, :, :, :ntok], v.transpose(2, 3)) < 6e-3
            # ... with per-row activation scales (the transposed V path multiplies four rows' factors per lane)
            qxr, rsr = ops.quant_fp8_rows(x.bfloat16())
            qkv_r = ((qxr.cpu().view(torch.float8_e4m3fn).float() * rsr.cpu()[:, None]) @ wd.t()).view(nb, ntok, 3, heads, 64)
            qr_, kr_, vr_ = [qkv_r[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
            pl = ops.gemm_heads_fp8(qxr, qw, sw, None, heads, nb, ntok, 0, 3, row_alpha=rsr)
            qp, kp, vp = [pl[nm].view(torch.bfloat16).float().cpu() for nm in ("q", "k", "v_tr")]
            assert rel_err(qp[:, :, :ntok], qr_) < 6e-3 and rel_err(kp[:, :, :ntok], kr_) < 6e-3
            assert rel_err(vp[:, :, :, :ntok], vr_.transpose(2, 3)) < 6e-3
            # ... and with the rotary (the partner column of a rotated pair is read from the window: it needs the row factor too)
            inv = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
            cs = ops.rope_tables(inv.to(dev), ntok)
            freqs = dit_oracle.rotary_freqs(inv, ntok)
            pl = ops.gemm_heads_fp8(qxr, qw, sw, cs, heads, nb, ntok, 0, 3, row_alpha=rsr)
            qp, kp = [pl[nm].view(torch.bfloat16).float().cpu() for nm in ("q", "k")]
            assert rel_err(qp[:, :, :ntok], dit_oracle.apply_rotary(qr_, freqs)) < 6e-3
            assert rel_err(kp[:, :, :ntok], dit_oracle.apply_rotary(kr_, freqs)) < 6e-3
            # ... and per-output-channel weight scales through the same epilogues (transposed V: one factor per head dim; rotary: the
            # partner column has its own factor)
            wvar = (w * (torch.rand(w.shape[0], 1) * 8 + 0.05).to(dev)).bfloat16()
            qwc, swc = ops.quant_fp8_rows(wvar)
            wdc = qwc.cpu().view(torch.float8_e4m3fn).float() * swc.cpu()[:, None]
            qkv_c = ((qxr.cpu().view(torch.float8_e4m3fn).float() * rsr.cpu()[:, None]) @ wdc.t()).view(nb, ntok, 3, heads, 64)
            qc_, kc_, vc_ = [qkv_c[:, :, i].permute(0, 2, 1, 3) for i in range(3)]
            pl = ops.gemm_heads_fp8(qxr, qwc, None, cs, heads, nb, ntok, 0, 3, row_alpha=rsr, col_alpha=swc)
            qp, kp, vp = [pl[nm].view(torch.bfloat16).float().cpu() for nm in ("q", "k", "v_tr")]
            assert rel_err(qp[:, :, :ntok], dit_oracle.apply_rotary(qc_, freqs)) < 6e-3
            assert rel_err(kp[:, :, :ntok], dit_oracle.apply_rotary(kc_, freqs)) < 6e-3
            assert rel_err(vp[:, :, :, :ntok], vc_.transpose(2, 3)) < 6e-3
        finally:
            ops.gemm_fp8_tile = None


def test_gemm_fp8_simulator(emu):
    _fp8_case(emu, "cpu")


@pytest.mark.gpu
def test_gemm_fp8_gpu(hip):
    _fp8_case(hip, "cuda")
