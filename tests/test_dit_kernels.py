"""Kernel-level parity of the DiT elementwise / normalisation kernels against torch (the reference's own building
blocks: F.layer_norm, transformer.py:236-241, and the adaLN modulate of transformer.py:675-701).
CPU leg under the host simulator, GPU leg through the gfx950 library; d = 1536 walks the 16-byte vector path, d = 200
the scalar path."""
import pytest
import torch
import torch.nn.functional as F

from golden_util import rel_err


def _ln_case(ops, dev, dtype, b, n, d, ada, seed):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(b, n, d, generator=gen).to(dev).to(dtype)
    gamma = (1.0 + 0.2 * torch.randn(d, generator=gen)).to(dev)
    beta = (0.1 * torch.randn(d, generator=gen)).to(dev)
    mod = (0.3 * torch.randn(b, 6 * d, generator=gen)).to(dev).to(dtype)
    scale, shift = (mod[:, d:2 * d], mod[:, 3 * d:4 * d]) if ada else (None, None)
    dy = torch.randn(b, n, d, generator=gen).to(dev).to(dtype)

    y, mean, rstd = ops.layernorm(x, gamma, beta, scale, shift, 1e-5, save_stats=True)
    dx, dgamma, dscale, dshift = ops.layernorm_bwd(dy, x, gamma, beta, scale, mean, rstd)

    xr = x.float().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    sc = scale.float().clone().requires_grad_(True) if ada else None
    sh = shift.float().clone().requires_grad_(True) if ada else None
    ref = F.layer_norm(xr, (d,), gr, beta, 1e-5)
    if ada:
        ref = ref * (1.0 + sc[:, None, :]) + sh[:, None, :]
    ref.backward(dy.float())
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2

    def close(a, r, scale_=1.0):
        err = (a.float() - r).abs().max().item()
        assert err <= tol * scale_ * max(r.abs().max().item(), 1e-3), (err, r.abs().max().item())
    close(y, ref.detach())
    close(dx, xr.grad)
    close(dgamma, gr.grad, 4.0)
    if ada:
        close(dscale, sc.grad, 4.0)
        close(dshift, sh.grad, 4.0)
    # the residual-path addend of dx (sat_layernorm_bwd_res): dx + dres in the kernel's own pass; parameter gradients unchanged
    dres = torch.randn(b, n, d, generator=gen).to(dev).to(dtype)
    dx2, dgamma2, _, _ = ops.layernorm_bwd(dy, x, gamma, beta, scale, mean, rstd, dres=dres)
    close(dx2, xr.grad + dres.float())
    assert torch.equal(dgamma2, dgamma)


CASES = [(torch.float32, 2, 9, 1536, True), (torch.float32, 1, 7, 200, False), (torch.bfloat16, 2, 9, 1536, True),
         (torch.bfloat16, 1, 5, 1536, False), (torch.bfloat16, 2, 6, 256, True), (torch.float32, 1, 5, 256, False)]


@pytest.mark.parametrize("case", CASES)
def test_layernorm_sim(emu, case):
    _ln_case(emu, "cpu", *case, seed=11)


@pytest.mark.gpu
def test_layernorm_gpu(hip):
    for case in CASES + [(torch.bfloat16, 4, 1025, 1536, True), (torch.float32, 2, 1025, 1536, False)]:
        _ln_case(hip, "cuda", *case, seed=12)


def _ln_fp8_case(ops, dev, b, n, d, mod, seed):
    """LayerNorm -> fp8 rows (sat_layernorm_fwd_fp8, round 4): the de-quantised rows against the fp32 LayerNorm (+ adaLN) of the same
    bf16 input within e4m3's half-step (2^-4 relative to the row's own scale), and the row scale = max |LN row| / 448."""
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(b, n, d, generator=gen).bfloat16().to(dev)
    gamma = (1 + 0.1 * torch.randn(d, generator=gen)).to(dev)
    beta = (0.1 * torch.randn(d, generator=gen)).to(dev)
    sc = sh = None
    if mod:
        mo = (0.2 * torch.randn(b, 2 * d, generator=gen)).bfloat16().to(dev)
        sc, sh = mo[:, :d], mo[:, d:]
    out = ops.layernorm_fp8(x, gamma, beta, sc, sh, 1e-5)
    assert out is not None
    q, rs = out
    ref = F.layer_norm(x.float().cpu(), (d,), gamma.cpu(), beta.cpu(), 1e-5)
    if mod:
        ref = ref * (1 + sc.float().cpu()[:, None, :]) + sh.float().cpu()[:, None, :]
    ref = ref.reshape(b * n, d)
    am = ref.abs().amax(dim=1)
    assert rel_err(rs.cpu(), am / 448.0) < 1e-5
    deq = q.cpu().view(torch.float8_e4m3fn).float() * rs.cpu()[:, None]
    assert float(((deq - ref).abs() / am[:, None]).max()) < 2.0 ** -4 + 1e-3        # half a step of the top binade
    assert float((deq - ref).norm() / ref.norm()) < 4e-2


def test_layernorm_fp8_sim(emu):
    _ln_fp8_case(emu, "cpu", 2, 5, 512, False, 21)
    _ln_fp8_case(emu, "cpu", 2, 3, 1536, True, 22)
    assert emu.layernorm_fp8(torch.randn(1, 3, 72).bfloat16(), torch.ones(72), None, None, None, 1e-5) is None     # outside the vector path


@pytest.mark.gpu
def test_layernorm_fp8_gpu(hip):
    _ln_fp8_case(hip, "cuda", 2, 1025, 1536, False, 23)
    _ln_fp8_case(hip, "cuda", 2, 130, 1536, True, 24)


# ---------------------------------------------------------------------------------------------------------------------
# attention: forward and backward against F.scaled_dot_product_attention (the reference's CPU path, transformer.py:440),
# self and GQA cross shapes, sequence lengths around the 64-wide tiles (the kernels double-buffer them)
# ---------------------------------------------------------------------------------------------------------------------
ATT_CASES = [  # (B, H, Hkv, Nq, Nk)
    (1, 2, 2, 64, 64), (2, 2, 2, 65, 65), (1, 4, 2, 130, 37), (1, 2, 1, 37, 200), (1, 2, 2, 193, 193), (1, 4, 4, 1, 70),
    (1, 4, 2, 100, 130), (1, 2, 2, 40, 256), (1, 2, 1, 70, 257),      # the DiT's 130 context tokens on a GQA pair; the short-key kernels' limit and one past it
]


def _attn_case(ops, dev, dtype, case, seed, spikes=(), cross=True):
    """cross: bf16 shapes with Nk <= 256 run on the short-key kernels (csrc/attention_cross.h) when True, on the general flash-style
    kernels when False — both implementations are held to the same SDPA reference."""
    b, h, hkv, nq, nk = case
    saved = ops.cross_kernels, ops.attn_q64
    ops.cross_kernels = cross
    try:
        assert ops.cross_ok(h, hkv, nk, 64, 1) == (cross and nk <= 256)
        # the general bf16 forward has two kernel shapes (32 / 64 queries per wave, picked by grid size): both are held to the reference
        for q64 in ((False, True) if (dtype == torch.bfloat16 and not ops.cross_ok(h, hkv, nk, 64, 1)) else (None,)):
            ops.attn_q64 = q64
            _attn_case_body(ops, dev, dtype, case, seed, spikes)
    finally:
        ops.cross_kernels, ops.attn_q64 = saved


def _attn_case_body(ops, dev, dtype, case, seed, spikes):
    b, h, hkv, nq, nk = case
    gen = torch.Generator().manual_seed(seed)
    q = torch.randn(b, h, nq, 64, generator=gen)
    k = torch.randn(b, hkv, nk, 64, generator=gen)
    for (qi, ki, gain) in spikes:        # key ki lines up with query qi: its score jumps by gain * |q|^2 / 8 at that key tile
        k[:, :, ki] = gain * q[:, ::h // hkv, qi]
    q, k = q.to(dev).to(dtype), k.to(dev).to(dtype)
    v = torch.randn(b, hkv, nk, 64, generator=gen).to(dev).to(dtype)
    do = torch.randn(b, nq, h * 64, generator=gen).to(dev).to(dtype)
    o, lse, planes = ops.attention(q, k, v, 0.125, return_planes=True)
    dq, dk, dv = ops.attention_bwd(planes, o, do, lse, 0.125, hkv, nk)

    qr, kr, vr = (t.float().detach().requires_grad_(True) for t in (q, k, v))
    rep = h // hkv
    ref = F.scaled_dot_product_attention(qr, kr.repeat_interleave(rep, 1), vr.repeat_interleave(rep, 1), scale=0.125)
    ref = ref.permute(0, 2, 1, 3).reshape(b, nq, h * 64)
    ref.backward(do.float())
    tol = 2e-4 if dtype == torch.float32 else 3e-2

    def close(a, r):
        err = (a.float() - r).abs().max().item()
        assert err <= tol * max(r.abs().max().item(), 1e-2), (case, err, r.abs().max().item())
    close(o, ref.detach())
    close(dq, qr.grad)
    close(dk, kr.grad)
    close(dv, vr.grad)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", ATT_CASES)
def test_attention_sim(emu, case, dtype):
    _attn_case(emu, "cpu", dtype, case, seed=21)
    if dtype == torch.bfloat16 and case[4] <= 256:
        _attn_case(emu, "cpu", dtype, case, seed=21, cross=False)


# The forward moves its running max only when a row outgrows it by more than 2^4 (deferred rescale): rows whose max jumps far past the
# threshold late in the sequence (the branch), rows that grow by less than it (stale max, P > 1), and both in one wave.
SPIKES = [(5, 150, 6.0), (7, 100, 0.3), (40, 190, 0.5), (41, 70, 3.0), (64, 192, 8.0)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_deferred_max_sim(emu, dtype):
    for cross in ((True, False) if dtype == torch.bfloat16 else (True,)):      # the one-pass softmax of the short-key kernels sees the same spikes
        _attn_case(emu, "cpu", dtype, (1, 2, 2, 130, 193), seed=23, spikes=SPIKES, cross=cross)
        _attn_case(emu, "cpu", dtype, (1, 4, 2, 70, 200), seed=24, spikes=SPIKES[:4], cross=cross)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_deferred_max_gpu(hip, dtype):
    for cross in ((True, False) if dtype == torch.bfloat16 else (True,)):
        _attn_case(hip, "cuda", dtype, (1, 2, 2, 130, 193), seed=23, spikes=SPIKES, cross=cross)
    _attn_case(hip, "cuda", dtype, (2, 24, 24, 1025, 1025), seed=25, spikes=SPIKES + [(1000, 1024, 5.0), (1024, 3, 4.0)])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_gpu(hip, dtype):
    for case in ATT_CASES + [(2, 24, 24, 1025, 1025), (2, 24, 12, 1025, 130), (4, 24, 12, 1025, 130), (16, 24, 12, 1025, 130), (1, 4, 4, 300, 6145)]:
        _attn_case(hip, "cuda", dtype, case, seed=22)
        if dtype == torch.bfloat16 and case[4] <= 256:
            _attn_case(hip, "cuda", dtype, case, seed=22, cross=False)


def _cfg_step_case(ops, dev):
    """sat_cfg_step vs the reference formulas (models/dit.py:400-410 + the v-DDIM update of inference/sampling.py:254-307)."""
    torch.manual_seed(3)
    b, c, t = 2, 6, 37
    out2 = torch.randn(2 * b, c, t).to(dev)
    x = torch.randn(b, c, t).to(dev)
    scale, phi = 6.0, 0.75
    cond, uncond = torch.chunk(out2.cpu(), 2, dim=0)
    g = uncond + (cond - uncond) * scale
    ref = phi * (g * (cond.std(dim=1, keepdim=True) / g.std(dim=1, keepdim=True))) + (1 - phi) * g
    assert rel_err(ops.cfg_step(out2, 2, scale, phi), ref) < 1e-5
    assert rel_err(ops.cfg_step(out2, 2, scale, 0.0), g) < 1e-6
    coef = (0.3, -0.7, 1.1, 0.2)
    y0, y1 = ops.cfg_step(out2, 2, scale, phi, x=x, coef=coef, want_second=True)
    assert rel_err(y0, coef[0] * x.cpu() + coef[1] * ref) < 1e-5 and rel_err(y1, coef[2] * x.cpu() + coef[3] * ref) < 1e-5
    y0, y1 = ops.cfg_step(out2[:b].contiguous(), 1, x=x, coef=coef, want_second=True)
    assert rel_err(y0, coef[0] * x.cpu() + coef[1] * out2[:b].cpu()) < 1e-6
    ob = ops.cfg_step(out2.bfloat16(), 2, scale, phi)
    assert ob.dtype == torch.bfloat16 and rel_err(ob.float(), ref) < 2e-2
    # the general form (sat_sampler_step): third operand + unconditioned-output terms, host and device coefficients
    prev = torch.randn(b, c, t).to(dev)
    c8 = (0.3, -0.7, 0.45, 0.15, 1.1, 0.2, -0.6, 0.8)
    want0 = c8[0] * x.cpu() + c8[1] * ref + c8[2] * prev.cpu() + c8[3] * uncond
    want1 = c8[4] * x.cpu() + c8[5] * ref + c8[6] * prev.cpu() + c8[7] * uncond
    y0, y1 = ops.cfg_step(out2, 2, scale, phi, x=x, coef=c8, want_second=True, prev=prev)
    assert rel_err(y0, want0) < 1e-5 and rel_err(y1, want1) < 1e-5
    y0d, y1d = ops.cfg_step(out2, 2, scale, phi, x=x, coef=torch.tensor(c8, dtype=torch.float32, device=dev), want_second=True, prev=prev)
    assert torch.equal(y0d, y0) and torch.equal(y1d, y1)
    y0n, _ = ops.cfg_step(out2[:b].contiguous(), 1, x=x, coef=c8, want_second=True, prev=prev)      # ncond 1: u = v
    assert rel_err(y0n, c8[0] * x.cpu() + (c8[1] + c8[3]) * out2[:b].cpu() + c8[2] * prev.cpu()) < 1e-5
    with pytest.raises(ValueError):
        ops.cfg_step(out2, 2, scale, phi, x=x, coef=c8, want_second=True)                           # coefficient on prev without prev


def test_cfg_step_simulator(emu):
    _cfg_step_case(emu, "cpu")


@pytest.mark.gpu
def test_cfg_step_gpu(hip):
    _cfg_step_case(hip, "cuda")
