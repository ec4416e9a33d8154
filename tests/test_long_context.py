"""BASELINE.json configs[4] — Stable-Audio-2.0-style long context: 285 s of stereo audio = 12 582 912 samples -> 6144 latent
frames -> N = 6145 tokens (reference configs/model_configs/txt2audio/stable_audio_2_0.json:3), fp8 projections.

  * attention forward / backward at N = 6145 against float64 SDPA (a 2-head slice) and size-independent properties at the full
    24 heads; LayerNorm and the projection GEMMs at M = 6145;
  * one DiT block stack (depth 1) at N = 6145 in float32 against the CPU oracle at 1e-3, and the same block in bf16 and with fp8
    projections against that float32 result with the bounds stated at the asserts;
  * a full depth-24 bf16-mixed training step at N = 6145, batch 1, WITHOUT activation checkpointing: peak HBM is recorded (the
    reference needs per-layer checkpointing, transformer.py:840-845; on 288 GB it is not needed) — DESIGN.md quotes the number.
The simulator twin covers the fp8 model path on a tiny config.
"""
import pytest
import torch

import dit_oracle
import seeded
from gen_golden import dit_inputs
from golden_util import rel_err

N_LONG = 6145


def l2_err(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.gpu
def test_attention_long_context_gpu(hip):
    torch.manual_seed(0)
    b, h, n, d = 1, 24, N_LONG, 64
    q, k, v = (torch.randn(b, h, n, d, device="cuda") for _ in range(3))
    o = hip.attention(q, k, v, 0.125)
    assert rel_err(hip.attention(q, k, torch.ones_like(v), 0.125), torch.ones(b, n, h * d)) < 1e-5      # softmax rows sum to one
    assert rel_err(hip.attention(q, k, 2.5 * v, 0.125), 2.5 * o) < 1e-5                                  # linear in V
    hs = 2                                                                                               # float64 reference on 2 heads
    qd, kd, vd = (t[:, :hs].double().cpu().requires_grad_(True) for t in (q, k, v))
    ref = torch.nn.functional.scaled_dot_product_attention(qd, kd, vd)
    assert rel_err(o.view(b, n, h, d)[:, :, :hs], ref.permute(0, 2, 1, 3)) < 1e-4
    # backward through the product autograd Function, same 2 heads
    from stable_audio_tools_amd.transformer import _AttentionCoreFn
    q2, k2, v2 = (t[:, :hs].contiguous().requires_grad_(True) for t in (q, k, v))
    g = torch.randn(b, n, hs * d, device="cuda")
    out = _AttentionCoreFn.apply(q2, k2, v2, 0.125)
    dq, dk, dv = torch.autograd.grad(out, (q2, k2, v2), g)
    rq, rk, rv = torch.autograd.grad(ref, (qd, kd, vd), g.cpu().double().view(b, n, hs, d).permute(0, 2, 1, 3))
    assert rel_err(dq, rq) < 1e-3 and rel_err(dk, rk) < 1e-3 and rel_err(dv, rv) < 1e-3
    ob = hip.attention(q.bfloat16(), k.bfloat16(), v.bfloat16(), 0.125)
    assert rel_err(ob.float().view(b, n, h, d)[:, :, :hs], ref.permute(0, 2, 1, 3)) < 2e-2


@pytest.mark.gpu
def test_layernorm_and_projections_long_context_gpu(hip):
    torch.manual_seed(1)
    x = torch.randn(1, N_LONG, 1536, device="cuda")
    gamma = (1 + 0.1 * torch.randn(1536, device="cuda"))
    y = hip.layernorm(x, gamma, torch.zeros(1536, device="cuda"))
    assert rel_err(y, torch.nn.functional.layer_norm(x.cpu().double(), (1536,), gamma.cpu().double())) < 1e-5
    a = x[0].bfloat16()
    w = (torch.randn(4608, 1536, device="cuda") / 39).bfloat16()
    assert rel_err(hip.gemm_bf16(a, w, out_dtype=torch.float32), a.float().cpu() @ w.float().cpu().t()) < 1e-5
    qa, sa = hip.quant_fp8(a)
    qw, sw = hip.quant_fp8(w)
    ref8 = (qa.cpu().view(torch.float8_e4m3fn).float() * sa.cpu()) @ (qw.cpu().view(torch.float8_e4m3fn).float() * sw.cpu()).t()
    assert rel_err(hip.gemm_fp8(qa, qw, sa * sw, out_dtype=torch.float32), ref8) < 1e-4      # MX MFMA accumulation: 2e-5 measured


def _block_inputs(n_lat):
    spec = seeded.FULL_DIT
    cfg = spec["config"]
    x = torch.from_numpy(seeded.seeded_array((1, cfg["io_channels"], n_lat), 3101))
    cross = torch.from_numpy(seeded.seeded_array((1, spec["context_length"], cfg["cond_token_dim"]), 3102))
    glob = torch.from_numpy(seeded.seeded_array((1, cfg["global_cond_dim"]), 3103))
    return x, torch.tensor([0.37]), cross, glob


# bounds vs the float32 result of the same weights (relative L2): bf16 storage (unit round-off 2e-3 through ~12 roundings of one
# layer) 1.5e-2; fp8 e4m3 projections (3 mantissa bits on both operands of 7 GEMMs, products averaged over K >= 768) 8e-2
BF16_BLOCK, FP8_BLOCK = 1.5e-2, 8e-2


@pytest.mark.gpu
def test_dit_block_long_context_gpu(hip):
    from stable_audio_tools_amd import linear
    from stable_audio_tools_amd.dit import DiffusionTransformer
    cfg = dict(seeded.FULL_DIT["config"], depth=1)
    model = DiffusionTransformer(**cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith("inv_freq")}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in seeded.seeded_state_dict(shapes, 3100).items()}, strict=False)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x, t, cross, glob = _block_inputs(N_LONG - 1)
    with torch.no_grad():
        ref = dit_oracle.dit_forward(sd, cfg, x, t, cross, glob)
        model = model.cuda().train(False)
        kw = dict(cross_attn_cond=cross.cuda(), global_embed=glob.cuda())
        out32 = model(x.cuda(), t.cuda(), **kw)
        assert rel_err(out32, ref) < 1e-3
        model = model.to(torch.bfloat16)
        kwb = {k: v.bfloat16() for k, v in kw.items()}
        outb = model(x.cuda().bfloat16(), t.cuda().bfloat16(), **kwb)
        eb = l2_err(outb.float(), ref)
        assert linear.set_fp8(model, True) >= 7
        out8 = model(x.cuda().bfloat16(), t.cuda().bfloat16(), **kwb)
        e8 = l2_err(out8.float(), ref)
    print(f"N=6145 DiT block: fp32 {rel_err(out32, ref):.2e}; bf16 (rel. L2) {eb:.2e}; fp8 projections {e8:.2e}")
    assert eb < BF16_BLOCK and e8 < FP8_BLOCK, (eb, e8)


# Depth 24 at N = 6145 with fp8 projections (BASELINE.json configs[4], the configuration bench.py's `long_context` object times): relative
# L2 of the FINAL output against fp32 on the same 16-bit-rounded weights.  One block with fp8 projections measures <= 2e-2 from the fp32
# block (the assert above holds it to FP8_BLOCK = 8e-2); the blocks' fresh errors are independent roundings of 3-bit mantissas entering a
# residual stream, so they add in quadrature: 2e-2 * sqrt(24) = 0.098 expected, bound = 2 x that = 0.2 (bench.py FP8_DEPTH24_BOUND is the
# same number).  The guided output before the rescale is held to the triangle inequality on the two measured half errors.
FP8_DEPTH24 = 0.2
# policy "attn" (round 6: fp8 on the attention projections only, the accuracy-first setting): the round-5 verdict's criterion for the plain
# output, 0.08; the ablation that picked the policy measured 0.043 on bench.py's weights (profiles/r06_experiments/fp8_policy/: the
# feed-forward pair carries the distance — the round's first guess, fp8 on the feed-forward pair only, was the wrong way round)
FP8_ATTN_DEPTH24 = 0.08


@pytest.mark.gpu
def test_dit_depth24_fp8_long_context_final_output_gpu(hip):
    """The timed long-context model's final output: depth 24, N = 6145, fp8 e4m3 projections, bf16 attention — conditioned half,
    unconditioned half and the guided combination (CFG scale 6, native combine kernel) against the fp32 oracle on the host (two
    batch-1 evaluations, ~30 s)."""
    from stable_audio_tools_amd import linear
    from stable_audio_tools_amd.dit import DiffusionTransformer
    cfg = seeded.FULL_DIT["config"]
    model = DiffusionTransformer(**cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith("inv_freq")}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in seeded.seeded_state_dict(shapes, seeded.FULL_DIT["seed"]).items()}, strict=False)
    model = model.to(torch.bfloat16).train(False)
    sd = {k: v.detach().float().clone() for k, v in model.state_dict().items()}        # the 16-bit-rounded weights, widened
    x, t, cross, glob = _block_inputs(N_LONG - 1)
    x, cross, glob = x.bfloat16().float(), cross.bfloat16().float(), glob.bfloat16().float()      # both sides see the same inputs
    torch.set_num_threads(min(32, __import__("os").cpu_count() or 8))
    with torch.no_grad():
        c_ref = dit_oracle.dit_forward(sd, cfg, x, t, cross, glob)
        u_ref = dit_oracle.dit_forward(sd, cfg, x, t, torch.zeros_like(cross), glob)
        g_ref = u_ref + (c_ref - u_ref) * 6.0                                           # models/dit.py:402
        model = model.cuda()
        assert linear.set_fp8(model, True) >= 7 * 24
        kw = dict(cross_attn_cond=cross.cuda().bfloat16(), global_embed=glob.cuda().bfloat16())
        xb, tb = x.cuda().bfloat16(), t.cuda().bfloat16()
        c8 = model(xb, tb, **kw).float().cpu()
        u8 = model(xb, tb, cross_attn_cond=torch.zeros_like(kw["cross_attn_cond"]), global_embed=kw["global_embed"]).float().cpu()
        g8 = model(xb, tb, cfg_scale=6.0, scale_phi=0.0, **kw).float().cpu()
    ec, eu, eg = l2_err(c8, c_ref), l2_err(u8, u_ref), l2_err(g8, g_ref)
    gb = (6.0 * float((c8 - c_ref).norm()) + 5.0 * float((u8 - u_ref).norm())) / float(g_ref.norm()) + 2.0 ** -8
    print(f"N=6145 depth-24 fp8 projections, final output (rel. L2 vs fp32): conditioned {ec:.2e} unconditioned {eu:.2e} (bound {FP8_DEPTH24}); "
          f"guided pre-rescale {eg:.2e} (triangle bound {gb:.2e})")
    assert ec < FP8_DEPTH24 and eu < FP8_DEPTH24, (ec, eu)
    assert eg <= gb, (eg, gb)
    # the accuracy-first policy: attention projections in fp8, the feed-forward pair in bf16 — same oracle results
    with torch.no_grad():
        assert linear.set_fp8(model, True, policy="attn") == 5 * 24 + 6
        ca = model(xb, tb, **kw).float().cpu()
        ua = model(xb, tb, cross_attn_cond=torch.zeros_like(kw["cross_attn_cond"]), global_embed=kw["global_embed"]).float().cpu()
    eca, eua = l2_err(ca, c_ref), l2_err(ua, u_ref)
    print(f"   policy 'attn' (fp8 on the attention projections only): conditioned {eca:.2e} unconditioned {eua:.2e} (bound {FP8_ATTN_DEPTH24})")
    assert eca < FP8_ATTN_DEPTH24 and eua < FP8_ATTN_DEPTH24 and eca < ec, (eca, eua, ec)


# Trajectory level (round 6).  10 v-DDIM steps with CFG 6 + rescale 0.75 at N = 6145 from one noise tensor; the final latents of the fp8 model
# and of the bf16 model against the float32 trajectory.  Bound, stated before the first measurement: one step moves x by sin(dtheta) * v
# with dtheta = pi / 20 (x' = cos(theta') pred + sin(theta') eps, pred = cos x - sin v, eps = sin x + cos v: d x' / d v = sin(theta' - theta)), so
# per-evaluation output errors e_i |v| accumulate to at most sum_i sin(pi / 20) e_i |v| = 1.56 e |v| if they were perfectly coherent over the
# ten steps (the errors made at different noise levels are not: quadrature would give 0.49 e |v|); with |v| ~ |x_final| and the measured
# single-evaluation guided error e = 0.34 of round 5 the coherent bound is 0.53 — FP8_TRAJECTORY = 0.5 is asserted, and the fp8 model may
# not be further from fp32 than FP8_OVER_BF16 = 8 x the bf16 model (one e4m3 rounding is 2^-4 against bf16's 2^-9 = 32 x coarser per
# operand, averaged over K >= 768 products per output and 7 of a layer's ~12 roundings: measured single-evaluation ratio 5 in round 5).
FP8_TRAJECTORY, FP8_OVER_BF16 = 0.5, 8.0


@pytest.mark.gpu
def test_dit_fp8_long_context_trajectory_gpu(hip):
    from golden_util import dit_trajectory_distances
    from stable_audio_tools_amd import linear
    from stable_audio_tools_amd.dit import DiffusionTransformer
    cfg = seeded.FULL_DIT["config"]
    model = DiffusionTransformer(**cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith("inv_freq")}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in seeded.seeded_state_dict(shapes, seeded.FULL_DIT["seed"]).items()}, strict=False)
    model = model.to(torch.bfloat16).train(False).cuda()
    assert linear.set_fp8(model, True) >= 7 * 24
    x, _, cross, glob = _block_inputs(N_LONG - 1)
    kw = dict(cross_attn_cond=cross.cuda().bfloat16(), global_embed=glob.cuda().bfloat16(), cfg_scale=6.0, scale_phi=0.75)
    d = dit_trajectory_distances(model, cfg, x.cuda().bfloat16(), kw, steps=10)
    print(f"N=6145 depth-24, 10 v-DDIM steps, CFG 6: final latents vs the fp32 trajectory (rel. L2): fp8 projections {d['lowp']:.3e}, bf16 {d['bf16']:.3e} "
          f"(ratio {d['lowp'] / d['bf16']:.2f}; bounds {FP8_TRAJECTORY}, {FP8_OVER_BF16} x)")
    assert d["finite"] and d["lowp"] < FP8_TRAJECTORY and d["lowp"] < FP8_OVER_BF16 * d["bf16"], d


@pytest.mark.gpu
def test_dit_train_step_long_context_memory_gpu(hip):
    """Depth-24 bf16-mixed training step at N = 6145, batch 1, every activation kept resident (no checkpointing)."""
    from stable_audio_tools_amd.dit import DiffusionTransformer
    from stable_audio_tools_amd.training import DiTTrainStep
    torch.manual_seed(0)
    model = DiffusionTransformer(**seeded.FULL_DIT["config"])
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith("to_out.weight") or ".ff.ff.2." in n_ or "process_conv" in n_:
                p.normal_(0.0, 0.02)
    model = model.cuda().train(True)
    stepper = DiTTrainStep(model, lr=1e-5, cfg_dropout_prob=0.1, autocast_dtype=torch.bfloat16)
    x, _, cross, glob = _block_inputs(N_LONG - 1)
    torch.cuda.reset_peak_memory_stats()
    losses = [float(stepper(x.cuda(), cross_attn_cond=cross.cuda(), global_embed=glob.cuda())["loss"]) for _ in range(2)]
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    print(f"N=6145 depth-24 bf16-mixed train step, batch 1, no checkpointing: peak HBM {peak:.1f} GiB; losses {losses}")
    assert all(torch.isfinite(torch.tensor(losses))) and peak < 200.0


def test_dit_fp8_projections_simulator(emu_modules):
    """fp8 forward on the tiny config (simulator): same bound reasoning, 4 layers."""
    from stable_audio_tools_amd import linear
    from stable_audio_tools_amd.dit import DiffusionTransformer
    name = "tiny_prepend"
    cfg = seeded.DIT_CONFIGS[name]
    model = DiffusionTransformer(**cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith("inv_freq")}
    model.load_state_dict({k: torch.from_numpy(v) for k, v in seeded.seeded_state_dict(shapes, 700).items()}, strict=False)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    inp = dit_inputs(name)
    with torch.no_grad():
        ref = dit_oracle.dit_forward(sd, cfg, inp["x"], inp["t"], inp["cross_attn_cond"], inp["global_embed"])
        model = model.to(torch.bfloat16).train(False)
        assert linear.set_fp8(model, True, min_features=64) > 0
        out = model(inp["x"].bfloat16(), inp["t"].bfloat16(), cross_attn_cond=inp["cross_attn_cond"].bfloat16(),
                    global_embed=inp["global_embed"].bfloat16())
    assert l2_err(out.float(), ref) < 1.5e-1
