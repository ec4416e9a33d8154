"""Parity of the native DiffusionTransformer (forward / sampling path) against golden vectors produced
by the reference's DiffusionTransformer and against the oracle (oracle/dit_oracle.py).

fp32 mode: 1e-3 relative (BASELINE.json) — the attention kernel runs its bf16x3 split on fp32 inputs.
bf16 mode (the perf configuration of configs[2..4]): compared with the fp32 oracle at 3e-2, the
tolerance of bf16 storage (8-bit mantissa) through 4 residual layers; stated here because the
north-star's 1e-3 applies to fp32.
"""
import numpy as np
import pytest
import torch

import dit_oracle
import seeded
from gen_golden import dit_inputs
from golden_util import load_golden, rel_err

TOL = 1e-3
NAMES = list(seeded.DIT_CONFIGS)


def _build(name, seed, device, dtype=torch.float32):
    from stable_audio_tools_amd.dit import DiffusionTransformer
    model = DiffusionTransformer(**seeded.DIT_CONFIGS[name])
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith("inv_freq")}
    sd = {k: torch.from_numpy(v) for k, v in seeded.seeded_state_dict(shapes, seed).items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("inv_freq") for k in missing)
    full_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return model.to(device=device, dtype=dtype).train(False), full_sd


def _case(name, idx, device):
    g = load_golden("dit_" + name)
    model, sd = _build(name, 700 + 10 * idx, device)
    assert sorted(model.state_dict().keys()) == list(g["keys"]), "state_dict keys differ from the reference"
    inp = {k: v.to(device) for k, v in dit_inputs(name).items()}
    kw = dict(cross_attn_cond=inp["cross_attn_cond"], global_embed=inp["global_embed"], prepend_cond=inp.get("prepend_cond"),
              prepend_cond_mask=inp.get("prepend_cond_mask"))
    with torch.no_grad():
        plain = model(inp["x"], inp["t"], cfg_scale=1.0, **kw)
        guided = model(inp["x"], inp["t"], cfg_scale=6.0, scale_phi=0.75, **kw)
        _, info = model(inp["x"], inp["t"], return_info=True, **kw)
    assert rel_err(info["hidden_states"][0], g["hidden_first"]) < TOL
    assert rel_err(info["hidden_states"][-1], g["hidden_last"]) < TOL
    assert rel_err(plain, g["plain"]) < TOL
    assert rel_err(guided, g["guided"]) < TOL
    # oracle agrees with the reference too (pins oracle/dit_oracle.py)
    cpu = dit_inputs(name)
    o_plain = dit_oracle.dit_forward(sd, seeded.DIT_CONFIGS[name], cpu["x"], cpu["t"], cpu["cross_attn_cond"], cpu["global_embed"],
                                     cpu.get("prepend_cond"))
    o_guided = dit_oracle.dit_forward(sd, seeded.DIT_CONFIGS[name], cpu["x"], cpu["t"], cpu["cross_attn_cond"], cpu["global_embed"],
                                      cpu.get("prepend_cond"), cfg_scale=6.0, scale_phi=0.75)
    assert rel_err(o_plain, g["plain"]) < 1e-4
    assert rel_err(o_guided, g["guided"]) < 1e-4


@pytest.mark.parametrize("idx,name", list(enumerate(NAMES)))
def test_dit_matches_reference_golden_simulator(emu_modules, idx, name):
    _case(name, idx, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("idx,name", list(enumerate(NAMES)))
def test_dit_matches_reference_golden_gpu(hip, idx, name):
    _case(name, idx, "cuda")


def _bf16(device):
    name = "tiny_adaln"
    model, sd = _build(name, 710, device, dtype=torch.bfloat16)
    inp = dit_inputs(name)
    kw = dict(cross_attn_cond=inp["cross_attn_cond"].to(device), global_embed=inp["global_embed"].to(device))
    with torch.no_grad():
        out = model(inp["x"].to(device), inp["t"].to(device), cfg_scale=1.0, **kw)
    assert out.dtype == torch.bfloat16
    # oracle in fp32 on the SAME bf16-rounded weights and inputs: isolates kernel/activation rounding from
    # weight quantisation (guidance at scale 6 would multiply that rounding noise by 6, so it is compared unguided)
    def q(a):
        return a.to(torch.bfloat16).float() if a.is_floating_point() else a
    sdq = {k: q(v) for k, v in sd.items()}
    ref = dit_oracle.dit_forward(sdq, seeded.DIT_CONFIGS[name], q(inp["x"]), q(inp["t"]), q(inp["cross_attn_cond"]), q(inp["global_embed"]))
    assert rel_err(out.float(), ref) < 3e-2


def test_dit_bf16_simulator(emu_modules):
    _bf16("cpu")


def _inference_caches(device):
    """The no-grad bf16 path keeps the embedded conditioning (dit._embed_cond, the CFG batch) and every cross-attention layer's K / V
    planes while the caller passes the SAME conditioning tensor (a sampler does, at every step).  A second call must hit the caches and
    return the same values; an in-place change of the conditioning, another tensor with other values, or new weights must miss."""
    from stable_audio_tools_amd.transformer import Attention
    name = "tiny_adaln"
    model, _ = _build(name, 710, device, dtype=torch.bfloat16)
    inp = dit_inputs(name)
    x, t, g = inp["x"].to(device), inp["t"].to(device), inp["global_embed"].to(device)
    cond = inp["cross_attn_cond"].to(device).to(torch.bfloat16)
    cross = [m for m in model.modules() if isinstance(m, Attention) and hasattr(m, "to_q")]
    assert cross

    from stable_audio_tools_amd.dit import clear_inference_caches

    def run(c, fresh=False):
        if fresh:
            clear_inference_caches(model)
        with torch.no_grad():
            return model(x, t, cross_attn_cond=c, global_embed=g, cfg_scale=4.0, scale_phi=0.5).float().clone()

    first = cond.clone()
    a = run(cond, fresh=True)
    planes = [m._kv_planes["k"].data_ptr() for m in cross]
    assert len(set(planes)) == len(cross)                      # one buffer per layer
    ctx0 = cross[0]._kv_ctx
    b = run(cond)                                              # same object: every cache hits
    assert cross[0]._kv_ctx is ctx0 and torch.equal(a, b)
    cond.mul_(1.5)                                             # in-place edit: version counter -> miss
    c = run(cond)
    assert cross[0]._kv_ctx is not ctx0 and not torch.equal(a, c)
    assert torch.equal(c, run(cond, fresh=True))
    other = first                                              # another tensor with the first values -> first result again
    assert torch.equal(run(other), a)
    with torch.no_grad():
        cross[0].to_kv.weight.mul_(0.5)                        # new weights -> that layer re-projects
    assert torch.equal(run(other), run(other, fresh=True))


def test_inference_caches_simulator(emu_modules):
    _inference_caches("cpu")


@pytest.mark.gpu
def test_inference_caches_gpu(hip):
    _inference_caches("cuda")


@pytest.mark.gpu
def test_dit_bf16_gpu(hip):
    _bf16("cuda")


@pytest.mark.gpu
def test_attention_full_size_properties_gpu(hip):
    """Size-independent checks at the BASELINE shapes (N = 1025 self, M = 130 GQA cross, 24 heads):
    (1) rows of softmax sum to one -> attention of constant V returns the constant;
    (2) linearity in V; (3) bf16 and fp32-split paths agree to bf16 precision."""
    torch.manual_seed(0)
    b, h, n, d = 2, 24, 1025, 64
    q = torch.randn(b, h, n, d, device="cuda")
    k = torch.randn(b, h, n, d, device="cuda")
    v = torch.randn(b, h, n, d, device="cuda")
    ones = torch.ones_like(v)
    assert rel_err(hip.attention(q, k, ones, 0.125), torch.ones(b, n, h * d)) < 1e-5
    o1 = hip.attention(q, k, v, 0.125)
    o2 = hip.attention(q, k, 2.5 * v, 0.125)
    assert rel_err(o2, 2.5 * o1) < 1e-5
    ref = torch.nn.functional.scaled_dot_product_attention(q.double().cpu(), k.double().cpu(), v.double().cpu())
    ref = ref.permute(0, 2, 1, 3).reshape(b, n, h * d)
    assert rel_err(o1, ref) < 1e-4
    ob = hip.attention(q.bfloat16(), k.bfloat16(), v.bfloat16(), 0.125)
    assert rel_err(ob.float(), ref) < 2e-2
    kc = torch.randn(b, 12, 130, d, device="cuda")
    vc = torch.randn(b, 12, 130, d, device="cuda")
    oc = hip.attention(q, kc, vc, 0.125)
    refc = torch.nn.functional.scaled_dot_product_attention(q.double().cpu(), kc.double().cpu().repeat_interleave(2, 1),
                                                            vc.double().cpu().repeat_interleave(2, 1))
    assert rel_err(oc, refc.permute(0, 2, 1, 3).reshape(b, n, h * d)) < 1e-4


def _gradients(name, idx, device):
    """DiT training gradients (v-objective MSE, training/diffusion.py:406-449 restated: noised = x*alpha + n*sigma,
    target = n*alpha - x*sigma, loss = mse(model(noised, t), target)) against autograd through the oracle."""
    import math
    model, sd = _build(name, 700 + 10 * idx, device)
    model.train(True)
    inp = dit_inputs(name)
    x0, t = inp["x"], inp["t"]
    noise = torch.from_numpy(seeded.seeded_array(tuple(x0.shape), 999))
    alpha, sigma = torch.cos(t * math.pi / 2)[:, None, None], torch.sin(t * math.pi / 2)[:, None, None]
    noised, target = x0 * alpha + noise * sigma, noise * alpha - x0 * sigma
    kw = {k: v.to(device) for k, v in inp.items() if k in ("cross_attn_cond", "global_embed", "prepend_cond", "prepend_cond_mask")}
    xin = noised.to(device).requires_grad_(True)
    out = model(xin, t.to(device), **kw)
    loss = torch.nn.functional.mse_loss(out, target.to(device))
    names = [n for n, p in model.named_parameters()]
    grads = torch.autograd.grad(loss, [xin] + list(model.parameters()))
    sdo = {k: v.clone().requires_grad_(v.is_floating_point() and not k.endswith("inv_freq") and not k.endswith(".beta")) for k, v in sd.items()}
    xo = noised.clone().requires_grad_(True)
    oo = dit_oracle.dit_forward(sdo, seeded.DIT_CONFIGS[name], xo, t, inp["cross_attn_cond"], inp["global_embed"], inp.get("prepend_cond"))
    lo = torch.nn.functional.mse_loss(oo, target)
    gref = torch.autograd.grad(lo, [xo] + [sdo[n] for n in names])
    assert abs(float(loss) - float(lo)) <= 1e-4 * abs(float(lo))
    worst = ("", 0.0)
    for n, a, b in zip(["<input>"] + names, grads, gref):
        e = rel_err(a, b)
        if e > worst[1]:
            worst = (n, e)
    assert worst[1] < TOL, worst


@pytest.mark.parametrize("idx,name", list(enumerate(NAMES)))
def test_dit_training_gradients_simulator(emu_modules, idx, name):
    _gradients(name, idx, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("idx,name", list(enumerate(NAMES)))
def test_dit_training_gradients_gpu(hip, idx, name):
    _gradients(name, idx, "cuda")


def _sampler_case(device, use_graph, steps=5):
    """Five v-DDIM steps with CFG through the native DiT vs the same loop (reference inference/sampling.py:254-307
    restated in stable_audio_tools_amd/sampling.py) around the CPU oracle's forward."""
    from stable_audio_tools_amd.sampling import get_alphas_sigmas, sample_v_ddim
    name, idx = NAMES[1], 1
    model, sd = _build(name, 700 + 10 * idx, device)
    inp = dit_inputs(name)
    kw = dict(cross_attn_cond=inp["cross_attn_cond"], global_embed=inp["global_embed"], prepend_cond=inp.get("prepend_cond"),
              prepend_cond_mask=inp.get("prepend_cond_mask"))
    dkw = {k: (v.to(device) if v is not None else None) for k, v in kw.items()}
    out = sample_v_ddim(model, inp["x"].to(device), steps, cfg_scale=6.0, scale_phi=0.75, use_graph=use_graph, **dkw)
    x = inp["x"]
    t = torch.linspace(1.0, 0, steps + 1)[:-1]
    alphas, sigmas = get_alphas_sigmas(t)
    for i in range(steps):
        v = dit_oracle.dit_forward(sd, seeded.DIT_CONFIGS[name], x, torch.ones(x.shape[0]) * t[i], kw["cross_attn_cond"],
                                   kw["global_embed"], kw.get("prepend_cond"), cfg_scale=6.0, scale_phi=0.75)
        pred = x * alphas[i] - v * sigmas[i]
        eps = x * sigmas[i] + v * alphas[i]
        if i < steps - 1:
            x = pred * alphas[i + 1] + eps * sigmas[i + 1]
    assert rel_err(out, pred) < TOL


def test_sampler_v_ddim_simulator(emu_modules):
    _sampler_case("cpu", False, steps=3)


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph", [False, True])
def test_sampler_v_ddim_gpu(hip, use_graph):
    _sampler_case("cuda", use_graph)


def _euler_case(device, use_graph=False):
    """Rectified-flow Euler sampler (reference inference/sampling.py:98-135) with the update fused into the guidance kernel,
    vs the same loop around the CPU oracle's forward."""
    from stable_audio_tools_amd.sampling import sample_discrete_euler
    name, idx = "small_rf", 2
    model, sd = _build(name, 700 + 10 * idx, device)
    inp = dit_inputs(name)
    kw = dict(cross_attn_cond=inp["cross_attn_cond"], global_embed=inp["global_embed"], prepend_cond=inp.get("prepend_cond"),
              prepend_cond_mask=inp.get("prepend_cond_mask"))
    dkw = {k: (v.to(device) if v is not None else None) for k, v in kw.items()}
    steps = 4
    out = sample_discrete_euler(model, inp["x"].to(device), steps, use_graph=use_graph, cfg_scale=3.0, scale_phi=0.5, **dkw)
    x = inp["x"]
    t = torch.linspace(1.0, 0, steps + 1)
    for tc, tp in zip(t[:-1], t[1:]):
        v = dit_oracle.dit_forward(sd, seeded.DIT_CONFIGS[name], x, torch.ones(x.shape[0]) * tc, kw["cross_attn_cond"],
                                   kw["global_embed"], kw.get("prepend_cond"), cfg_scale=3.0, scale_phi=0.5)
        x = x + (tp - tc) * v
    assert rel_err(out, x) < TOL


def test_sampler_euler_fused_update_simulator(emu_modules):
    _euler_case("cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("use_graph", [False, True])
def test_sampler_euler_fused_update_gpu(hip, use_graph):
    _euler_case("cuda", use_graph)
