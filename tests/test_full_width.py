"""Parity at the REAL widths of BASELINE.json configs[1] and configs[2] (short crops), against fixtures produced by the
reference itself (oracle/gen_golden_full.py -> tests/golden/full_*.npz):

  * stable_audio_2_0_vae architecture (channels 128, c_mults 1/2/4/8/16, strides 2/4/4/8/8, 156 M parameters) on a
    32768-sample stereo crop: pre-latents, z, KL, decoded audio, the generator loss (MR-STFT sum/diff + L + R, 7
    resolutions, A-weighted, + 1e-4 KL) and gradients of EVERY parameter (norm for all; full tensor or a seeded
    1024-element probe per parameter) for (i) a linear functional of the output — well conditioned, held to 1e-3 — and
    (ii) the generator loss.  Its gradient is ill-conditioned in the reference itself: the A-weighted log-magnitude
    term weights a bin by 1/|Y| and the clamp at 1e-4 (auraloss.py:385-387) switches bins on and off, so (a) the
    reference's own float32 gradient is up to 1.5e-3 from its float64 gradient (3.4e-3 for dL/d(decoded)) and (b)
    displacing the decoded audio by 1e-5 of its peak (the accuracy class of ANY non-bit-identical float32-class forward;
    ours: bf16x3 products, 8e-6 measured) moves the float64 reference's parameter gradients by up to 4.1e-2.  Both
    numbers are measured by the generator on the reference and stored per parameter in the fixture.  The test therefore
    checks the two factors of the chain rule separately at the float32 bar — the MR-STFT backward ALONE on the golden
    decoded audio (<= max(1e-3, 3 x the reference's float32 distance)), the conv-stack backward through the linear
    functional (1e-3) — and the composite at max(1e-3, 3 x reference float32 distance, 4 x reference sensitivity to the
    1e-5 white forward displacement), parameter by parameter (measured worst case on MI355X: 1.6e-2 = 3.0 x that sensitivity).
  * 2 layers of the Stable Audio Open DiT block (d=1536, 24 x 64 heads, GQA 24:12, N=1025, M=130, batch 2): fp32 at
    1e-3 (output, hidden states, loss, every gradient) and bf16 with the bound stated at the assert.
  * depth-24 forward (plain and CFG), fp32 at 1e-3 and bf16 vs the fp32 reference with the bound stated at the assert.

`-m gpu`: the product path (gfx950 library).  `-m "not gpu"`: the ORACLE against the same fixtures (pins the oracle at
full width; the host-side simulator is far too slow for 80 GFLOP).
"""
import math

import numpy as np
import pytest
import torch

import dit_oracle
import seeded
import stft_oracle
import vae_oracle
from gen_golden_full import dit_full_inputs
from golden_util import load_golden, rel_err

TOL = 1e-3


def l2_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _check_grads(g, tag, names, grads, bar):
    """Every parameter: gradient norm, and the stored full tensor / probe, each within bar(name)."""
    worst = ("", 0.0, 0.0)
    bad = []
    for n, gr in zip(names, grads):
        tol = bar(n)
        gn = float(g[f"gnorm_{tag}/{n}"]) if f"gnorm_{tag}/{n}" in g else float(g[f"gnorm/{n}"])
        e_norm = abs(float(gr.double().norm()) - gn) / max(gn, 1e-12)
        key_full = f"grad_{tag}/{n}" if f"grad_{tag}/{n}" in g else f"grad/{n}"
        key_probe = f"probe_{tag}/{n}" if f"probe_{tag}/{n}" in g else f"probe/{n}"
        if key_full in g:
            e = rel_err(gr, g[key_full])
        else:
            idx = torch.from_numpy(seeded.probe_index(n, gr.numel()))
            # probe values relative to the tensor's largest entry: norm * sqrt(#)/sqrt(numel) is its typical magnitude
            e = float((gr.reshape(-1).cpu()[idx].double() - torch.from_numpy(g[key_probe]).double()).abs().max()
                      / torch.from_numpy(g[key_probe]).double().abs().max().clamp_min(1e-30))
        e = max(e, e_norm)
        if e / tol > worst[1]:
            worst = (n, e / tol, e)
        if not e < tol:
            bad.append((n, float(f"{e:.3g}"), float(f"{tol:.3g}")))
    assert not bad, (tag, len(bad), sorted(bad, key=lambda r: -r[1] / r[2])[:12])
    return worst


# ------------------------------------------------------------------------------------------------ VAE, full width
def _vae_inputs(device):
    audio, noise, proj = [torch.from_numpy(a).to(device) for a in seeded.full_vae_inputs()]
    return audio, noise, proj


def _vae_state(shapes):
    return {k: torch.from_numpy(v) for k, v in seeded.seeded_state_dict(shapes, seeded.FULL_VAE["seed"]).items()}


def _vae_bar(g, tag):
    if tag == "lin":
        return lambda n: TOL
    # 4 x the white-noise probe: the native forward deviation (bf16x3 rounding, 8e-6 of the peak) is not white — it is correlated
    # along time, i.e. richer in exactly the low-frequency bins the A-weighted log-magnitude term amplifies (measured: up to 3.0 x)
    # floor: the reference's own worst float32-vs-float64 parameter-gradient distance on this loss (1.47e-3)
    floor = max(float(v) for k, v in g.items() if k.startswith("refdist_gen/"))
    return lambda n: max(TOL, floor, 3.0 * float(g[f"refdist_gen/{n}"]), 4.0 * float(g[f"sens_gen/{n}"]))


def _vae_asserts(g, pre, z, kl, dec, loss_gen, loss_lin):
    assert rel_err(pre, g["pre"]) < TOL
    assert rel_err(z, g["z"]) < TOL
    assert rel_err(kl, g["kl"]) < TOL
    assert rel_err(dec, g["decoded"]) < TOL
    assert abs(float(loss_gen) - float(g["loss_gen_f64"])) < TOL * abs(float(g["loss_gen_f64"]))
    assert abs(float(loss_lin) - float(g["loss_lin"])) < TOL * max(abs(float(g["loss_lin"])), 1.0)


@pytest.mark.gpu
def test_vae_full_width_matches_reference_gpu(hip):
    from stable_audio_tools_amd.auraloss import AutoencoderSpectralLoss
    from stable_audio_tools_amd.autoencoders import create_autoencoder_from_config
    g = load_golden("full_vae")
    cfg = seeded.full_vae_config()
    model = create_autoencoder_from_config(cfg)
    model.load_state_dict(_vae_state({k: tuple(v.shape) for k, v in model.state_dict().items()}))
    model = model.cuda()
    audio, noise, proj = _vae_inputs("cuda")
    spectral = AutoencoderSpectralLoss(44100, weight=1.0, **seeded.STFT_CFG).cuda()
    # the MR-STFT backward alone, at the reference's decoded audio
    dref = torch.from_numpy(g["decoded"]).cuda().requires_grad_(True)
    (gdec,) = torch.autograd.grad(spectral(audio, dref), dref)
    e_dec = rel_err(gdec, g["gdec_f64"])
    assert e_dec < max(TOL, 3.0 * float(g["gdec_refdist"])), (e_dec, float(g["gdec_refdist"]))
    z, info = model.encode(audio, return_info=True, noise=noise)
    dec = model.decode(z)
    loss_gen = spectral(audio, dec) + seeded.FULL_VAE["kl_weight"] * info["kl"]
    loss_lin = (dec * proj).sum() / proj.numel() ** 0.5 + 0.1 * info["kl"]
    _vae_asserts(g, info["pre_bottleneck_latents"].detach(), z.detach(), info["kl"].detach(), dec.detach(), loss_gen.detach(),
                 loss_lin.detach())
    names = [n for n, _ in model.named_parameters()]
    params = list(model.parameters())
    g_lin = torch.autograd.grad(loss_lin, params, retain_graph=True)
    w_lin = _check_grads(g, "lin", names, g_lin, _vae_bar(g, "lin"))
    g_gen = torch.autograd.grad(loss_gen, params)
    w_gen = _check_grads(g, "gen", names, g_gen, _vae_bar(g, "gen"))
    print(f"full-width VAE: decoded {rel_err(dec.detach(), g['decoded']):.2e}; dL/d(decoded) of the MR-STFT loss {e_dec:.2e} (reference f32: {float(g['gdec_refdist']):.2e})")
    print(f"full-width VAE: worst lin grad {w_lin[0]} {w_lin[2]:.2e}; worst gen grad {w_gen[0]} {w_gen[2]:.2e} ({w_gen[1]:.2f} of its bar)")


def test_vae_full_width_oracle_matches_reference():
    """Pins oracle/vae_oracle.py + oracle/stft_oracle.py at the real widths (float32, CPU)."""
    g = load_golden("full_vae")
    cfg = seeded.full_vae_config()
    from stable_audio_tools_amd.autoencoders import create_autoencoder_from_config
    shapes = {k: tuple(v.shape) for k, v in create_autoencoder_from_config(cfg).state_dict().items()}
    sd = {k: v.requires_grad_(True) for k, v in _vae_state(shapes).items()}
    audio, noise, proj = _vae_inputs("cpu")
    z, kl, pre = vae_oracle.autoencoder_encode(sd, cfg["model"], audio, noise)
    dec = vae_oracle.autoencoder_decode(sd, cfg["model"], z)
    dref = torch.from_numpy(g["decoded"]).requires_grad_(True)
    (gdec,) = torch.autograd.grad(stft_oracle.autoencoder_spectral_loss(audio, dref, seeded.STFT_CFG, 44100), dref)
    assert rel_err(gdec, g["gdec_f64"]) < max(TOL, 3.0 * float(g["gdec_refdist"]))
    loss_gen = stft_oracle.autoencoder_spectral_loss(audio, dec, seeded.STFT_CFG, 44100) + seeded.FULL_VAE["kl_weight"] * kl
    loss_lin = (dec * proj).sum() / proj.numel() ** 0.5 + 0.1 * kl
    _vae_asserts(g, pre.detach(), z.detach(), kl.detach(), dec.detach(), loss_gen.detach(), loss_lin.detach())
    names = list(sd.keys())
    g_lin = torch.autograd.grad(loss_lin, [sd[n] for n in names], retain_graph=True)
    _check_grads(g, "lin", names, g_lin, _vae_bar(g, "lin"))
    g_gen = torch.autograd.grad(loss_gen, [sd[n] for n in names])
    _check_grads(g, "gen", names, g_gen, _vae_bar(g, "gen"))


# ------------------------------------------------------------------------------------------------ DiT, full width
def _dit_state(model):
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith("inv_freq")}
    return {k: torch.from_numpy(v) for k, v in seeded.seeded_state_dict(shapes, seeded.FULL_DIT["seed"]).items()}


def _build_dit(depth, device, dtype):
    from stable_audio_tools_amd.dit import DiffusionTransformer
    model = DiffusionTransformer(**dict(seeded.FULL_DIT["config"], depth=depth))
    missing, unexpected = model.load_state_dict(_dit_state(model), strict=False)
    assert not unexpected and all(k.endswith("inv_freq") for k in missing)
    return model.to(device=device, dtype=dtype)


def _dit2_run(model, inp, device, dtype):
    xin = inp["noised"].to(device, dtype).requires_grad_(True)
    out, info = model(xin, inp["t"].to(device, dtype), cross_attn_cond=inp["cross"].to(device, dtype),
                      global_embed=inp["glob"].to(device, dtype), return_info=True)
    loss = torch.nn.functional.mse_loss(out.float(), inp["target"].to(device))
    names = [n for n, _ in model.named_parameters()]
    grads = torch.autograd.grad(loss, [xin] + list(model.parameters()))
    return out, info, loss, names, grads


@pytest.mark.gpu
def test_dit_block_full_width_fp32_gpu(hip):
    g = load_golden("full_dit2")
    model = _build_dit(2, "cuda", torch.float32).train(True)
    out, info, loss, names, grads = _dit2_run(model, dit_full_inputs(2), "cuda", torch.float32)
    assert rel_err(out.detach(), g["out"]) < TOL
    assert rel_err(info["hidden_states"][0].detach()[:, ::16], g["hidden_first"]) < TOL
    assert rel_err(info["hidden_states"][-1].detach()[:, ::16], g["hidden_last"]) < TOL
    assert abs(float(loss) - float(g["loss"])) < TOL * float(g["loss"])
    assert rel_err(grads[0], g["grad/<input>"]) < TOL
    w = _check_grads(g, "", names, grads[1:], lambda n: TOL)
    print(f"full-width DiT block fp32: worst gradient {w[0]} {w[2]:.2e}")


# bf16 bounds (stated here, as the 1e-3 bar of BASELINE.json is a float32 bar): every tensor is stored with an 8-bit
# mantissa (unit round-off 2^-9 = 2e-3) and a token passes ~25 such roundings through two layers; relative L2 distance
# to the float32 reference is dominated by the bf16 rounding of dy in the wgrad GEMMs.  Bars = 2 x the measured distances on MI355X
# (forward 7.7e-3, gradients 1.8e-2).
BF16_FWD, BF16_GRAD = 1.6e-2, 3.6e-2


@pytest.mark.gpu
def test_dit_block_full_width_bf16_gpu(hip):
    g = load_golden("full_dit2")
    model = _build_dit(2, "cuda", torch.bfloat16).train(True)
    out, info, loss, names, grads = _dit2_run(model, dit_full_inputs(2), "cuda", torch.bfloat16)
    assert out.dtype == torch.bfloat16
    e_out = l2_err(out.detach().float(), g["out"])
    assert e_out < BF16_FWD, e_out
    assert abs(float(loss) - float(g["loss"])) < BF16_FWD * float(g["loss"])
    e_in = l2_err(grads[0].float(), g["grad/<input>"])
    assert e_in < BF16_GRAD, e_in
    worst = ("", 0.0)
    for n, gr in zip(names, grads[1:]):
        gn = float(g["gnorm/" + n])
        e = abs(float(gr.double().norm()) - gn) / gn
        if ("grad/" + n) in g:
            e = max(e, l2_err(gr.float(), g["grad/" + n]))
        else:
            idx = torch.from_numpy(seeded.probe_index(n, gr.numel()))
            e = max(e, l2_err(gr.reshape(-1).cpu()[idx].float(), g["probe/" + n]))
        if e > worst[1]:
            worst = (n, e)
        assert e < BF16_GRAD, (n, e)
    print(f"full-width DiT block bf16: out {e_out:.2e}, d/dx {e_in:.2e}, worst parameter gradient {worst[0]} {worst[1]:.2e}")


def test_dit_block_full_width_oracle_matches_reference():
    g = load_golden("full_dit2")
    from stable_audio_tools_amd.dit import DiffusionTransformer
    cfg = dict(seeded.FULL_DIT["config"], depth=2)
    model = DiffusionTransformer(**cfg)
    names = [n for n, _ in model.named_parameters()]
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd.update(_dit_state(model))
    for n in names:
        sd[n].requires_grad_(True)
    inp = dit_full_inputs(2)
    xin = inp["noised"].clone().requires_grad_(True)
    out = dit_oracle.dit_forward(sd, cfg, xin, inp["t"], inp["cross"], inp["glob"])
    loss = torch.nn.functional.mse_loss(out, inp["target"])
    grads = torch.autograd.grad(loss, [xin] + [sd[n] for n in names])
    assert rel_err(out.detach(), g["out"]) < 1e-4
    assert rel_err(grads[0], g["grad/<input>"]) < TOL
    _check_grads(g, "", names, grads[1:], lambda n: TOL)


# depth 24: the float32 model is held to 1e-3; the bf16 model to a relative L2 distance of 2.8e-2 (2 x measured) from the float32 reference
# (24 layers x ~12 bf16 roundings of the residual stream each, unit round-off 2e-3, accumulating like a random walk:
# 2e-3 * sqrt(288) = 3.4e-2 is the expectation if every rounding hit the full stream; measured 1.4e-2).  CFG at scale 6 amplifies the difference of two such outputs,
# so the guided output is compared in float32 only.
BF16_DEPTH24 = 2.8e-2


@pytest.mark.gpu
def test_dit_depth24_forward_gpu(hip):
    g = load_golden("full_dit24")
    inp = dit_full_inputs(1)
    model = _build_dit(24, "cuda", torch.float32).train(False)
    kw = dict(cross_attn_cond=inp["cross"].cuda(), global_embed=inp["glob"].cuda())
    with torch.no_grad():
        plain = model(inp["noised"].cuda(), inp["t"].cuda(), cfg_scale=1.0, **kw)
        guided = model(inp["noised"].cuda(), inp["t"].cuda(), cfg_scale=6.0, scale_phi=0.75, **kw)
    e32, eg = rel_err(plain, g["plain"]), rel_err(guided, g["guided"])
    assert e32 < TOL and eg < TOL, (e32, eg)
    model = model.to(torch.bfloat16)
    with torch.no_grad():
        pb = model(inp["noised"].cuda().bfloat16(), inp["t"].cuda().bfloat16(), cfg_scale=1.0,
                   cross_attn_cond=kw["cross_attn_cond"].bfloat16(), global_embed=kw["global_embed"].bfloat16())
    eb = l2_err(pb.float(), g["plain"])
    print(f"depth-24 DiT: fp32 plain {e32:.2e} guided {eg:.2e}; bf16 plain (relative L2) {eb:.2e}")
    assert eb < BF16_DEPTH24, eb


def test_dit_depth24_oracle_matches_reference():
    g = load_golden("full_dit24")
    from stable_audio_tools_amd.dit import DiffusionTransformer
    cfg = seeded.FULL_DIT["config"]
    with torch.device("meta"):
        shapes = {k: tuple(v.shape) for k, v in DiffusionTransformer(**cfg).state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in seeded.seeded_state_dict({k: s for k, s in shapes.items() if not k.endswith("inv_freq")},
                                                                      seeded.FULL_DIT["seed"]).items()}
    half = 32
    sd["transformer.rotary_pos_emb.inv_freq"] = 1.0 / (10000 ** (torch.arange(0, half, 2).float() / half))
    inp = dit_full_inputs(1)
    with torch.no_grad():
        plain = dit_oracle.dit_forward(sd, cfg, inp["noised"], inp["t"], inp["cross"], inp["glob"])
    assert rel_err(plain, g["plain"]) < 2e-4
