"""Helpers shared by the CPU (simulator) and GPU parity tests.  TEST INFRASTRUCTURE ONLY."""
import os

import numpy as np
import torch

import seeded

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def rel_err(a, b):
    """max |a-b| / max |b|  — the 1e-3 'rel fp32' bar of BASELINE.json is applied to this."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def ae_state_dict(name, seed, shapes):
    sd = seeded.seeded_state_dict(shapes, seed)
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def build_native_ae(name, seed, device="cpu"):
    """Product AudioAutoencoder for AE_CONFIGS[name] with the seeded reference-format state_dict."""
    from stable_audio_tools_amd.autoencoders import create_autoencoder_from_config
    model = create_autoencoder_from_config(seeded.AE_CONFIGS[name])
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(ae_state_dict(name, seed, shapes))
    return model.to(device)


def dit_trajectory_distances(model_lowp, dcfg, noise, cond_kw, steps=10, variants=("bf16",)):
    """N-step v-DDIM trajectories (inference/sampling.py:254-307, eta 0, CFG as in `cond_kw`) of ONE set of 16-bit-rounded weights in
    several arithmetic modes, from the same initial noise: relative L2 distance of the FINAL latents to the float32 trajectory.
    `model_lowp`: the native bf16 model as configured by the caller (e.g. fp8 projections on) — its trajectory is 'lowp'; `variants`
    adds 'bf16' (a bf16 copy with fp8 off).  The float32 trajectory runs on the native fp32 path (bf16x3 products — held to the
    reference's own fp32 DiffusionTransformer at depth 24 by tests/test_full_width.py::test_dit_depth24_forward_gpu at 1e-3, measured
    3e-6): the reference's CPU path would take ~35 s per evaluation at N = 6145.  Returns {'lowp': e, 'bf16': e, 'final_norm': ...}."""
    from stable_audio_tools_amd.dit import DiffusionTransformer
    from stable_audio_tools_amd.sampling import sample_v_ddim
    dev = noise.device
    sd = {k: v.detach().float() for k, v in model_lowp.state_dict().items()}
    out = {}
    with torch.no_grad():
        m32 = DiffusionTransformer(**dcfg).to(dev).train(False)
        m32.load_state_dict(sd, strict=False)
        kw32 = {k: (v.float() if torch.is_tensor(v) else v) for k, v in cond_kw.items()}
        x32 = sample_v_ddim(m32, noise.float(), steps, **kw32).float()
        del m32
        torch.cuda.empty_cache()
        xl = sample_v_ddim(model_lowp, noise, steps, **cond_kw).float()
        out["lowp"] = float((xl - x32).norm() / x32.norm())
        if "bf16" in variants:
            mb = DiffusionTransformer(**dcfg).to(dev, torch.bfloat16).train(False)      # a fresh module: no fp8 switches, no cached operands
            mb.load_state_dict(model_lowp.state_dict(), strict=False)
            xb = sample_v_ddim(mb, noise, steps, **cond_kw).float()
            out["bf16"] = float((xb - x32).norm() / x32.norm())
            del mb
            torch.cuda.empty_cache()
    out["final_norm"] = float(x32.norm())
    out["finite"] = bool(torch.isfinite(xl).all())
    return out
