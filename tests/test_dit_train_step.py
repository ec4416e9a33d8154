"""DiT optimisation step (stable_audio_tools_amd.training.DiTTrainStep) against a plain-PyTorch restatement:
oracle forward (oracle/dit_oracle.py) + autograd + torch.optim.AdamW, with the timesteps and the noise injected so
both sides see identical data (training/diffusion.py:381-449 draws them from RNGs)."""
import math

import pytest
import torch

import dit_oracle
import seeded
from gen_golden import dit_inputs
from test_dit_parity import _build


def _steps(device, name="tiny_adaln", idx=1, nsteps=2):
    from stable_audio_tools_amd.training import DiTTrainStep
    model, sd = _build(name, 700 + 10 * idx, device)
    model.train(True)
    stepper = DiTTrainStep(model, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-3, cfg_dropout_prob=0.0, use_ema=True)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not k.endswith("inv_freq") and not k.endswith(".beta")}
    frozen = {k: v for k, v in sd.items() if k not in params}
    names = [n for n, _ in model.named_parameters()]
    opt = torch.optim.AdamW([params[n] for n in names], lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-3)
    inp = dit_inputs(name)
    cfg = seeded.DIT_CONFIGS[name]
    for s in range(nsteps):
        x0 = torch.from_numpy(seeded.seeded_array(tuple(inp["x"].shape), 1000 + s))
        noise = torch.from_numpy(seeded.seeded_array(tuple(inp["x"].shape), 2000 + s))
        t = torch.tensor([0.3 + 0.1 * s, 0.8 - 0.1 * s])
        out = stepper(x0.to(device), cross_attn_cond=inp["cross_attn_cond"].to(device), global_embed=inp["global_embed"].to(device),
                      t=t.to(device), noise=noise.to(device))
        al, si = torch.cos(t * math.pi / 2)[:, None, None], torch.sin(t * math.pi / 2)[:, None, None]
        full = dict(frozen, **params)
        o = dit_oracle.dit_forward(full, cfg, x0 * al + noise * si, t, inp["cross_attn_cond"], inp["global_embed"])
        loss = torch.nn.functional.mse_loss(o, noise * al - x0 * si)
        opt.zero_grad()
        loss.backward()
        opt.step()
        assert abs(float(out["loss"]) - float(loss.detach())) <= 1e-3 * abs(float(loss.detach())), (s, float(out["loss"]), float(loss.detach()))
    worst = 0.0
    now = dict(model.named_parameters())
    for n in names:
        upd = now[n].detach().cpu() - sd[n]
        upd_ref = params[n].detach() - sd[n]
        worst = max(worst, float((upd - upd_ref).norm() / upd_ref.norm().clamp_min(1e-12)))
    assert worst < 2e-2, worst


def test_dit_train_step_matches_oracle_simulator(emu_modules):
    _steps("cpu")


@pytest.mark.gpu
def test_dit_train_step_matches_oracle_gpu(hip):
    _steps("cuda")


@pytest.mark.gpu
def test_dit_train_step_bf16_autocast_gpu(hip):
    """bf16-mixed (fp32 master weights, bf16 activations/kernels): finite loss that decreases on a fixed batch."""
    from stable_audio_tools_amd.training import DiTTrainStep
    model, _ = _build("tiny_prepend", 700, "cuda")
    model.train(True)
    stepper = DiTTrainStep(model, lr=1e-4, cfg_dropout_prob=0.0, autocast_dtype=torch.bfloat16)
    inp = {k: v.cuda() for k, v in dit_inputs("tiny_prepend").items()}
    noise = torch.randn_like(inp["x"])
    losses = [float(stepper(inp["x"], cross_attn_cond=inp["cross_attn_cond"], global_embed=inp["global_embed"], t=inp["t"], noise=noise)["loss"])
              for _ in range(10)]
    assert all(math.isfinite(v) for v in losses) and min(losses[-3:]) < losses[0], losses
