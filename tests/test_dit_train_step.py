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


def _autocast_steps(device, name, idx, nsteps):
    """bf16-mixed (fp32 master weights, bf16 activations / kernels, Lightning '--precision bf16-mixed'): finite loss that
    decreases on a fixed batch.  Runs both global-conditioning modes: under autocast the adaLN modulation
    (fp32 parameter + bf16 embedding) must be cast to the activation dtype (transformer.py:677)."""
    from stable_audio_tools_amd.training import DiTTrainStep
    model, _ = _build(name, 700 + 10 * idx, device)
    model.train(True)
    stepper = DiTTrainStep(model, lr=1e-4, cfg_dropout_prob=0.0, autocast_dtype=torch.bfloat16)
    inp = {k: v.to(device) for k, v in dit_inputs(name).items()}
    noise = torch.from_numpy(seeded.seeded_array(tuple(inp["x"].shape), 4321)).to(device)
    losses = [float(stepper(inp["x"], cross_attn_cond=inp["cross_attn_cond"], global_embed=inp["global_embed"], t=inp["t"], noise=noise)["loss"])
              for _ in range(nsteps)]
    assert all(math.isfinite(v) for v in losses) and min(losses[-3:]) < losses[0], losses


@pytest.mark.gpu
@pytest.mark.parametrize("idx,name", [(0, "tiny_prepend"), (1, "tiny_adaln")])
def test_dit_train_step_bf16_autocast_gpu(hip, idx, name):
    _autocast_steps("cuda", name, idx, 10)


@pytest.mark.parametrize("idx,name", [(0, "tiny_prepend"), (1, "tiny_adaln")])
def test_dit_train_step_bf16_autocast_simulator(emu_modules, idx, name):
    _autocast_steps("cpu", name, idx, 4)


# ---- N > 1: world_size-2 gloo processes on CPU (kernels on the simulator) — BASELINE.json configs[3] (DiT training, DDP) ----
def _dit_ddp_worker(rank, world, port, q):
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for pth in (os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), here):
        if pth not in sys.path:
            sys.path.insert(0, pth)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from emu_util import use_emu_ops
    from stable_audio_tools_amd.training import DiTTrainStep
    use_emu_ops()
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model, _ = _build("tiny_adaln", 710, "cpu")
        model.train(True)
        stepper = DiTTrainStep(model, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-3, cfg_dropout_prob=0.0, use_ema=True)
        inp = dit_inputs("tiny_adaln")
        x0 = torch.from_numpy(seeded.seeded_array(tuple(inp["x"].shape), 1000))
        noise = torch.from_numpy(seeded.seeded_array(tuple(inp["x"].shape), 2000))
        t = torch.tensor([0.3, 0.8])
        sl = slice(rank, rank + 1)                               # global batch of 2, one item per rank
        stepper(x0[sl], cross_attn_cond=inp["cross_attn_cond"][sl], global_embed=inp["global_embed"][sl], t=t[sl], noise=noise[sl])
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        avg_grad = (stepper.flat.grad[:stepper.flat.numel] * stepper.comm.grad_scale).clone()     # what the optimizer consumed
        q.put((rank, ([g.numpy() for g in gathered], avg_grad.numpy()) if rank == 0 else None))
    finally:
        dist.destroy_process_group()


def test_dit_data_parallel_step_gloo_world2(emu_modules):
    """Every rank ends the step with identical parameters, equal to a single-process step on the concatenated batch
    (the v-objective MSE is a per-item mean, so the mean of per-rank gradients is the batch gradient)."""
    import os

    import torch.multiprocessing as mp
    from stable_audio_tools_amd.training import DiTTrainStep
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29000 + (os.getpid() % 400)
    procs = [ctx.Process(target=_dit_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = (torch.from_numpy(a) for a in results[0][0])
    avg_grad = torch.from_numpy(results[0][1])
    assert torch.equal(r0, r1), "ranks diverged after the all-reduced step"
    model, sd = _build("tiny_adaln", 710, "cpu")
    model.train(True)
    init = torch.cat([p.detach().reshape(-1).clone() for p in model.parameters()])
    stepper = DiTTrainStep(model, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-3, cfg_dropout_prob=0.0, use_ema=True)
    inp = dit_inputs("tiny_adaln")
    x0 = torch.from_numpy(seeded.seeded_array(tuple(inp["x"].shape), 1000))
    noise = torch.from_numpy(seeded.seeded_array(tuple(inp["x"].shape), 2000))
    stepper(x0, cross_attn_cond=inp["cross_attn_cond"], global_embed=inp["global_embed"], t=torch.tensor([0.3, 0.8]), noise=noise)
    single = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    # the AVERAGED gradient equals the single-process batch gradient (summation order only) — Adam's normalisation
    # would hide a mis-scaled bucket, so this is checked on the gradient itself, and the update after it
    batch_grad = stepper.flat.grad[:stepper.flat.numel]
    assert float((avg_grad - batch_grad).norm() / batch_grad.norm()) < 1e-5
    assert float((avg_grad - batch_grad).abs().max() / batch_grad.abs().max()) < 1e-5
    upd_ddp, upd_single = r0 - init, single - init
    assert float((upd_ddp - upd_single).norm() / upd_single.norm()) < 1e-3
