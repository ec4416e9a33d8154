"""The autoencoder generator step (stable_audio_tools_amd.training.AutoencoderTrainStep) against a
plain-PyTorch restatement built from the oracle: oracle forward (vae_oracle + stft_oracle), torch
autograd, torch.optim.AdamW — i.e. what the reference's Lightning wrapper computes
(training/autoencoders.py:367-527, generator branch, no discriminator).

Also covers the N>1 path with world_size-2 gloo processes on CPU (kernels on the simulator): every
rank must end the step with identical parameters, equal to a single-process step on the
concatenated batch (SURVEY.md §4 test plan item 4)."""
import copy
import os
import sys

import pytest
import torch

import seeded
import stft_oracle
import vae_oracle
from golden_util import build_native_ae, rel_err

NAME, SEED = "tiny", 100


def _model_config():
    cfg = copy.deepcopy(seeded.AE_CONFIGS[NAME])
    cfg["training"] = {
        "learning_rate": 1e-3, "use_ema": True,
        "optimizer_configs": {"autoencoder": {
            # eps is raised from torch's 1e-8 so that the first Adam steps (update ~ g / (|g| + eps)) stay a smooth
            # function of the gradient: with eps -> 0 the update is sign(g) and a 1e-5-level difference on a
            # near-zero gradient element flips a full +-lr step, which says nothing about parity
            "optimizer": {"type": "AdamW", "config": {"betas": [0.8, 0.99], "lr": 1e-3, "weight_decay": 1e-3, "eps": 1e-3}},
            # warmup 0.9 (the shipped config uses 0.999): lr_eff = 1e-4 / 1.9e-4 on the two steps, so the updates are
            # well above the fp32 resolution of the parameters they are subtracted from
            "scheduler": {"type": "InverseLR", "config": {"inv_gamma": 200000, "power": 0.5, "warmup": 0.9}}}},
        "loss_configs": {"spectral": {"type": "mrstft", "config": {"fft_sizes": [256, 128, 64, 32], "hop_sizes": [64, 32, 16, 8],
                                                                   "win_lengths": [256, 128, 64, 32], "perceptual_weighting": True},
                                      "weights": {"mrstft": 1.0}},
                         "bottleneck": {"type": "kl", "weights": {"kl": 1e-4}}},
    }
    return cfg


def _batch(batch, seed, device="cpu"):
    audio = torch.from_numpy(seeded.seeded_array((batch, 2, 512), seed, scale=0.3)).to(device)
    noise = torch.from_numpy(seeded.seeded_array((batch, 4, 64), seed + 1)).to(device)
    return audio, noise


def _oracle_steps(cfg, batches, lr_fn):
    """Plain-PyTorch restatement of N generator steps on CPU; returns final state_dict."""
    from stable_audio_tools_amd.autoencoders import create_autoencoder_from_config
    shapes = {k: tuple(v.shape) for k, v in create_autoencoder_from_config(cfg).state_dict().items()}
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in seeded.seeded_state_dict(shapes, SEED).items()}
    oc = cfg["training"]["optimizer_configs"]["autoencoder"]["optimizer"]["config"]
    opt = torch.optim.AdamW(list(sd.values()), lr=oc["lr"], betas=tuple(oc["betas"]), weight_decay=oc["weight_decay"], eps=oc["eps"])
    sc = cfg["training"]["loss_configs"]["spectral"]["config"]
    losses = []
    for step, (audio, noise) in enumerate(batches):
        for gparam in opt.param_groups:
            gparam["lr"] = lr_fn(step)
        z, kl, _ = vae_oracle.autoencoder_encode(sd, cfg["model"], audio, noise)
        dec = vae_oracle.autoencoder_decode(sd, cfg["model"], z)
        loss = stft_oracle.autoencoder_spectral_loss(audio, dec, sc, cfg["sample_rate"]) + 1e-4 * kl
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return {k: v.detach() for k, v in sd.items()}, losses


def _native_steps(cfg, batches, device, **step_kw):
    from stable_audio_tools_amd.training import AutoencoderTrainStep
    model = build_native_ae(NAME, SEED, device)
    stepper = AutoencoderTrainStep(model, cfg, **step_kw)
    losses = []
    for audio, noise in batches:
        out = stepper(audio.to(device), noise=noise.to(device))
        losses.append(float(out["loss"]))
    return model, stepper, losses


def _check_against_oracle(device, ops, bf16x3):
    """bf16x3=False: the fp32-MFMA conv kernels — per-parameter updates must match the oracle to 2e-2.
    bf16x3=True (the default product path for the k7 convs): its ~1e-5 forward differences are amplified by the
    DISCONTINUOUS gradient of the L1-log-magnitude STFT term (sign(log|X| - log|Y|) flips on bins where the two
    spectra nearly coincide — decoded ~ reals in this test), so individual small parameters can move differently;
    the losses must still agree to 1e-3 and the whole update vector must point the same way (cosine >= 0.995)."""
    from stable_audio_tools_amd.training import inverse_lr
    prev = ops.use_bf16x3
    ops.use_bf16x3 = bf16x3
    try:
        _check_body(device, bf16x3, inverse_lr)
    finally:
        ops.use_bf16x3 = prev


def _check_body(device, bf16x3, inverse_lr):
    cfg = _model_config()
    batches = [_batch(2, 900), _batch(2, 910)]
    sch = cfg["training"]["optimizer_configs"]["autoencoder"]["scheduler"]["config"]
    ref_sd, ref_losses = _oracle_steps(cfg, batches, lambda s: inverse_lr(s, 1e-3, **sch))
    model, stepper, losses = _native_steps(cfg, batches, device)
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 1e-3 * abs(b), (losses, ref_losses)
    sd = model.state_dict()
    # after two AdamW steps each parameter moved by ~2*lr; compare the UPDATE, not the parameter
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    init = {k: torch.from_numpy(v) for k, v in seeded.seeded_state_dict(shapes, SEED).items()}
    worst = 0.0
    ups, refs = [], []
    for k in sd:
        upd = sd[k].detach().cpu() - init[k]
        upd_ref = ref_sd[k] - init[k]
        ups.append(upd.reshape(-1))
        refs.append(upd_ref.reshape(-1))
        worst = max(worst, float((upd - upd_ref).norm() / upd_ref.norm().clamp_min(1e-12)))
    cos = float(torch.nn.functional.cosine_similarity(torch.cat(ups), torch.cat(refs), dim=0))
    assert cos >= 0.995, cos
    if not bf16x3:
        # per-parameter bound for the fp32-MFMA path; 1e-7-level forward differences (e.g. a different sin()
        # implementation) already move the worst small parameter by 1-3 % through the STFT term's sign flips
        assert worst < 1e-1 and cos >= 0.999, (worst, cos)
    assert stepper.opt.ema is not None and stepper.global_step == 2
    assert bool(torch.isfinite(stepper.opt.ema).all())


@pytest.mark.parametrize("bf16x3", [True])        # (the fp32-MFMA fallback mode runs on the GPU: test_generator_step_matches_oracle_gpu)
def test_generator_step_matches_oracle_simulator(emu_modules, bf16x3):
    _check_against_oracle("cpu", emu_modules, bf16x3)


@pytest.mark.gpu
@pytest.mark.parametrize("bf16x3", [False, True])
def test_generator_step_matches_oracle_gpu(hip, bf16x3):
    _check_against_oracle("cuda", hip, bf16x3)


def _ddp_worker(rank, world, port, q, ddp_mode="all_reduce", overlap=True, comm_dtype=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from emu_util import emu_ops, use_emu_ops
    use_emu_ops()
    sim = emu_ops()
    folds = []                                   # sizes of the gradient folds (one sat_multi_copy call each)
    plain_multi_copy = sim.multi_copy
    sim.multi_copy = lambda srcs, dsts: (folds.append(len(srcs)), plain_multi_copy(srcs, dsts))[1]
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = _model_config()
        audio, noise = _batch(2, 950)           # global batch of 2, one item per rank
        _, stepper, _ = _native_steps(cfg, [(audio[rank:rank + 1], noise[rank:rank + 1])], "cpu", ddp_mode=ddp_mode, ddp_overlap=overlap,
                                      bucket_bytes=4096,          # tiny buckets: many of them fire from the hooks mid-backward
                                      ddp_comm_dtype={None: None, "bf16": torch.bfloat16}[comm_dtype])
        assert len(stepper.comm.buckets) > 4
        # adopted gradients reach the flat buffer in batches: at most one fold per bucket (+ the final gather) when the hooks exchange,
        # ONE gather otherwise — never a copy per parameter — and every parameter's gradient ends up as its flat view
        nparams = len(stepper.flat.params)
        assert 1 <= len(folds) <= (len(stepper.comm.buckets) + 1 if overlap else 1) and sum(folds) <= nparams, (folds, nparams)
        assert max(folds) > 1 and all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(stepper.flat.params, stepper.flat._views))
        log = stepper.comm.last_launch_log
        assert sorted(b for b, _ in log) == list(range(len(stepper.comm.buckets)))                 # every bucket exactly once
        assert (sum(h for _, h in log) >= len(log) - 1) if overlap else not any(h for _, h in log)   # ... from the hooks when overlapped
        flat = stepper.flat.data.clone()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        avg_grad = (stepper.flat.grad * stepper.comm.grad_scale).clone()        # the averaged gradient the optimizer consumed
        q.put((rank, ([g.numpy() for g in gathered], avg_grad.numpy()) if rank == 0 else None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ddp_mode,overlap,comm_dtype", [("all_reduce", True, None), ("reduce_scatter", True, None), ("all_reduce", False, None),
                                                         ("all_reduce", True, "bf16")])
def test_data_parallel_step_gloo_world2(emu_modules, ddp_mode, overlap, comm_dtype):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + 7 * ["all_reduce", "reduce_scatter"].index(ddp_mode) + int(overlap) + 3 * int(comm_dtype is not None)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q, ddp_mode, overlap, comm_dtype)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = (torch.from_numpy(a) for a in results[0][0])
    avg_grad = torch.from_numpy(results[0][1])
    assert torch.equal(r0, r1), "ranks diverged after the all-reduced step"
    # single process, full batch: mean-of-per-rank-gradients == gradient of the batch-mean loss only if the
    # loss is a per-item mean; the spectral-convergence and KL terms are, the log-magnitude term is too.
    cfg = _model_config()
    audio, noise = _batch(2, 950)
    _, stepper, _ = _native_steps(cfg, [(audio, noise)], "cpu")
    # the AVERAGED gradient is the single-process batch gradient (fp32 summation order only; a bf16 exchange
    # rounds each rank's addend to 8 mantissa bits) — checked on the gradient itself: Adam's normalisation would hide a
    # mis-scaled bucket
    batch_grad = stepper.flat.grad
    tol = 1e-5 if comm_dtype is None else 8e-3
    assert float((avg_grad - batch_grad).norm() / batch_grad.norm()) < tol
    assert float((avg_grad - batch_grad).abs().max() / batch_grad.abs().max()) < tol
    single = stepper.flat.data
    shapes_init = build_native_ae(NAME, SEED)
    init = torch.cat([p.detach().reshape(-1) for p in shapes_init.parameters()])
    n = init.numel()
    upd_ddp = r0[:n] - init
    upd_single = single[:n] - init
    assert float((upd_ddp - upd_single).norm() / upd_single.norm()) < (1e-3 if comm_dtype is None else 5e-2)


def _nccl_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        cfg = _model_config()
        audio, noise = _batch(2, 950)
        _, stepper, _ = _native_steps(cfg, [(audio[rank:rank + 1], noise[rank:rank + 1])], f"cuda:{rank}", ddp_mode="reduce_scatter",
                                      bucket_bytes=4096)
        flat = stepper.flat.data.clone()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        q.put((rank, bool(torch.equal(gathered[0], gathered[1]))))
    finally:
        dist.destroy_process_group()


def _single_rank_worker(port, q):
    """ONE rank, backend nccl (= RCCL), the exchange forced on: hooks, side stream, per-bucket events, in-place reduce-scatter +
    all-gather / all_reduce / the bf16 exchange all execute on the GPU; with one rank every collective is the identity, so each mode
    must reproduce the no-process-group step bit for bit (fp32) / to bf16 rounding of the gradient (bf16 exchange)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    cfg = _model_config()
    batches = [_batch(2, 950), _batch(2, 960)]
    _, base, base_losses = _native_steps(cfg, batches, "cuda:0")            # no process group: the exchange is skipped
    assert not base.comm.active
    ref_data, ref_grad = base.flat.data.clone(), base.flat.grad.clone()
    dist.init_process_group("nccl", rank=0, world_size=1)
    out = {}
    try:
        for mode, overlap, cdt in (("all_reduce", True, None), ("reduce_scatter", True, None), ("all_reduce", False, None),
                                   ("reduce_scatter", True, torch.bfloat16)):
            _, st, losses = _native_steps(cfg, batches, "cuda:0", ddp_mode=mode, ddp_overlap=overlap, bucket_bytes=4096,
                                          ddp_comm_dtype=cdt, ddp_single_rank=True)
            assert st.comm.active and st.comm.backend == "nccl" and len(st.comm.buckets) > 4
            log = st.comm.last_launch_log
            torch.cuda.synchronize()
            out[(mode, overlap, str(cdt))] = {
                "every_bucket_once": sorted(b for b, _ in log) == list(range(len(st.comm.buckets))),
                "from_hooks": sum(h for _, h in log), "buckets": len(log), "side_stream": st.comm._side is not None,
                "data_equal": bool(torch.equal(st.flat.data, ref_data)), "grad_equal": bool(torch.equal(st.flat.grad, ref_grad)),
                "grad_rel": float((st.flat.grad - ref_grad).norm() / ref_grad.norm()),
                "loss_equal": losses == base_losses}
            st.comm.close()
    finally:
        dist.destroy_process_group()
    q.put(out)


@pytest.mark.gpu
def test_single_rank_rccl_exchange_gpu(hip):
    """P1 on hardware that has ONE GPU: the overlapped gradient exchange on a 1-rank RCCL communicator (the world-2 twin below
    needs two devices and is skipped on the 1-GPU box; the gloo twins above cover world 2 on CPU)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_rank_worker, args=(29900 + (os.getpid() % 90), q))
    p.start()
    out = q.get(timeout=600)
    p.join(timeout=60)
    assert p.exitcode == 0
    for key, r in out.items():
        mode, overlap, cdt = key
        assert r["every_bucket_once"], (key, r)
        assert r["side_stream"] == overlap and (r["from_hooks"] >= r["buckets"] - 1 if overlap else r["from_hooks"] == 0), (key, r)
        if cdt == "None":
            assert r["grad_equal"] and r["data_equal"] and r["loss_equal"], (key, r)      # identity collectives: bit-equal to no DDP
        else:
            assert 0 < r["grad_rel"] < 8e-3, (key, r)                                       # bf16 round trip of the gradient buckets


# ---- the native exchange (csrc/comm.hip: sat_allreduce_* — RCCL behind the C-ABI; GradAllReduce(native=True)) ----
def test_native_exchange_entry_points_host():
    """What can be checked without a GPU: RCCL is found with dlopen (PyTorch's copy is already in the process), rank 0's unique ids are
    128 bytes and fresh, every entry point validates its arguments, joining a communicator without a device fails with RCCL's message —
    never a crash — and finalising a null handle is a no-op."""
    import ctypes
    from stable_audio_tools_amd import _lib
    from stable_audio_tools_amd.ops import SatOps
    o = SatOps(_lib.bind(ctypes.CDLL(_lib.LIB_PATH)))
    assert o.allreduce_available()
    a, b = o.allreduce_unique_id(), o.allreduce_unique_id()
    assert len(a) == 128 and len(b) == 128 and a != b and any(a)
    with pytest.raises(ValueError):
        o.allreduce_init(a[:64], 1, 0)
    with pytest.raises(RuntimeError, match="bad arguments"):
        o.allreduce_init(a, 2, 2)
    with pytest.raises(RuntimeError, match="bad arguments"):
        o._chk(o.lib.sat_allreduce_bucket(None, None, 0, 0, 0, None))
    o.allreduce_finalize(None)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="RCCL error"):
            o.allreduce_init(a, 1, 0)


def test_native_exchange_refused_on_the_simulator(emu):
    assert not emu.allreduce_available()
    with pytest.raises(RuntimeError, match="no RCCL"):
        emu.allreduce_unique_id()


def _native_single_rank_worker(port, q):
    """_single_rank_worker with the buckets on the C-ABI's own communicator (GradAllReduce(native=True)): the 1-rank collectives are the
    identity, so every mode must reproduce the no-process-group step bit for bit (fp32) / to bf16 rounding (bf16 exchange)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    cfg = _model_config()
    batches = [_batch(2, 950), _batch(2, 960)]
    _, base, base_losses = _native_steps(cfg, batches, "cuda:0")
    ref_data, ref_grad = base.flat.data.clone(), base.flat.grad.clone()
    dist.init_process_group("nccl", rank=0, world_size=1)
    out = {}
    try:
        for mode, overlap, cdt in (("all_reduce", True, None), ("reduce_scatter", True, None), ("reduce_scatter", False, torch.bfloat16)):
            _, st, losses = _native_steps(cfg, batches, "cuda:0", ddp_mode=mode, ddp_overlap=overlap, bucket_bytes=4096,
                                          ddp_comm_dtype=cdt, ddp_single_rank=True, ddp_native=True)
            torch.cuda.synchronize()
            out[(mode, overlap, str(cdt))] = {
                "native": bool(st.comm.native), "buckets": len(st.comm.last_launch_log),
                "data_equal": bool(torch.equal(st.flat.data, ref_data)), "grad_equal": bool(torch.equal(st.flat.grad, ref_grad)),
                "grad_rel": float((st.flat.grad - ref_grad).norm() / ref_grad.norm()), "loss_equal": losses == base_losses}
            st.comm.close()
    finally:
        dist.destroy_process_group()
    q.put(out)


@pytest.mark.gpu
def test_native_exchange_gpu(hip):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_native_single_rank_worker, args=(29800 + (os.getpid() % 90), q))
    p.start()
    out = q.get(timeout=600)
    p.join(timeout=60)
    assert p.exitcode == 0
    for key, r in out.items():
        assert r["native"] and r["buckets"] > 4, (key, r)
        if key[2] == "None":
            assert r["grad_equal"] and r["data_equal"] and r["loss_equal"], (key, r)
        else:
            assert 0 < r["grad_rel"] < 8e-3, (key, r)


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI); the single-GPU box runs the gloo twin on CPU")
def test_data_parallel_step_rccl_world2(hip):
    """The overlapped reduce-scatter / all-gather exchange on RCCL proper: two ranks, two GPUs, replicas must stay identical."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, 29700 + (os.getpid() % 200), q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(results.values())


# ---- the real autoencoder step: MS-STFT discriminator, alternating discriminator / generator updates ----
def _disc_config():
    cfg = _model_config()
    tr = cfg["training"]
    tr["optimizer_configs"]["discriminator"] = {
        "optimizer": {"type": "AdamW", "config": {"betas": [0.8, 0.99], "lr": 2e-3, "weight_decay": 1e-3, "eps": 1e-3}},
        "scheduler": {"type": "InverseLR", "config": {"inv_gamma": 200000, "power": 0.5, "warmup": 0.9}}}
    tr["loss_configs"]["discriminator"] = {"type": "encodec", "config": {"filters": 4, "n_ffts": [64, 32], "hop_lengths": [16, 8],
                                                                         "win_lengths": [64, 32]},
                                           "weights": {"adversarial": 0.1, "feature_matching": 5.0}}
    return cfg


def _alternating(device):
    """Step 0 (even): generator update on spectral + kl + 0.1 adversarial + 5 feature-matching; step 1 (odd): discriminator update on
    its hinge loss with the UPDATED autoencoder (training/autoencoders.py:440-515).  Against a plain-torch restatement built from
    the oracles (vae_oracle, stft_oracle, disc_oracle) with torch.optim.AdamW."""
    import disc_oracle
    from stable_audio_tools_amd.training import AutoencoderTrainStep, inverse_lr
    cfg = _disc_config()
    torch.manual_seed(7)
    model = build_native_ae(NAME, SEED, device)
    stepper = AutoencoderTrainStep(model, cfg)
    dsd0 = {k: v.detach().cpu().clone() for k, v in stepper.discriminator.state_dict().items()}
    batches = [_batch(2, 900), _batch(2, 910)]
    out = [stepper(a.to(device), noise=n.to(device)) for a, n in batches]
    assert stepper.gen_steps == 1 and stepper.disc_steps == 1 and "feature_matching" in out[0] and "discriminator_loss" in out[1]

    # oracle side
    from stable_audio_tools_amd.autoencoders import create_autoencoder_from_config
    shapes = {k: tuple(v.shape) for k, v in create_autoencoder_from_config(cfg).state_dict().items()}
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in seeded.seeded_state_dict(shapes, SEED).items()}
    dsd = {("discriminators." + k if not k.startswith("discriminators.") else k): v.clone().requires_grad_(True) for k, v in dsd0.items()}
    tr = cfg["training"]
    oa, od = tr["optimizer_configs"]["autoencoder"], tr["optimizer_configs"]["discriminator"]
    opt_g = torch.optim.AdamW(list(sd.values()), lr=inverse_lr(0, oa["optimizer"]["config"]["lr"], **oa["scheduler"]["config"]),
                              betas=(0.8, 0.99), weight_decay=1e-3, eps=1e-3)
    opt_d = torch.optim.AdamW(list(dsd.values()), lr=inverse_lr(0, od["optimizer"]["config"]["lr"], **od["scheduler"]["config"]),
                              betas=(0.8, 0.99), weight_decay=1e-3, eps=1e-3)
    dc = tr["loss_configs"]["discriminator"]["config"]
    sc = tr["loss_configs"]["spectral"]["config"]
    (a0, n0), (a1, n1) = batches
    z, kl, _ = vae_oracle.autoencoder_encode(sd, cfg["model"], a0, n0)
    dec = vae_oracle.autoencoder_decode(sd, cfg["model"], z)
    _, adv, fm = disc_oracle.discriminator_losses(dsd, a0, dec, dc["n_ffts"], dc["hop_lengths"], dc["win_lengths"])
    loss_g = stft_oracle.autoencoder_spectral_loss(a0, dec, sc, cfg["sample_rate"]) + 1e-4 * kl + 0.1 * adv + 5.0 * fm
    opt_g.zero_grad()
    opt_d.zero_grad()
    loss_g.backward()
    opt_g.step()
    with torch.no_grad():
        z, _, _ = vae_oracle.autoencoder_encode(sd, cfg["model"], a1, n1)
        dec = vae_oracle.autoencoder_decode(sd, cfg["model"], z)
    opt_d.zero_grad()
    loss_d, _, _ = disc_oracle.discriminator_losses(dsd, a1, dec, dc["n_ffts"], dc["hop_lengths"], dc["win_lengths"])
    loss_d.backward()
    opt_d.step()
    assert abs(float(out[0]["loss"]) - float(loss_g)) <= 1e-3 * abs(float(loss_g)), (float(out[0]["loss"]), float(loss_g))
    assert abs(float(out[1]["loss"]) - float(loss_d)) <= 1e-3 * abs(float(loss_d)), (float(out[1]["loss"]), float(loss_d))
    # the discriminator update itself (direction of the step; its size is ~lr per element with Adam)
    ups, refs = [], []
    for k, v in stepper.discriminator.state_dict().items():
        ups.append((v.detach().cpu() - dsd0[k]).reshape(-1))
        refs.append((dsd[k].detach() - dsd0[k]).reshape(-1))
    cos = float(torch.nn.functional.cosine_similarity(torch.cat(ups), torch.cat(refs), dim=0))
    assert cos >= 0.995, cos


def test_alternating_discriminator_generator_steps_simulator(emu_modules):
    _alternating("cpu")


@pytest.mark.gpu
def test_alternating_discriminator_generator_steps_gpu(hip):
    _alternating("cuda")


def _ddp_disc_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "oracle"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from emu_util import use_emu_ops
    use_emu_ops()
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(7)                    # the discriminator's initialisation: the same on every rank
        batches = [_batch(2, 900), _batch(2, 910)]
        _, stepper, _ = _native_steps(_disc_config(), [(a[rank:rank + 1], n[rank:rank + 1]) for a, n in batches], "cpu",
                                      ddp_mode="reduce_scatter", bucket_bytes=4096)
        assert stepper.gen_steps == 1 and stepper.disc_steps == 1 and len(stepper.comm_d.buckets) >= 1
        both = torch.cat([stepper.flat.data.reshape(-1), stepper.flat_d.data.reshape(-1)])
        gathered = [torch.empty_like(both) for _ in range(world)]
        dist.all_gather(gathered, both)
        q.put((rank, ([g.numpy() for g in gathered], stepper.flat.data.numel()) if rank == 0 else None))
    finally:
        dist.destroy_process_group()


def test_alternating_steps_data_parallel_gloo_world2(emu_modules):
    """The REAL step (generator update with adversarial + feature-matching terms, then discriminator update) on two ranks, one item each,
    gradients of both parameter sets exchanged from the backward hooks (the discriminator's scale-by-scale backward included): the
    ranks stay identical and the updates equal the single-process full-batch ones."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + 23
    procs = [ctx.Process(target=_ddp_disc_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, r1), n_g = results[0]
    r0, r1 = torch.from_numpy(r0), torch.from_numpy(r1)
    assert torch.equal(r0, r1), "ranks diverged after the exchanged generator + discriminator updates"
    from stable_audio_tools_amd.training import AutoencoderTrainStep
    torch.manual_seed(7)
    model = build_native_ae(NAME, SEED, "cpu")
    stepper = AutoencoderTrainStep(model, _disc_config())
    g0, d0 = stepper.flat.data.clone(), stepper.flat_d.data.clone()
    for a, n in [_batch(2, 900), _batch(2, 910)]:
        stepper(a, noise=n)
    ng, nd = g0.numel(), d0.numel()
    for name, got, ref, init in (("generator", r0[:ng], stepper.flat.data.reshape(-1), g0.reshape(-1)),
                                 ("discriminator", r0[n_g:n_g + nd], stepper.flat_d.data.reshape(-1), d0.reshape(-1))):
        up, ur = got - init, ref - init
        cos = float(torch.nn.functional.cosine_similarity(up, ur, dim=0))
        assert cos >= 0.98, (name, cos)


def _adamw_case(ops, dev):
    """sat_adamw_step (16-byte vector body + scalar tail) against torch.optim.AdamW over three steps, sizes around the vector width,
    with the gradient scale (1 / world) and the EMA shadow (updated from the parameters BEFORE the step, training/autoencoders.py:504-515)."""
    for n in (1, 3, 4, 1030, 4099):
        gen = torch.Generator().manual_seed(n)
        p0 = torch.randn(n, generator=gen)
        ref = torch.nn.Parameter(p0.clone())
        opt = torch.optim.AdamW([ref], lr=1e-2, betas=(0.8, 0.95), eps=1e-6, weight_decay=0.1)
        p = p0.clone().to(dev)
        m, v, ema = torch.zeros(n, device=dev), torch.zeros(n, device=dev), p0.clone().to(dev)
        ema_ref = p0.clone()
        for step in (1, 2, 3):
            g = torch.randn(n, generator=gen)
            ema_ref = 0.9 * ema_ref + 0.1 * ref.detach()
            ref.grad = g.clone()
            opt.step()
            ops.adamw_step(p, (2.0 * g).to(dev), m, v, 1e-2, 0.8, 0.95, 1e-6, 0.1, step, grad_scale=0.5, ema=ema, ema_decay=0.9)
            assert torch.allclose(p.cpu(), ref.detach(), rtol=1e-5, atol=1e-6), (n, step)
            assert torch.allclose(ema.cpu(), ema_ref, rtol=1e-6, atol=1e-7), (n, step)


def test_adamw_kernel_simulator(emu):
    _adamw_case(emu, "cpu")


@pytest.mark.gpu
def test_adamw_kernel_gpu(hip):
    _adamw_case(hip, "cuda")


def _clip_case(device):
    from stable_audio_tools_amd.training import FlatParameters, clip_flat_grads
    gen = torch.Generator().manual_seed(5)
    for max_norm, world in ((0.5, 1), (1e3, 1), (0.25, 4)):
        params = [torch.nn.Parameter(torch.randn(*shp, generator=gen).to(device)) for shp in ((7, 3), (5,), (2, 4, 6), (300, 131))]
        refs = [torch.nn.Parameter(q.detach().clone()) for q in params]
        flat = FlatParameters(params, pad_to=world)
        for q, r in zip(params, refs):
            g = torch.randn(q.shape, generator=gen).to(device)
            q.grad.copy_(g * world)                 # the flat buffer holds the SUM over ranks
            r.grad = g.clone()                      # the mean gradient the reference clips
        total = torch.nn.utils.clip_grad_norm_(refs, max_norm)
        norm = clip_flat_grads(flat, max_norm, grad_scale=1.0 / world)
        assert torch.allclose(norm, total, rtol=1e-5)
        for q, r in zip(params, refs):
            assert torch.allclose(q.grad / world, r.grad, rtol=1e-5, atol=1e-7)


def test_clip_flat_grads_matches_torch(emu_modules):
    """training.clip_flat_grads == torch.nn.utils.clip_grad_norm_ (the reference's clipping, training/autoencoders.py:491-492, :509-510)
    on the flat buffer, incl. the 1 / world scale of a summed multi-rank gradient and the no-clip case.  The norm runs on ops.sum_all
    (sat_rowsum passes), not on a torch reduction."""
    _clip_case("cpu")


@pytest.mark.gpu
def test_clip_flat_grads_matches_torch_gpu(hip):
    _clip_case("cuda")


def _sum_all_case(device, sizes):
    from stable_audio_tools_amd import functional as Fn
    gen = torch.Generator().manual_seed(11)
    for shape in sizes:
        x = torch.randn(*shape, generator=gen).to(device).requires_grad_(True)
        ref = x.detach().double().sum()
        got = Fn.sum_all(x)
        assert got.shape == () and abs(float(got) - float(ref)) <= 2e-6 * float(x.detach().abs().double().sum()), (shape, float(got), float(ref))
        # autograd: d/dx of 3 * mean(relu(1 - x)) as torch's own mean gives it
        (3.0 * Fn.mean_all(torch.relu(1 - x))).backward()
        x2 = x.detach().clone().requires_grad_(True)
        (3.0 * torch.relu(1 - x2).mean()).backward()
        assert torch.equal(x.grad, x2.grad), shape
    # a non-contiguous (strided) input and a 16-bit one
    base = torch.randn(6, 50, generator=gen).to(device)
    assert abs(float(Fn.sum_all(base[:, 3:43:2])) - float(base[:, 3:43:2].double().sum())) <= 1e-4
    h = torch.randn(1000, generator=gen).to(device).to(torch.bfloat16)
    assert abs(float(Fn.sum_all(h)) - float(h.double().sum())) <= 1e-3
    assert float(Fn.sum_all(torch.zeros(0, device=device))) == 0.0


def test_sum_all_simulator(emu_modules):
    """functional.sum_all / mean_all (ops.sum_all: passes of sat_rowsum over (1, 1, N)) against float64 sums: sizes with and without
    the 16-byte row path, one element, more than one pass (N > 16384)."""
    _sum_all_case("cpu", [(1,), (7, 3), (4, 1024), (3, 5, 4099), (40000,)])


@pytest.mark.gpu
def test_sum_all_gpu(hip):
    _sum_all_case("cuda", [(1,), (7, 3), (4, 1024), (3, 5, 4099), (1, 2, 2097152), (33554433,)])


@pytest.mark.gpu
def test_sum_all_survives_graph_replay_gpu(hip):
    """The reason ops.sum_all exists: replayed from a HIP graph, torch's multi-block reductions return stale / foreign values after a few
    replays on this stack (tools/diag_graph_reduce.py reproduces it with torch ops alone; profiles/r04_experiments/graph_reductions/).
    The row-sum passes are plain kernels without a semaphore buffer: forty replays of forty reductions over 4 M elements each, every
    output equal to the eager evaluation of the same input."""
    from stable_audio_tools_amd import functional as Fn
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(1)
    src = [torch.randn(4 << 20, device=dev, generator=gen) * s for s in (0.1, 0.5, 0.25)]
    static = src[0].clone()

    def body(inp):
        outs, y = [], inp
        for k in range(40):
            y = y * 1.0001 + 0.001 * k
            outs.append(Fn.mean_all(torch.relu(1 - y)))
            outs.append(Fn.sum_all(y.abs()))
        return outs
    body(static)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = body(static)
    for r in range(40):
        static.copy_(src[r % 3] * (1.0 + 0.01 * r))
        junk = [torch.randn((8 + (r + j) % 5) << 20, device=dev) for j in range(6)]      # allocator churn between replays
        del junk
        graph.replay()
        got = [float(o) for o in outs]
        want = [float(o) for o in body(static)]
        assert got == want, (r, [(i, a, b) for i, (a, b) in enumerate(zip(got, want)) if a != b][:4])


def test_warmup_and_time_losses_simulator(emu_modules):
    """warmup_steps = 1 (mode 'adv'): step 0 is a generator step WITHOUT adversarial / feature-matching terms but with the time-domain
    L1 / MSE terms, step 1 a discriminator step, step 2 a generator step with them (training/autoencoders.py:378-379, :440-452, :476-483);
    step 0's loss against the oracles."""
    import copy as _copy
    from stable_audio_tools_amd.training import AutoencoderTrainStep
    cfg = _copy.deepcopy(_disc_config())
    cfg["training"]["warmup_steps"] = 1
    cfg["training"]["loss_configs"]["time"] = {"weights": {"l1": 0.5, "l2": 0.25}}
    cfg["training"]["clip_grad_norm"] = 10.0
    torch.manual_seed(7)
    model = build_native_ae(NAME, SEED, "cpu")
    stepper = AutoencoderTrainStep(model, cfg)
    batches = [_batch(2, 900), _batch(2, 910), _batch(2, 920)]
    out = [stepper(a, noise=n) for a, n in batches]
    assert "feature_matching" not in out[0] and "l1_time_loss" in out[0] and "discriminator_loss" in out[1] and "feature_matching" in out[2]
    assert stepper.gen_steps == 2 and stepper.disc_steps == 1
    from stable_audio_tools_amd.autoencoders import create_autoencoder_from_config
    shapes = {k: tuple(v.shape) for k, v in create_autoencoder_from_config(cfg).state_dict().items()}
    sd = {k: torch.from_numpy(v).clone() for k, v in seeded.seeded_state_dict(shapes, SEED).items()}
    a0, n0 = batches[0]
    z, kl, _ = vae_oracle.autoencoder_encode(sd, cfg["model"], a0, n0)
    dec = vae_oracle.autoencoder_decode(sd, cfg["model"], z)
    sc = cfg["training"]["loss_configs"]["spectral"]["config"]
    ref = stft_oracle.autoencoder_spectral_loss(a0, dec, sc, cfg["sample_rate"]) + 1e-4 * kl \
        + 0.5 * (a0 - dec).abs().mean() + 0.25 * ((a0 - dec) ** 2).mean()
    assert abs(float(out[0]["loss"]) - float(ref)) <= 1e-3 * abs(float(ref)), (float(out[0]["loss"]), float(ref))


# ---- the whole update as ONE HIP graph (training.GraphedTrainStep): same kernels, same order -> bit-identical to the eager step ----
def _graphed_vs_eager(cfg, nsteps, device="cuda", demo_between=False):
    from stable_audio_tools_amd.training import AutoencoderTrainStep, GraphedTrainStep
    batches = [_batch(2, 900 + 10 * i) for i in range(nsteps)]

    def run(graphed):
        torch.manual_seed(7)
        model = build_native_ae(NAME, SEED, device)
        stepper = AutoencoderTrainStep(model, cfg)
        step = GraphedTrainStep(stepper, eager_steps=1) if graphed else stepper
        losses = []
        for a, n in batches:
            if demo_between and graphed:
                # a demo / validation pass at the CURRENT parameter epoch right before the update (and therefore right before the capture):
                # its derived-weight cache entries must not be what the captured update reads (capture froze cache hits)
                with torch.no_grad():
                    model.decode(model.encode(a.to(device), noise=n.to(device)))
            out = step(a.to(device), noise=n.to(device))
            losses.append({k: float(v) for k, v in out.items()})
        torch.cuda.synchronize()
        extra = (step.replays, dict(step.fallback), len(step.graphs)) if graphed else None
        disc = stepper.flat_d.data.clone() if stepper.discriminator is not None else None
        return stepper.flat.data.clone(), stepper.opt.ema.clone(), disc, losses, (stepper.gen_steps, stepper.disc_steps, stepper.opt.t), extra
    pe, ee, de, le, ce, _ = run(False)
    pg, eg, dg, lg, cg, (replays, fallback, ngraphs) = run(True)
    assert not fallback, fallback
    assert ce == cg and replays > 0 and ngraphs >= 1
    # same kernels, same order, the same fp32 optimizer scalars (FusedAdamW.hyper rounds them as the eager launch does): every loss of
    # every step is bit-identical; the parameters may differ in the last bit (the by-value and the from-memory AdamW launches are two
    # compilations of the same expression)
    assert le == lg, (le, lg)

    def close(x, y):
        return float((x - y).abs().max()) <= 1e-6 * float(x.abs().max())
    assert close(pe, pg) and close(ee, eg), (float((pe - pg).abs().max()), float((ee - eg).abs().max()))
    if de is not None:
        assert close(de, dg), float((de - dg).abs().max())
    return replays, ngraphs


@pytest.mark.gpu
def test_graphed_generator_step_equals_eager_gpu(hip):
    """Six generator steps (InverseLR warm-up, Adam bias corrections and the EMA decay all change from step to step): replays of the
    captured update — the per-step scalars read from device memory — reproduce the eager steps."""
    replays, ngraphs = _graphed_vs_eager(_model_config(), 6)
    assert replays == 5 and ngraphs == 1          # call 1 eager, call 2 captures and replays, calls 3..6 replay


@pytest.mark.gpu
def test_graphed_alternating_step_equals_eager_gpu(hip):
    """The real step (generator / discriminator alternating, adversarial + feature-matching terms): one graph per kind of update."""
    replays, ngraphs = _graphed_vs_eager(_disc_config(), 8)
    assert ngraphs == 2 and replays == 6          # per kind: one eager call, three replays


@pytest.mark.gpu
def test_graphed_step_with_demo_passes_between_gpu(hip):
    """no_grad encode / decode passes between the updates (what the wrapper's demo callback and validation do) fill the derived-weight
    caches at the parameter epoch the capture runs at; the captured update must still contain its own fold / pack launches: losses of
    every replayed step equal the eager stepper's."""
    replays, ngraphs = _graphed_vs_eager(_disc_config(), 8, demo_between=True)
    assert ngraphs == 2 and replays == 6


# ---- wrapper options of the reference's training step: force_input_mono, latent_mask_ratio, LossModule.decay ----
def _wrapper_options(device):
    """training/autoencoders.py:45-46, :387-388 (mono encoder input), :411-413 (latent masking), training/losses/losses.py:9-24 + :102-104
    (the weight of a loss is multiplied by `decay` at every evaluation, before it is applied).  Two generator steps of the native step
    against the oracle forward assembled the same way + torch.optim.AdamW."""
    from stable_audio_tools_amd.autoencoders import create_autoencoder_from_config
    from stable_audio_tools_amd.training import AutoencoderTrainStep, inverse_lr
    cfg = _model_config()
    cfg["model"]["encoder"]["config"]["in_channels"] = 1          # stereo items, mono encoder, stereo decoder
    cfg["training"]["force_input_mono"] = True
    cfg["training"]["latent_mask_ratio"] = 0.25
    cfg["training"]["loss_configs"]["spectral"]["decay"] = 0.5
    cfg["training"]["loss_configs"]["time"] = {"weights": {"l1": 0.3, "l2": 0.2}, "decay": 0.8}
    torch.manual_seed(3)
    model = create_autoencoder_from_config(cfg)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    init = {k: torch.from_numpy(v) for k, v in seeded.seeded_state_dict(shapes, SEED).items()}
    model.load_state_dict(init)
    model = model.to(device)
    stepper = AutoencoderTrainStep(model, cfg)
    assert stepper.step_scalars_change
    batches = [_batch(2, 900), _batch(2, 910)]
    masks = [torch.from_numpy(seeded.seeded_array((2, 4, 64), 77 + i)) < -0.6 for i in range(2)]
    outs = [stepper(a.to(device), noise=n.to(device), latent_mask=mk.to(device)) for (a, n), mk in zip(batches, masks)]
    # the same two steps from the oracle
    sd = {k: v.clone().requires_grad_(True) for k, v in init.items()}
    oc = cfg["training"]["optimizer_configs"]["autoencoder"]
    opt = torch.optim.AdamW(list(sd.values()), lr=1e-3, betas=(0.8, 0.99), weight_decay=1e-3, eps=1e-3)
    sc = cfg["training"]["loss_configs"]["spectral"]["config"]
    for step, ((audio, noise), mk) in enumerate(zip(batches, masks)):
        for gp in opt.param_groups:
            gp["lr"] = inverse_lr(step, 1e-3, **oc["scheduler"]["config"])
        z, kl, _ = vae_oracle.autoencoder_encode(sd, cfg["model"], audio.mean(dim=1, keepdim=True), noise)
        z = torch.where(mk, torch.zeros_like(z), z)
        dec = vae_oracle.autoencoder_decode(sd, cfg["model"], z)
        k = step + 1
        spec = stft_oracle.autoencoder_spectral_loss(audio, dec, sc, cfg["sample_rate"]) * 0.5 ** k
        l1, l2 = (audio - dec).abs().mean(), ((audio - dec) ** 2).mean()
        loss = spec + 1e-4 * kl + 0.3 * 0.8 ** k * l1 + 0.2 * 0.8 ** k * l2
        assert abs(float(outs[step]["mrstft_loss"]) - float(spec)) <= 1e-3 * abs(float(spec)), (step, float(outs[step]["mrstft_loss"]), float(spec))
        assert abs(float(outs[step]["l1_time_loss"]) - float(0.3 * 0.8 ** k * l1)) <= 1e-3 * abs(float(l1))
        assert abs(float(outs[step]["loss"]) - float(loss)) <= 1e-3 * abs(float(loss)), (step, float(outs[step]["loss"]), float(loss))
        opt.zero_grad()
        loss.backward()
        opt.step()


def test_wrapper_options_simulator(emu_modules):
    _wrapper_options("cpu")


# ---- teacher distillation (training/factory.py:31-40, training/autoencoders.py:169-179, :405-408, :429-437) ----
def _teacher_case(device):
    """With a teacher the generator loss is five terms at mrstft / 4 each: MSE(teacher latents, own latents before masking) and the
    sum-and-difference STFT loss on (reals, decoded), (teacher decoded, decoded), (reals, teacher-decoder(own latents)) and
    (reals, own-decoder(teacher latents)) — the last two computed under no_grad in the reference, i.e. value only —, no per-channel
    L / R terms; `decay` multiplies all five.  Two steps of the native step against the oracle forward assembled that way +
    torch.optim.AdamW (the second step's losses check the first update)."""
    from stable_audio_tools_amd.training import AutoencoderTrainStep, inverse_lr
    cfg = _model_config()
    cfg["training"]["loss_configs"]["spectral"]["decay"] = 0.9
    cfg["training"]["latent_mask_ratio"] = 0.25
    TSEED = SEED + 7
    model = build_native_ae(NAME, SEED, device)
    teacher = build_native_ae(NAME, TSEED, device)
    stepper = AutoencoderTrainStep(model, cfg, teacher_model=teacher)
    assert not any(p.requires_grad for p in teacher.parameters()) and not teacher.training
    batches = [_batch(2, 900), _batch(2, 910)]
    tnoise = [torch.from_numpy(seeded.seeded_array((2, 4, 64), 55 + i)) for i in range(2)]
    masks = [torch.from_numpy(seeded.seeded_array((2, 4, 64), 77 + i)) < -0.6 for i in range(2)]
    outs = [stepper(a.to(device), noise=n.to(device), latent_mask=mk.to(device), teacher_noise=tn.to(device))
            for (a, n), mk, tn in zip(batches, masks, tnoise)]
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in seeded.seeded_state_dict(shapes, SEED).items()}
    tsd = {k: torch.from_numpy(v).clone() for k, v in seeded.seeded_state_dict(shapes, TSEED).items()}
    oc = cfg["training"]["optimizer_configs"]["autoencoder"]
    opt = torch.optim.AdamW(list(sd.values()), lr=1e-3, betas=(0.8, 0.99), weight_decay=1e-3, eps=1e-3)
    sc = cfg["training"]["loss_configs"]["spectral"]["config"]
    taps = stft_oracle.aweighting_fir_taps(cfg["sample_rate"])

    def sdl(x, y):      # AuralossLoss: module(target_key tensor, input_key tensor)
        return stft_oracle.sum_and_difference_loss(x, y, sc["fft_sizes"], sc["hop_sizes"], sc["win_lengths"], taps)
    for step, ((audio, noise), mk, tn) in enumerate(zip(batches, masks, tnoise)):
        for gp in opt.param_groups:
            gp["lr"] = inverse_lr(step, 1e-3, **oc["scheduler"]["config"])
        z, kl, _ = vae_oracle.autoencoder_encode(sd, cfg["model"], audio, noise)
        with torch.no_grad():
            tz, _, _ = vae_oracle.autoencoder_encode(tsd, cfg["model"], audio, tn)
        zm = torch.where(mk, torch.zeros_like(z), z)
        dec = vae_oracle.autoencoder_decode(sd, cfg["model"], zm)
        with torch.no_grad():
            tdec = vae_oracle.autoencoder_decode(tsd, cfg["model"], tz)
            own_t = vae_oracle.autoencoder_decode(tsd, cfg["model"], zm)
            t_own = vae_oracle.autoencoder_decode(sd, cfg["model"], tz)
        w = 0.25 * 0.9 ** (step + 1)
        terms = {"latent_distill_loss": ((tz - z) ** 2).mean(), "mrstft_loss": sdl(audio, dec), "mrstft_loss_distill": sdl(tdec, dec),
                 "mrstft_loss_own_latents_teacher": sdl(audio, own_t), "mrstft_loss_teacher_latents_own": sdl(audio, t_own)}
        loss = w * sum(terms.values()) + 1e-4 * kl
        for k, v in terms.items():
            assert abs(float(outs[step][k]) - float(w * v)) <= 1e-3 * abs(float(w * v)), (step, k, float(outs[step][k]), float(w * v))
        assert abs(float(outs[step]["loss"]) - float(loss)) <= 1e-3 * abs(float(loss)), (step, float(outs[step]["loss"]), float(loss))
        opt.zero_grad()
        loss.backward()
        opt.step()
    # the update itself: parameters after two steps
    got = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    worst = max(rel_err(got[k], sd[k].detach()) for k in sd)
    assert worst <= 1e-3, worst
    # the config route of training/factory.py: a teacher config without its checkpoint is an error
    cfg2 = copy.deepcopy(cfg)
    cfg2["training"]["teacher_model"] = copy.deepcopy(seeded.AE_CONFIGS[NAME])
    with pytest.raises(ValueError):
        AutoencoderTrainStep(build_native_ae(NAME, SEED, device), cfg2)


def test_teacher_distillation_simulator(emu_modules):
    _teacher_case("cpu")


@pytest.mark.gpu
def test_teacher_distillation_gpu(hip):
    _teacher_case("cuda")


@pytest.mark.gpu
def test_wrapper_options_gpu(hip):
    _wrapper_options("cuda")


def _multi_copy_case(ops, dev):
    """sat_multi_copy (the gradient gather of FlatParameters.gather_grads): 401 pairs — three launches of <= 160 entries riding in the
    kernel arguments — of sizes around the 16-byte vector width and the 16384-element block, at aligned and odd offsets."""
    gen = torch.Generator().manual_seed(3)
    sizes = [1, 2, 3, 4, 5, 7, 8, 63, 64, 65, 255, 1000, 16383, 16384, 16385, 40000] * 25 + [123457]
    big = torch.zeros(sum(sizes) + len(sizes), device=dev)
    srcs, dsts, off = [], [], 0
    for i, n in enumerate(sizes):
        srcs.append(torch.randn(n + 1, generator=gen).to(dev)[i % 2:][:n])         # every other source starts 4 bytes off alignment
        dsts.append(big[off:off + n])
        off += n + 1                                                                # one guard element between destinations
    ops.multi_copy(srcs, dsts)
    off = 0
    for s, n in zip(srcs, sizes):
        assert torch.equal(big[off:off + n], s) and float(big[off + n]) == 0.0
        off += n + 1
    with pytest.raises(ValueError):
        ops.multi_copy([srcs[0]], [dsts[1]])


def test_multi_copy_sim(emu):
    _multi_copy_case(emu, "cpu")


@pytest.mark.gpu
def test_multi_copy_gpu(hip):
    _multi_copy_case(hip, "cuda")
