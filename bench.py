#!/usr/bin/env python
"""bench.py — train-step throughput of the hot path on MI355X.

Workload (BASELINE.json configs[1], named in `config.workload`): one *generator optimisation step* of
the Oobleck audio VAE on 47.55 s stereo 44.1 kHz items (sample_size 2097152): encode -> VAE sample ->
decode -> multi-resolution STFT loss (sum/diff + L + R, 7 resolutions, A-weighted) + KL -> backward ->
data-parallel gradient all-reduce -> fused AdamW (+EMA).  fp32, synthetic audio, random-init weights of
the stable_audio_2_0_vae architecture.  The discriminator half of the reference step is out of scope
this round and is NOT inside the timed region — stated in `config`.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 launched by torch.distributed.run
(one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

SAMPLE_SIZE = 2097152
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (= fp32 vector peak)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="items per GPU per step")
    ap.add_argument("--sample-size", type=int, default=SAMPLE_SIZE)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-samples", type=int, default=32768)
    return ap.parse_args()


class ConvProfiler:
    """Times every launch of the dominant kernel (sat_conv1d_kernel) with HIP events on the launch stream
    and tallies its ALGORITHMIC flops (2 * Cin * Cout * K * Tout * B per launch — DESIGN.md §Kernels)."""

    def __init__(self, ops):
        self.records = []
        self.enabled = False
        orig = ops.lib.sat_conv1d   # the C-ABI entry point: exactly one sat_conv1d_kernel launch per call

        def timed(*a):
            if not self.enabled:
                return orig(*a)
            b, cin, cout, _tin, tout, k = a[12:18]
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()          # torch's current stream == the stream handed to the C-ABI (ops._stream)
            rc = orig(*a)
            e.record()
            self.records.append((s, e, 2.0 * b * cin * cout * k * tout))
            return rc

        ops.lib.sat_conv1d = timed

    def summary(self):
        torch.cuda.synchronize()
        ms = fl = 0.0
        for s, e, f in self.records:
            ms += s.elapsed_time(e)
            fl += f
        return len(self.records), ms, fl


def cpu_baseline(cfg, nsamples):
    """The oracle (CPU restatement of the reference path, oracle/*.py) timed on this box's host cores on a
    bounded sample: ONE generator step (fwd + autograd bwd + torch AdamW) on a `nsamples`-long stereo crop;
    the model is fully convolutional, so cost scales linearly with length."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import seeded
    import stft_oracle
    import vae_oracle
    from stable_audio_tools_amd.autoencoders import create_autoencoder_from_config
    # torch's CPU conv path degrades badly when oversubscribed (measured: 256 threads on the GPU box's host
    # took 700 s for what 8 threads do in ~25 s) — use a bounded pool and report the threads actually used
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    # same random-init recipe as the GPU replica (reference-format state_dict consumed by the oracle)
    torch.manual_seed(1234)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in create_autoencoder_from_config(cfg).state_dict().items()}
    opt = torch.optim.AdamW(list(sd.values()), lr=1.5e-4, betas=(0.8, 0.99), weight_decay=1e-3)
    g = torch.Generator().manual_seed(0)
    audio = 0.1 * torch.randn(1, 2, nsamples, generator=g)
    noise = torch.randn(1, cfg["model"]["latent_dim"], nsamples // cfg["model"]["downsampling_ratio"], generator=g)
    sc = cfg["training"]["loss_configs"]["spectral"]["config"]
    t0 = time.perf_counter()
    z, kl, _ = vae_oracle.autoencoder_encode(sd, cfg["model"], audio, noise)
    dec = vae_oracle.autoencoder_decode(sd, cfg["model"], z)
    loss = stft_oracle.autoencoder_spectral_loss(audio, dec, sc, cfg["sample_rate"]) + 1e-4 * kl
    loss.backward()
    opt.step()
    dt = time.perf_counter() - t0
    scale = SAMPLE_SIZE / nsamples
    return {"value": 1.0 / (dt * scale), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"1 generator step (oracle fwd + autograd bwd + AdamW) on a {nsamples}-sample stereo crop "
                      f"({dt:.2f} s), scaled x{scale:.0f} to {SAMPLE_SIZE} samples"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path for the product kernels)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)

    from stable_audio_tools_amd import ops as O
    from stable_audio_tools_amd.autoencoders import create_autoencoder_from_config
    from stable_audio_tools_amd.training import AutoencoderTrainStep

    cfg = json.load(open(os.path.join(ROOT, "stable_audio_tools_amd", "configs", "stable_audio_2_0_vae.json")))
    torch.manual_seed(1234)                     # identical random-init replica on every rank
    model = create_autoencoder_from_config(cfg).to(dev)
    # de-zero the SnakeBeta parameters a little so the activation path is not the trivial alpha=beta=1 case
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith("alpha") or n_.endswith("beta"):
                p.normal_(0.0, 0.1)
    stepper = AutoencoderTrainStep(model, cfg)
    ops = O.get_ops()
    prof = ConvProfiler(ops)

    g = torch.Generator().manual_seed(rank)     # per-rank data (train.py:30-33 seeds ranks differently)
    batches = [(0.1 * torch.randn(args.batch, 2, args.sample_size, generator=g)).to(dev) for _ in range(2)]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        stepper(batches[i % 2])
    sync()
    prof.enabled = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = stepper(batches[i % 2])
    sync()
    elapsed = time.perf_counter() - t0
    prof.enabled = False
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(out["loss"])

    if rank == 0:
        nlaunch, ms, flops = prof.summary()
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        line = {
            "metric": "train-step samples/sec (47s@44.1kHz)",
            "value": args.batch * world * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "oobleck_vae_generator_train_step(encode+vae_sample+decode+mrstft_sumdiff_LR_7res_aweighted+kl,"
                                   " backward, dp_allreduce, fused_adamw_ema); stable_audio_2_0_vae architecture, random init;"
                                   " discriminator terms excluded (SURVEY.md 8 f-3)",
                       "sample_size": args.sample_size, "channels": 2, "sample_rate": 44100,
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}", "final_loss": loss},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                         "kernel": "sat_conv1d_kernel", "launches": nlaunch,
                         "avg_launch_ms": (ms / nlaunch) if nlaunch else None,
                         "note": "algorithmic flops 2*Cin*Cout*K*Tout*B per launch over HIP-event time of EVERY sat_conv1d_kernel "
                                 "launch in the timed region (forward convs and data-gradients, incl. the dsnake epilogue "
                                 "variant); peak = fp32 MFMA (v_mfma_f32_32x32x2_f32) dense"},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_baseline_samples)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
