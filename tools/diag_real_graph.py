"""Diagnostic (round 4, last GPU minutes): `bench.py`'s real_step.hip_graph object reported a discriminator loss of 7238 from the replayed
discriminator update where the eager update gives 2.00003.  Replays the alternating step at the bench size under a few variants and prints,
per call, the loss and taps of the discriminator update's inputs (latents / decoded / reals max |x|, per-scale hinge terms).
usage: python tools/diag_real_graph.py [out.jsonl] [sample_size]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout"
SAMPLES = int(sys.argv[2]) if len(sys.argv) > 2 else 2097152
T0 = time.time()


def emit(o):
    o["t"] = round(time.time() - T0, 1)
    with open(OUT, "a") as f:
        f.write(json.dumps(o) + "\n")


def main():
    from stable_audio_tools_amd import ops as O
    from stable_audio_tools_amd import autoencoders as AE
    from stable_audio_tools_amd.autoencoders import create_autoencoder_from_config
    from stable_audio_tools_amd.training import AutoencoderTrainStep, GraphedTrainStep
    dev = torch.device("cuda", 0)
    cfg = json.load(open(os.path.join(ROOT, "stable_audio_tools_amd", "configs", "stable_audio_2_0_vae.json")))
    torch.manual_seed(1234)
    model = create_autoencoder_from_config(cfg).to(dev)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith("alpha") or n_.endswith("beta"):
                p.normal_(0.0, 0.1)
    stepper = AutoencoderTrainStep(model, cfg, use_discriminator=True)
    ops = O.get_ops()
    g = torch.Generator().manual_seed(0)
    batches = [(0.1 * torch.randn(1, 2, SAMPLES, generator=g)).to(dev) for _ in range(2)]
    emit({"built": True, "samples": SAMPLES})

    taps = {}

    def disc_body(reals, kw):
        """AutoencoderTrainStep._disc_body with taps"""
        s, m = stepper, stepper.model
        s.flat_d.zero_grad()
        with torch.no_grad():
            latents = m.encode(s._encoder_input(reals), **s._encode_kw(kw))
            latents = s._mask_latents(latents, kw)
            decoded, reals_t = s._trim(m.decode(latents), reals)
        decoded = decoded.contiguous()
        t = {"lat": latents.abs().max(), "dec": decoded.abs().max(), "real": reals_t.abs().max(), "dec_nan": torch.isnan(decoded).sum()}
        loss_dis = torch.zeros((), device=reals.device)
        for i in range(s.discriminator.discriminators.num_discriminators):
            dis_i, _, _ = s.discriminator.scale_losses(i, reals_t, decoded, need_fm=False)
            dis_i.backward()
            t[f"dis{i}"] = dis_i.detach()
            loss_dis = loss_dis + dis_i.detach()
        s.flat_d.gather_grads()
        t["gnorm"] = s.flat_d.grad.norm()
        s.comm_d()
        s.opt_d.step(lr=s._lr("disc"), grad_scale=s.comm_d.grad_scale)
        t["pnorm"] = s.flat_d.data.norm()
        out = {"loss": loss_dis.detach(), "discriminator_loss": loss_dis.detach()}
        out.update({"tap_" + k: v for k, v in t.items()})
        return out

    stepper._disc_body = disc_body

    def run(name, n_calls=10, noise=False, only=None, pre=None, post=None):
        if pre:
            pre()
        ops.release_workspaces()
        torch.cuda.empty_cache()
        g2 = GraphedTrainStep(stepper, eager_steps=1)
        if only is not None:
            orig = g2._capture

            def cap(key, kind, reals, nz):
                if kind != only:
                    raise RuntimeError("diag: this kind stays eager")
                return orig(key, kind, reals, nz)
            g2._capture = cap
        stepper.global_step = 0
        rows = []
        nz = [torch.randn(1, 64, SAMPLES // 2048, device=dev) for _ in range(2)] if noise else None
        for i in range(n_calls):
            kind = stepper._kind()
            o = g2(batches[i % 2], noise=nz[i % 2] if noise else None)
            row = {"i": i, "kind": kind, "graphed": g2.replays}
            row.update({k: float(v) for k, v in o.items() if k == "loss" or k.startswith("tap_")})
            rows.append(row)
        torch.cuda.synchronize()
        cont = []
        for i in range(2):
            kind = stepper._kind()
            o = stepper(batches[i % 2], noise=nz[i % 2] if noise else None)
            cont.append({"kind": kind, **{k: float(v) for k, v in o.items() if k == "loss" or k.startswith("tap_")}})
        emit({"variant": name, "graphs": len(g2.graphs), "replays": g2.replays, "fallback": list(g2.fallback.values()), "calls": rows, "eager_after": cont})
        del g2
        if post:
            post()

    # eager reference first
    stepper.global_step = 0
    rows = []
    for i in range(4):
        kind = stepper._kind()
        o = stepper(batches[i % 2])
        rows.append({"i": i, "kind": kind, **{k: float(v) for k, v in o.items() if k == "loss" or k.startswith("tap_")}})
    emit({"variant": "eager", "calls": rows})

    def attempt(*a, **k):
        try:
            run(*a, **k)
        except Exception as e:      # noqa: BLE001
            emit({"variant": a[0], "error": repr(e)[:500]})
            if k.get("post"):
                k["post"]()

    attempt("bench_like")
    attempt("explicit_noise", noise=True)
    attempt("disc_graph_only", only="disc")

    def fuse_off():
        AE.ResidualUnit.fuse = False

    def fuse_auto():
        AE.ResidualUnit.fuse = None
    ip = AE._inference_pass

    def cache_off():
        AE._inference_pass = lambda *t: False

    def cache_on():
        AE._inference_pass = ip
    attempt("no_derived_cache", pre=cache_off, post=cache_on)
    attempt("no_fused_unit", pre=fuse_off, post=fuse_auto)

    def emit_off():
        type(ops).k7_emit = False

    def emit_on():
        type(ops).k7_emit = True
    attempt("no_plane_emission", pre=emit_off, post=emit_on)
    attempt("gen_graph_only", only="gen")
    emit({"done": True})


if __name__ == "__main__":
    main()
