"""Full-width golden fixtures: the REFERENCE itself (imported from /root/reference, CPU) at the real
widths of BASELINE.json configs[1] and configs[2], on short crops.  Run in the build container only:

    python oracle/gen_golden_full.py [vae] [dit2] [dit24]

TEST INFRASTRUCTURE ONLY.  Weights and inputs come from oracle/seeded.py (numpy legacy RandomState), so
the tests regenerate them bit-identically on the GPU box; only outputs are committed (tests/golden/full_*.npz):

  full_vae.npz    stable_audio_2_0_vae architecture (channels 128, c_mults 1/2/4/8/16, strides 2/4/4/8/8,
                  156 M parameters), 32768-sample stereo crop: pre-latents, z, kl, decoded, the generator loss
                  (MR-STFT sum/diff + L + R, 7 resolutions, A-weighted, + 1e-4 KL — training/autoencoders.py:142-194)
                  and its gradients, plus the gradients of a linear functional of the output (well conditioned).
                  Gradients are taken from the reference run in float64 ("truth"); the distance of the reference's
                  own float32 gradients to that truth is stored beside them, and is the yard-stick of the test.
  full_dit2.npz   2 layers of the Stable Audio Open DiT block: d=1536, 24 x 64 heads, GQA 24:12 cross-attention
                  to 130 x 768 context tokens, N = 1025 tokens, batch 2; output, hidden states, every gradient
                  of the v-objective MSE (full tensors for the small parameters, norms + a seeded 1024-element
                  probe for the large ones).
  full_dit24.npz  the full depth-24 model forward (fp32 reference), plain and with CFG.
"""
import math
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refimport  # noqa: E402
import seeded  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _load_seeded(model, seed, dtype):
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith("inv_freq")}
    sd = seeded.seeded_state_dict(shapes, seed)
    model.load_state_dict({k: torch.from_numpy(v).to(dtype) for k, v in sd.items()}, strict=False)
    return model


def _ref_generator_loss(al, reals, decoded, kl, dtype):
    """training/autoencoders.py:142-146,186-194,423-427 restated around the reference's own auraloss modules."""
    cfg = seeded.STFT_CFG
    sd_loss = al.SumAndDifferenceSTFTLoss(sample_rate=44100, **cfg).to(dtype)
    lr_loss = al.MultiResolutionSTFTLoss(sample_rate=44100, **cfg).to(dtype)
    l_sd = sd_loss(reals, decoded)                       # AuralossLoss passes (target, input): losses.py:111
    l_l = lr_loss(reals[:, 0:1], decoded[:, 0:1])
    l_r = lr_loss(reals[:, 1:2], decoded[:, 1:2])
    return 1.0 * l_sd + 0.5 * l_l + 0.5 * l_r + seeded.FULL_VAE["kl_weight"] * kl, (l_sd, l_l, l_r)


def _vae_run(al, dtype, perturb=0.0):
    from stable_audio_tools.models.autoencoders import create_autoencoder_from_config
    spec = seeded.FULL_VAE
    cfg = seeded.full_vae_config()
    model = _load_seeded(create_autoencoder_from_config(cfg).to(dtype), spec["seed"], dtype)
    model.train(False)
    audio, noise, proj = [torch.from_numpy(a).to(dtype) for a in seeded.full_vae_inputs()]
    for p in model.parameters():
        p.requires_grad_(True)
    pre = model.encoder(audio)
    mean, scale = pre.chunk(2, dim=1)
    stdev = torch.nn.functional.softplus(scale) + 1e-4          # bottleneck.py:105-113 with the draw injected
    z = noise * stdev + mean
    kl = (mean * mean + stdev * stdev - torch.log(stdev * stdev) - 1).sum(1).mean()
    dec = model.decode(z)
    dec_for_loss = dec
    if perturb:
        # conditioning probe: the generator loss evaluated on the decoded audio displaced by a seeded perturbation of
        # relative size `perturb` (the forward accuracy of a float32-class implementation that is not bit-identical)
        delta = torch.from_numpy(seeded.seeded_array(tuple(dec.shape), spec["seed"] + 9)).to(dtype)
        dec_for_loss = dec + perturb * dec.detach().abs().max() * delta
    loss_gen, parts = _ref_generator_loss(al, audio, dec_for_loss, kl, dtype)
    loss_lin = (dec * proj).sum() / proj.numel() ** 0.5 + 0.1 * kl
    names = [n for n, _ in model.named_parameters()]
    params = list(model.parameters())
    g_dec = torch.autograd.grad(loss_gen, dec, retain_graph=True)[0]      # dL/d(decoded): the MR-STFT backward alone
    g_gen = torch.autograd.grad(loss_gen, params, retain_graph=True)
    g_lin = torch.autograd.grad(loss_lin, params)
    return dict(g_dec=g_dec, names=names, pre=pre.detach(), z=z.detach(), kl=kl.detach(), dec=dec.detach(), loss_gen=loss_gen.detach(),
                loss_lin=loss_lin.detach(), parts=[p.detach() for p in parts], g_gen=g_gen, g_lin=g_lin)


def gen_vae():
    al = refimport.import_auraloss()
    t0 = time.time()
    r32 = _vae_run(al, torch.float32)
    r64 = _vae_run(al, torch.float64)
    rpt = _vae_run(al, torch.float64, perturb=seeded.FULL_VAE["fwd_eps"])
    out = {"pre": r32["pre"].numpy(), "z": r32["z"].numpy(), "kl": r32["kl"].numpy(), "decoded": r32["dec"].numpy(),
           "loss_gen": r32["loss_gen"].numpy(), "loss_lin": r32["loss_lin"].numpy(),
           "loss_sd": r32["parts"][0].numpy(), "loss_left": r32["parts"][1].numpy(), "loss_right": r32["parts"][2].numpy(),
           "loss_gen_f64": r64["loss_gen"].numpy(), "decoded_f32_vs_f64": np.float64(_rel(r32["dec"], r64["dec"]))}
    out["gdec_f64"] = r64["g_dec"].float().numpy()                      # dL/d(decoded) of the generator loss, float64 run
    out["gdec_refdist"] = np.float64(_rel(r32["g_dec"], r64["g_dec"]))
    worst = {"gen": 0.0, "lin": 0.0, "sens": 0.0}
    for i, n in enumerate(r32["names"]):
        for tag in ("gen", "lin"):
            g64, g32 = r64["g_" + tag][i], r32["g_" + tag][i]
            d = _rel(g32, g64)
            worst[tag] = max(worst[tag], d)
            out[f"gnorm_{tag}/{n}"] = np.float64(g64.norm().item())
            out[f"refdist_{tag}/{n}"] = np.float64(d)
            if tag == "gen":
                sens = _rel(rpt["g_gen"][i], g64)
                worst["sens"] = max(worst["sens"], sens)
                out[f"sens_gen/{n}"] = np.float64(sens)
            if g64.numel() <= seeded.FULL_KEEP_NUMEL:
                out[f"grad_{tag}/{n}"] = g64.float().numpy()
            else:
                out[f"probe_{tag}/{n}"] = g64.reshape(-1)[seeded.probe_index(n, g64.numel())].float().numpy()
    np.savez_compressed(os.path.join(OUT, "full_vae.npz"), **out)
    print(f"full_vae: decoded {tuple(r32['dec'].shape)} loss_gen {float(r32['loss_gen']):.6f} (f64 {float(r64['loss_gen']):.6f}) "
          f"ref f32-vs-f64 grad distance: gen {worst['gen']:.2e} lin {worst['lin']:.2e}; dL/ddec {float(out['gdec_refdist']):.2e}; "
          f"gen-grad sensitivity to a {seeded.FULL_VAE['fwd_eps']:.0e} forward perturbation {worst['sens']:.2e}  [{time.time() - t0:.0f} s]")


def dit_full_inputs(batch):
    spec = seeded.FULL_DIT
    cfg = spec["config"]
    rs = np.random.RandomState(spec["seed"] + 7)
    x0 = torch.from_numpy(seeded.seeded_array((batch, cfg["io_channels"], spec["latent_length"]), spec["seed"] + 1))
    t = torch.from_numpy(rs.uniform(0.05, 0.95, size=(batch,)).astype(np.float32))
    cross = torch.from_numpy(seeded.seeded_array((batch, spec["context_length"], cfg["cond_token_dim"]), spec["seed"] + 2))
    glob = torch.from_numpy(seeded.seeded_array((batch, cfg["global_cond_dim"]), spec["seed"] + 3))
    noise = torch.from_numpy(seeded.seeded_array(tuple(x0.shape), spec["seed"] + 4))
    alpha, sigma = torch.cos(t * math.pi / 2)[:, None, None], torch.sin(t * math.pi / 2)[:, None, None]
    return dict(x0=x0, t=t, cross=cross, glob=glob, noised=x0 * alpha + noise * sigma, target=noise * alpha - x0 * sigma)


def gen_dit2():
    from stable_audio_tools.models.dit import DiffusionTransformer
    spec = seeded.FULL_DIT
    cfg = dict(spec["config"], depth=2)
    t0 = time.time()
    model = _load_seeded(DiffusionTransformer(**cfg).float(), spec["seed"], torch.float32)
    model.train(False)
    inp = dit_full_inputs(2)
    xin = inp["noised"].clone().requires_grad_(True)
    out, info = model(xin, inp["t"], cross_attn_cond=inp["cross"], global_embed=inp["glob"], return_info=True)
    loss = torch.nn.functional.mse_loss(out, inp["target"])      # v-objective, training/diffusion.py:406-449
    names = [n for n, _ in model.named_parameters()]
    grads = torch.autograd.grad(loss, [xin] + list(model.parameters()))
    res = {"out": out.detach().numpy(), "hidden_first": info["hidden_states"][0].detach().numpy()[:, ::16],
           "hidden_last": info["hidden_states"][-1].detach().numpy()[:, ::16], "loss": loss.detach().numpy(),
           "grad/<input>": grads[0].numpy()}
    for n, g in zip(names, grads[1:]):
        res["gnorm/" + n] = np.float64(g.double().norm().item())
        if g.numel() <= seeded.FULL_KEEP_NUMEL:
            res["grad/" + n] = g.numpy()
        else:
            res["probe/" + n] = g.reshape(-1)[seeded.probe_index(n, g.numel())].numpy()
    np.savez_compressed(os.path.join(OUT, "full_dit2.npz"), **res)
    print(f"full_dit2: out {tuple(out.shape)} loss {float(loss):.6f} params {sum(p.numel() for p in model.parameters())} [{time.time() - t0:.0f} s]")


def gen_dit24():
    from stable_audio_tools.models.dit import DiffusionTransformer
    spec = seeded.FULL_DIT
    t0 = time.time()
    model = _load_seeded(DiffusionTransformer(**spec["config"]).float(), spec["seed"], torch.float32)
    model.train(False)
    inp = dit_full_inputs(1)
    with torch.no_grad():
        plain = model(inp["noised"], inp["t"], cross_attn_cond=inp["cross"], global_embed=inp["glob"], cfg_scale=1.0)
        guided = model(inp["noised"], inp["t"], cross_attn_cond=inp["cross"], global_embed=inp["glob"], cfg_scale=6.0, scale_phi=0.75)
    np.savez_compressed(os.path.join(OUT, "full_dit24.npz"), plain=plain.numpy(), guided=guided.numpy())
    print(f"full_dit24: out {tuple(plain.shape)} |plain| {float(plain.abs().max()):.3f} |guided| {float(guided.abs().max()):.3f} [{time.time() - t0:.0f} s]")


def main():
    os.makedirs(OUT, exist_ok=True)
    refimport.import_reference()
    torch.set_num_threads(os.cpu_count() or 8)
    what = sys.argv[1:] or ["vae", "dit2", "dit24"]
    if "vae" in what:
        gen_vae()
    if "dit2" in what:
        gen_dit2()
    if "dit24" in what:
        gen_dit24()


if __name__ == "__main__":
    main()
