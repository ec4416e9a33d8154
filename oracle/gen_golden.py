"""Generates tests/golden/*.npz by running the REFERENCE itself (imported from /root/reference, CPU,
fp32) on seeded inputs.  Run in the build container only:

    python oracle/gen_golden.py

TEST INFRASTRUCTURE ONLY.  The reference ships no tests or golden vectors, so these
fixtures are what pins the oracle (tests/test_oracle_golden.py) and, through it, the HIP path.
Inputs and weights come from numpy's legacy RandomState so that tests can regenerate them
bit-identically without this script (oracle/seeded.py).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refimport  # noqa: E402
import seeded  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def _ae_config(name):
    return seeded.AE_CONFIGS[name]


def gen_vae(name, batch, in_len, seed):
    """Oobleck AudioAutoencoder through the reference factory: encode (pre-bottleneck), VAE sample with
    injected noise, decode, and autograd gradients of a fixed linear functional of the output."""
    from stable_audio_tools.models.autoencoders import create_autoencoder_from_config
    cfg = _ae_config(name)
    model = create_autoencoder_from_config(cfg).float()
    sd = seeded.seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.train(False)  # ResidualUnit checkpointing is a training-time memory trick only (autoencoders.py:78-79)
    ch = cfg["model"]["io_channels"]
    audio = torch.from_numpy(seeded.seeded_array((batch, ch, in_len), seed + 1, scale=0.5))
    lat_c = cfg["model"]["latent_dim"]
    ratio = cfg["model"]["downsampling_ratio"]
    noise = torch.from_numpy(seeded.seeded_array((batch, lat_c, in_len // ratio), seed + 2))
    proj = torch.from_numpy(seeded.seeded_array((batch, ch, in_len), seed + 3))

    for p in model.parameters():
        p.requires_grad_(True)
    pre = model.encoder(audio)                                   # == info["pre_bottleneck_latents"]
    mean, scale = pre.chunk(2, dim=1)
    # reference vae_sample draws randn internally (bottleneck.py:109); re-state with the injected draw
    stdev = torch.nn.functional.softplus(scale) + 1e-4
    z = noise * stdev + mean
    kl = (mean * mean + stdev * stdev - torch.log(stdev * stdev) - 1).sum(1).mean()
    dec = model.decode(z)
    loss = (dec * proj).sum() + 0.1 * kl
    names = [n for n, _ in model.named_parameters()]
    grads = torch.autograd.grad(loss, list(model.parameters()))
    # check the injected-noise restatement against the reference bottleneck under a shared torch seed
    torch.manual_seed(1234)
    zr, info = model.bottleneck.encode(pre.detach(), return_info=True)
    torch.manual_seed(1234)
    nz = torch.randn_like(mean)
    assert torch.allclose(zr, (nz * stdev + mean).detach(), atol=1e-6)
    out = {
        "pre": pre.detach().numpy(), "z": z.detach().numpy(), "kl": kl.detach().numpy(),
        "decoded": dec.detach().numpy(), "loss": loss.detach().numpy(),
    }
    # gradient fixtures: keep every parameter's gradient norm + full tensors for a representative subset
    keep = [n for n in names if any(s in n for s in ("encoder.layers.0.", "encoder.layers.1.layers.1.", "encoder.layers.1.layers.4.",
                                                      "decoder.layers.1.layers.1.", "decoder.layers.1.layers.0.",
                                                      "decoder.layers.1.layers.3.layers.0."))]
    for n, g in zip(names, grads):
        out["gnorm/" + n] = np.float32(g.norm().item())
        if n in keep:
            out["grad/" + n] = g.numpy()
    np.savez_compressed(os.path.join(OUT, f"vae_{name}.npz"), **out)
    print(f"vae_{name}: decoded {tuple(dec.shape)} loss {loss.item():.6f} ({len(keep)} full grads, {len(names)} norms)")


def gen_chunked(name, seed):
    """encode_audio / decode_audio with chunked=True (autoencoders.py:601-732)."""
    from stable_audio_tools.models.autoencoders import create_autoencoder_from_config
    cfg = _ae_config(name)
    model = create_autoencoder_from_config(cfg).float()
    sd = seeded.seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.train(False)
    ratio = cfg["model"]["downsampling_ratio"]
    lat = torch.from_numpy(seeded.seeded_array((1, cfg["model"]["latent_dim"], 44), seed + 5))
    with torch.no_grad():
        dec = model.decode_audio(lat, chunked=True, overlap=4, chunk_size=16)
        dec_full = model.decode_audio(lat, chunked=False)
        audio = torch.from_numpy(seeded.seeded_array((1, cfg["model"]["io_channels"], 44 * ratio), seed + 6, scale=0.5))
        # the reference's chunked encode drops kwargs and samples the VAE per chunk (autoencoders.py:646):
        # fix the CPU generator so the per-chunk randn_like draws are reproducible
        torch.manual_seed(4242)
        enc = model.encode_audio(audio, chunked=True, overlap=4, chunk_size=16)
    np.savez_compressed(os.path.join(OUT, f"vae_chunked_{name}.npz"), decoded_chunked=dec.numpy(), decoded_full=dec_full.numpy(),
                        encoded_chunked=enc.numpy())
    print(f"vae_chunked_{name}: {tuple(dec.shape)} chunk-vs-full maxdiff {float((dec - dec_full).abs().max()):.3e}")


def gen_stft(seed, batch=2, length=6000):
    al = refimport.import_auraloss()
    cfg = seeded.STFT_CFG
    sr = 44100
    torch.manual_seed(0)
    sd_loss = al.SumAndDifferenceSTFTLoss(sample_rate=sr, **cfg)
    lr_loss = al.MultiResolutionSTFTLoss(sample_rate=sr, **cfg)
    reals = torch.from_numpy(seeded.seeded_array((batch, 2, length), seed, scale=0.1))
    decoded = (reals + torch.from_numpy(seeded.seeded_array((batch, 2, length), seed + 1, scale=0.01))).requires_grad_(True)
    # AuralossLoss passes (target, input): loss_module(reals, decoded)  (training/losses/losses.py:111)
    l_sd = sd_loss(reals, decoded)
    l_l = lr_loss(reals[:, 0:1], decoded[:, 0:1])
    l_r = lr_loss(reals[:, 1:2], decoded[:, 1:2])
    total = 1.0 * l_sd + 0.5 * l_l + 0.5 * l_r
    (g,) = torch.autograd.grad(total, decoded)
    out = {"loss_sd": l_sd.detach().numpy(), "loss_left": l_l.detach().numpy(), "loss_right": l_r.detach().numpy(),
           "total": total.detach().numpy(), "grad_decoded": g.numpy(),
           "aw_taps": sd_loss.mrstft.stft_losses[0].prefilter.fir.weight.data.view(-1).numpy()}
    # per-resolution single STFTLoss values, mono, without and with A-weighting
    mono_x = reals[:, 0:1]
    mono_y = decoded[:, 0:1].detach()
    for n, h, w in zip(cfg["fft_sizes"], cfg["hop_sizes"], cfg["win_lengths"]):
        out[f"stft_plain_{n}"] = al.STFTLoss(n, h, w)(mono_x, mono_y).numpy()
        out[f"stft_aw_{n}"] = al.STFTLoss(n, h, w, perceptual_weighting=True, sample_rate=sr)(mono_x, mono_y).numpy()
    # mono model path: MultiResolutionSTFTLoss on a 1-channel signal
    out["loss_mono"] = lr_loss(mono_x, mono_y).numpy()
    np.savez_compressed(os.path.join(OUT, "mrstft.npz"), **out)
    print(f"mrstft: total {total.item():.6f} |grad| {g.norm().item():.4e}")


def dit_inputs(name, batch=2, length=37, ctx_len=11, seed=600):
    """Seeded DiT inputs shared with the tests (also imported by tests/test_dit_parity.py)."""
    cfg = seeded.DIT_CONFIGS[name]
    rs = np.random.RandomState(seed)
    out = {
        "x": torch.from_numpy(seeded.seeded_array((batch, cfg["io_channels"], length), seed + 1)),
        "t": torch.from_numpy(rs.uniform(0.05, 0.95, size=(batch,)).astype(np.float32)),
        "cross_attn_cond": torch.from_numpy(seeded.seeded_array((batch, ctx_len, cfg["cond_token_dim"]), seed + 2)),
        "global_embed": torch.from_numpy(seeded.seeded_array((batch, cfg["global_cond_dim"]), seed + 3)),
    }
    if cfg.get("prepend_cond_dim", 0) > 0:
        out["prepend_cond"] = torch.from_numpy(seeded.seeded_array((batch, 3, cfg["prepend_cond_dim"]), seed + 4))
        # the reference concatenates prepend masks unconditionally (dit.py:188) -> a mask must be supplied
        out["prepend_cond_mask"] = torch.ones(batch, 3, dtype=torch.bool)
    return out


def gen_dit(name, seed):
    from stable_audio_tools.models.dit import DiffusionTransformer
    cfg = seeded.DIT_CONFIGS[name]
    model = DiffusionTransformer(**cfg).float()
    model.train(False)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith("inv_freq")}
    sd = seeded.seeded_state_dict(shapes, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    inp = dit_inputs(name)
    with torch.no_grad():
        plain = model(inp["x"], inp["t"], cross_attn_cond=inp["cross_attn_cond"], global_embed=inp["global_embed"],
                      prepend_cond=inp.get("prepend_cond"), prepend_cond_mask=inp.get("prepend_cond_mask"),cfg_scale=1.0)
        guided = model(inp["x"], inp["t"], cross_attn_cond=inp["cross_attn_cond"], global_embed=inp["global_embed"],
                       prepend_cond=inp.get("prepend_cond"), prepend_cond_mask=inp.get("prepend_cond_mask"),cfg_scale=6.0, scale_phi=0.75)
        hidden = model(inp["x"], inp["t"], cross_attn_cond=inp["cross_attn_cond"], global_embed=inp["global_embed"],
                       prepend_cond=inp.get("prepend_cond"), prepend_cond_mask=inp.get("prepend_cond_mask"),return_info=True)[1]["hidden_states"]
    out = {"plain": plain.numpy(), "guided": guided.numpy(), "hidden_first": hidden[0].numpy(), "hidden_last": hidden[-1].numpy(),
           "keys": np.array(sorted(model.state_dict().keys()))}
    np.savez_compressed(os.path.join(OUT, f"dit_{name}.npz"), **out)
    print(f"dit_{name}: out {tuple(plain.shape)} |plain| {plain.abs().max():.3f} |guided| {guided.abs().max():.3f} params {sum(p.numel() for p in model.parameters())}")


def gen_disc(name, seed, batch=2, length=1500):
    """EncodecDiscriminator (models/discriminators.py:18-63 over models/encodec.py) through the reference classes: logits, feature
    maps and the three losses on seeded stereo signals, with autograd gradients w.r.t. the fake signal and every parameter."""
    from stable_audio_tools.models.discriminators import EncodecDiscriminator
    cfg = seeded.DISC_CONFIGS[name]
    disc = EncodecDiscriminator(**cfg).float()
    shapes = {k: tuple(v.shape) for k, v in disc.state_dict().items()}
    sd = seeded.seeded_state_dict(shapes, seed)
    disc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    reals = torch.from_numpy(seeded.seeded_array((batch, cfg["in_channels"], length), seed + 1, scale=0.3))
    fakes = (reals + torch.from_numpy(seeded.seeded_array((batch, cfg["in_channels"], length), seed + 2, scale=0.1))).requires_grad_(True)
    logits, fmaps = disc(fakes)
    dis, adv, fm = disc.loss(reals, fakes)
    total = dis + 0.1 * adv + 5.0 * fm
    names = [n for n, _ in disc.named_parameters()]
    grads = torch.autograd.grad(total, [fakes] + list(disc.parameters()))
    out = {"dis": dis.detach().numpy(), "adv": adv.detach().numpy(), "fm": fm.detach().numpy(), "grad/<fakes>": grads[0].numpy(),
           "keys": np.array(sorted(disc.state_dict().keys()))}
    for i, lg in enumerate(logits):
        out[f"logits/{i}"] = lg.detach().numpy()
        out[f"fmap_last/{i}"] = fmaps[i][-1].detach().numpy()
    for n, g in zip(names, grads[1:]):
        out["grad/" + n] = g.numpy()
    np.savez_compressed(os.path.join(OUT, f"disc_{name}.npz"), **out)
    print(f"disc_{name}: dis {dis.item():.5f} adv {adv.item():.5f} fm {fm.item():.5f} params {sum(p.numel() for p in disc.parameters())}")


def main():
    os.makedirs(OUT, exist_ok=True)
    refimport.import_reference()
    torch.set_num_threads(8)
    for i, name in enumerate(seeded.DIT_CONFIGS):
        gen_dit(name, seed=700 + 10 * i)
    gen_vae("tiny", batch=2, in_len=512, seed=100)
    gen_vae("mid", batch=1, in_len=1536, seed=200)
    gen_vae("mono", batch=2, in_len=320, seed=300)
    gen_chunked("tiny", seed=400)
    gen_stft(seed=500)
    gen_disc("tiny", seed=800)
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump({"generator": "oracle/gen_golden.py", "reference": "Stability-AI/stable-audio-tools v0.0.19 (/root/reference)",
                   "torch": torch.__version__, "numpy": np.__version__}, f, indent=1)


if __name__ == "__main__":
    main()
