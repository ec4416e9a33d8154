"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product package).

CPU restatement (plain torch fp32 functional ops, no nn.Module, no autograd tricks) of the
reference's Oobleck autoencoder path.  It consumes a *reference-format state_dict* (the keys the
reference modules produce) plus the JSON config, so it is also a check of the key layout.
Pinned against golden vectors generated from the reference itself by oracle/gen_golden.py
(tests/test_oracle_golden.py).

Every function cites the reference lines it restates (paths relative to
/root/reference/stable_audio_tools/).
"""
import math

import torch
import torch.nn.functional as F


def weight_norm_fold(v, g):
    """torch.nn.utils.weight_norm(dim=0): w = g * v / ||v||, norm over all dims except 0.
    models/autoencoders.py:23-27 (WNConv1d / WNConvTranspose1d).  For ConvTranspose1d dim 0 is Cin."""
    n = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return v * (g / n)


def snake_beta(x, alpha_log, beta_log):
    """models/blocks.py:291-292 + :321-329 (alpha_logscale=True): x + sin^2(x e^a) / (e^b + 1e-9)."""
    a = torch.exp(alpha_log).view(1, -1, 1)
    b = torch.exp(beta_log).view(1, -1, 1)
    return x + (1.0 / (b + 0.000000001)) * torch.pow(torch.sin(x * a), 2)


def _wn_conv(sd, prefix, x, stride=1, padding=0, dilation=1):
    w = weight_norm_fold(sd[prefix + ".weight_v"], sd[prefix + ".weight_g"])
    return F.conv1d(x, w, sd.get(prefix + ".bias"), stride=stride, padding=padding, dilation=dilation)


def _wn_convtr(sd, prefix, x, stride, padding):
    w = weight_norm_fold(sd[prefix + ".weight_v"], sd[prefix + ".weight_g"])
    return F.conv_transpose1d(x, w, sd.get(prefix + ".bias"), stride=stride, padding=padding)


def residual_unit(sd, prefix, x, dilation):
    """models/autoencoders.py:58-83: x + conv1x1(snake(conv7_dil(snake(x)))), pad = 3*dil (symmetric)."""
    h = snake_beta(x, sd[prefix + ".layers.0.alpha"], sd[prefix + ".layers.0.beta"])
    h = _wn_conv(sd, prefix + ".layers.1", h, padding=(dilation * 6) // 2, dilation=dilation)
    h = snake_beta(h, sd[prefix + ".layers.2.alpha"], sd[prefix + ".layers.2.beta"])
    h = _wn_conv(sd, prefix + ".layers.3", h)
    return x + h


def encoder_block(sd, prefix, x, stride):
    """models/autoencoders.py:233-250: RU(d=1), RU(3), RU(9), snake, WNConv1d(k=2s, stride s, pad ceil(s/2))."""
    for i, d in enumerate((1, 3, 9)):
        x = residual_unit(sd, f"{prefix}.layers.{i}", x, d)
    x = snake_beta(x, sd[prefix + ".layers.3.alpha"], sd[prefix + ".layers.3.beta"])
    return _wn_conv(sd, prefix + ".layers.4", x, stride=stride, padding=math.ceil(stride / 2))


def decoder_block(sd, prefix, x, stride):
    """models/autoencoders.py:252-283: snake, WNConvTranspose1d(k=2s, stride s, pad ceil(s/2)), RU(1), RU(3), RU(9)."""
    x = snake_beta(x, sd[prefix + ".layers.0.alpha"], sd[prefix + ".layers.0.beta"])
    x = _wn_convtr(sd, prefix + ".layers.1", x, stride, math.ceil(stride / 2))
    for i, d in enumerate((1, 3, 9)):
        x = residual_unit(sd, f"{prefix}.layers.{2 + i}", x, d)
    return x


def oobleck_encoder(sd, cfg, x, prefix=""):
    """models/autoencoders.py:285-317.  cfg = the JSON 'encoder.config' dict."""
    strides = cfg.get("strides", [2, 4, 8, 8])
    depth = len(cfg.get("c_mults", [1, 2, 4, 8])) + 1
    p = prefix + "layers"
    x = _wn_conv(sd, f"{p}.0", x, padding=3)
    for i in range(depth - 1):
        x = encoder_block(sd, f"{p}.{1 + i}", x, strides[i])
    x = snake_beta(x, sd[f"{p}.{depth}.alpha"], sd[f"{p}.{depth}.beta"])
    return _wn_conv(sd, f"{p}.{depth + 1}", x, padding=1)


def oobleck_decoder(sd, cfg, z, prefix=""):
    """models/autoencoders.py:320-362.  cfg = the JSON 'decoder.config' dict."""
    strides = cfg.get("strides", [2, 4, 8, 8])
    depth = len(cfg.get("c_mults", [1, 2, 4, 8])) + 1
    p = prefix + "layers"
    x = _wn_conv(sd, f"{p}.0", z, padding=3)
    for n, i in enumerate(range(depth - 1, 0, -1)):
        x = decoder_block(sd, f"{p}.{1 + n}", x, strides[i - 1])
    x = snake_beta(x, sd[f"{p}.{depth}.alpha"], sd[f"{p}.{depth}.beta"])
    x = _wn_conv(sd, f"{p}.{depth + 1}", x, padding=3)
    if cfg.get("final_tanh", True):
        x = torch.tanh(x)
    return x


def vae_sample(mean, scale, noise):
    """models/bottleneck.py:105-113 with the randn draw made explicit."""
    stdev = F.softplus(scale) + 1e-4
    var = stdev * stdev
    logvar = torch.log(var)
    latents = noise * stdev + mean
    kl = (mean * mean + var - logvar - 1).sum(1).mean()
    return latents, kl


def autoencoder_encode(sd, model_cfg, audio, noise):
    """AudioAutoencoder.encode (models/autoencoders.py:446-491) + VAEBottleneck.encode (bottleneck.py:119-133).
    Returns (latents, kl, pre_bottleneck_latents)."""
    pre = oobleck_encoder(sd, model_cfg["encoder"]["config"], audio, prefix="encoder.")
    mean, scale = pre.chunk(2, dim=1)
    z, kl = vae_sample(mean, scale, noise)
    return z, kl, pre


def autoencoder_decode(sd, model_cfg, latents):
    """AudioAutoencoder.decode (models/autoencoders.py:493-534); VAEBottleneck.decode is identity (bottleneck.py:133)."""
    y = oobleck_decoder(sd, model_cfg["decoder"]["config"], latents, prefix="decoder.")
    if model_cfg["decoder"].get("soft_clip", False):
        y = torch.tanh(y)
    return y
