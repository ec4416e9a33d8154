"""Deterministic, version-stable test data (numpy legacy RandomState) shared by oracle/gen_golden.py
(which feeds it to the reference) and by the tests (which feed the same data to the oracle and the
HIP path).  TEST INFRASTRUCTURE ONLY."""
import zlib

import numpy as np

# JSON surfaces exactly as the reference factory consumes them
# (configs/model_configs/autoencoders/stable_audio_2_0_vae.json layout, scaled down).
AE_CONFIGS = {
    # BASELINE.json configs[0]: frozen tiny Oobleck pretransform — stereo variant
    "tiny": {
        "model_type": "autoencoder", "sample_size": 512, "sample_rate": 16000, "audio_channels": 2,
        "model": {
            "encoder": {"type": "oobleck", "config": {"in_channels": 2, "channels": 8, "c_mults": [1, 2], "strides": [2, 4],
                                                        "latent_dim": 8, "use_snake": True}},
            "decoder": {"type": "oobleck", "config": {"out_channels": 2, "channels": 8, "c_mults": [1, 2], "strides": [2, 4],
                                                        "latent_dim": 4, "use_snake": True, "final_tanh": False}},
            "bottleneck": {"type": "vae"}, "latent_dim": 4, "downsampling_ratio": 8, "io_channels": 2},
    },
    # three levels, strides 2/4/8, channel counts that are not multiples of the 128-wide tile
    "mid": {
        "model_type": "autoencoder", "sample_size": 1536, "sample_rate": 44100, "audio_channels": 2,
        "model": {
            "encoder": {"type": "oobleck", "config": {"in_channels": 2, "channels": 12, "c_mults": [1, 2, 4], "strides": [2, 4, 8],
                                                        "latent_dim": 12, "use_snake": True}},
            "decoder": {"type": "oobleck", "config": {"out_channels": 2, "channels": 12, "c_mults": [1, 2, 4], "strides": [2, 4, 8],
                                                        "latent_dim": 6, "use_snake": True, "final_tanh": True}},
            "bottleneck": {"type": "vae"}, "latent_dim": 6, "downsampling_ratio": 64, "io_channels": 2},
    },
    # the C1 plumbing case of BASELINE.json: mono 16 kHz, ratio 8, 4 latent channels
    "mono": {
        "model_type": "autoencoder", "sample_size": 320, "sample_rate": 16000, "audio_channels": 1,
        "model": {
            "encoder": {"type": "oobleck", "config": {"in_channels": 1, "channels": 8, "c_mults": [1, 2], "strides": [2, 4],
                                                        "latent_dim": 8, "use_snake": True}},
            "decoder": {"type": "oobleck", "config": {"out_channels": 1, "channels": 8, "c_mults": [1, 2], "strides": [2, 4],
                                                        "latent_dim": 4, "use_snake": True, "final_tanh": False}},
            "bottleneck": {"type": "vae"}, "latent_dim": 4, "downsampling_ratio": 8, "io_channels": 1},
    },
}

# training.loss_configs.spectral.config of stable_audio_2_0_vae.json:93-103
STFT_CFG = {
    "fft_sizes": [2048, 1024, 512, 256, 128, 64, 32],
    "hop_sizes": [512, 256, 128, 64, 32, 16, 8],
    "win_lengths": [2048, 1024, 512, 256, 128, 64, 32],
    "perceptual_weighting": True,
}


# DiffusionTransformer kwargs (the `model.diffusion.config` JSON block).  BASELINE.json configs[0]:
# tiny DiT, 4 layers, d=256, 4 heads of 64, 64-d context -> GQA 4 q-heads / 1 kv-head, run with both
# global-conditioning modes.
DIT_CONFIGS = {
    "tiny_prepend": dict(io_channels=4, embed_dim=256, depth=4, num_heads=4, cond_token_dim=64, global_cond_dim=64,
                         project_cond_tokens=False, global_cond_type="prepend"),
    "tiny_adaln": dict(io_channels=4, embed_dim=256, depth=4, num_heads=4, cond_token_dim=64, global_cond_dim=64,
                       project_cond_tokens=False, global_cond_type="adaLN"),
    # projected context (full multi-head cross attention), rectified-flow objective, prepend conditioning tokens
    "small_rf": dict(io_channels=6, embed_dim=128, depth=2, num_heads=2, cond_token_dim=48, global_cond_dim=32,
                     prepend_cond_dim=24, project_cond_tokens=True, global_cond_type="prepend",
                     diffusion_objective="rectified_flow"),
}


def seeded_array(shape, seed, scale=1.0):
    rs = np.random.RandomState(seed)
    return (rs.standard_normal(size=shape) * scale).astype(np.float32)


def seeded_state_dict(shapes, seed):
    """shapes: {key: shape}.  Values depend only on (key, shape, seed)."""
    out = {}
    for key, shape in shapes.items():
        rs = np.random.RandomState((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 31))
        if key.endswith("weight_v"):
            fan = int(np.prod(shape[1:]))
            a = rs.standard_normal(size=shape) / np.sqrt(fan)
        elif key.endswith("weight_g"):
            a = 0.3 + 0.2 * np.abs(rs.standard_normal(size=shape))
        elif key.endswith("bias"):
            a = 0.1 * rs.standard_normal(size=shape)
        elif key.endswith("alpha") or key.endswith("beta"):
            a = 0.3 * rs.standard_normal(size=shape)
        elif key.endswith("gamma"):
            a = 1.0 + 0.1 * rs.standard_normal(size=shape)
        elif key.endswith("to_scale_shift_gate"):
            a = rs.standard_normal(size=shape) / np.sqrt(shape[0] / 6)
        elif key.endswith("timestep_features.weight"):
            a = rs.standard_normal(size=shape)
        elif key.endswith(".weight") and len(shape) >= 2:
            # nn.Linear / 1x1 Conv1d: includes the branches the reference zero-initialises (to_out, ff.2,
            # pre/postprocess_conv) so that parity tests are not vacuous
            a = rs.standard_normal(size=shape) / np.sqrt(int(np.prod(shape[1:])))
        else:
            a = rs.standard_normal(size=shape)
        out[key] = a.astype(np.float32)
    return out


# ---------------------------------------------------------------------------------------------------------------
# Full-width cases (BASELINE.json configs[1] / configs[2] at their real widths, short crops) — oracle/gen_golden_full.py
# ---------------------------------------------------------------------------------------------------------------
import json as _json
import os as _os

_CFG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "stable_audio_tools_amd", "configs")

# stable_audio_2_0_vae architecture on a 32768-sample stereo crop (16 latent frames), batch 1
# fwd_eps: forward accuracy class of the bf16x3 split-MFMA conv stack (2^-17 per product, measured 8e-6 on the decoded audio) —
# the size of the perturbation with which gen_golden_full.py probes the conditioning of the generator-loss gradient
FULL_VAE = {"seed": 2100, "batch": 1, "length": 32768, "kl_weight": 1e-4, "fwd_eps": 1e-5}
# Stable Audio Open DiT block stack: d=1536, 24 x 64 heads, GQA 24:12 cross-attention, N = 1 + 1024 tokens, M = 130
FULL_DIT = {"seed": 2300, "latent_length": 1024, "context_length": 130,
            "config": dict(io_channels=64, embed_dim=1536, depth=24, num_heads=24, cond_token_dim=768, global_cond_dim=1536,
                           project_cond_tokens=False, transformer_type="continuous_transformer")}
FULL_KEEP_NUMEL = 4096       # gradients up to this size are stored whole, larger ones as norm + a seeded probe
FULL_PROBE = 1024


def full_vae_config():
    """The shipped stable_audio_2_0_vae.json (reference configs/model_configs/autoencoders/stable_audio_2_0_vae.json)."""
    with open(_os.path.join(_CFG_DIR, "stable_audio_2_0_vae.json")) as f:
        return _json.load(f)


def full_vae_inputs():
    """(audio, vae noise, projection) of the full-width VAE case."""
    b, n, s = FULL_VAE["batch"], FULL_VAE["length"], FULL_VAE["seed"]
    return (seeded_array((b, 2, n), s + 1, scale=0.1), seeded_array((b, 64, n // 2048), s + 2), seeded_array((b, 2, n), s + 3))


def probe_index(name, numel, count=FULL_PROBE):
    """Seeded sample of flat indices of a large gradient tensor (depends only on the parameter name and size)."""
    rs = np.random.RandomState(zlib.crc32(name.encode()) % (2 ** 31))
    return np.sort(rs.choice(numel, size=min(count, numel), replace=False))


# MS-STFT discriminator (training.loss_configs.discriminator.config of stable_audio_2_0_vae.json:80-87, scaled down)
DISC_CONFIGS = {
    "tiny": dict(filters=8, in_channels=2, n_ffts=[128, 64, 32], hop_lengths=[32, 16, 8], win_lengths=[128, 64, 32]),
}
