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
