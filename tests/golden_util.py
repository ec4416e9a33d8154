"""Helpers shared by the CPU (simulator) and GPU parity tests.  TEST INFRASTRUCTURE ONLY."""
import os

import numpy as np
import torch

import seeded

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def rel_err(a, b):
    """max |a-b| / max |b|  — the 1e-3 'rel fp32' bar of BASELINE.json is applied to this."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def ae_state_dict(name, seed, shapes):
    sd = seeded.seeded_state_dict(shapes, seed)
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def build_native_ae(name, seed, device="cpu"):
    """Product AudioAutoencoder for AE_CONFIGS[name] with the seeded reference-format state_dict."""
    from stable_audio_tools_amd.autoencoders import create_autoencoder_from_config
    model = create_autoencoder_from_config(seeded.AE_CONFIGS[name])
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(ae_state_dict(name, seed, shapes))
    return model.to(device)
