"""VAE bottleneck on the HIP path (reference: stable_audio_tools/models/bottleneck.py:6-23, :105-134)."""
import torch
from torch import nn

from . import functional as Fn


class Bottleneck(nn.Module):
    def __init__(self, is_discrete: bool = False):
        super().__init__()
        self.is_discrete = is_discrete

    def encode(self, x, return_info=False, **kwargs):
        raise NotImplementedError

    def decode(self, x):
        raise NotImplementedError


class VAEBottleneck(Bottleneck):
    """encode: (B, 2C, T) -> z = randn*softplus(scale)+1e-4 .. + mean, info['kl'].

    Like the reference it samples on EVERY call, inference included (bottleneck.py:109, :124).  The
    N(0,1) draw comes from ``torch.randn_like`` (same generator the reference uses), or from the
    ``noise=`` kwarg so parity tests can inject it; the arithmetic runs in csrc/elementwise.hip."""

    def __init__(self):
        super().__init__(is_discrete=False)

    def encode(self, x, return_info=False, noise=None, **kwargs):
        info = {}
        c = x.shape[1] // 2
        if noise is None:
            noise = torch.randn_like(x[:, :c])
        z, kl = Fn.VaeSampleFn.apply(x, noise)
        info["kl"] = kl
        if return_info:
            return z, info
        return z

    def decode(self, x):
        return x
