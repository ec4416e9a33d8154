"""Minimal in-tree samplers around the DiT call, restated from the reference so that the
"DiT sampling steps/s" metric can be measured without third-party k-diffusion
(stable_audio_tools/inference/sampling.py: get_alphas_sigmas :9-12, sample (v-DDIM) :254-307,
sample_discrete_euler :98-135).  The reference's own `generate_diffusion_cond`
(inference/generation.py:91) keeps working unchanged on top of the native modules — these loops
exist for bench.py and the parity tests.  Sampler-step fusion is a "next" row."""
import math

import torch


def get_alphas_sigmas(t):
    return torch.cos(t * math.pi / 2), torch.sin(t * math.pi / 2)


@torch.no_grad()
def sample_v_ddim(model, x, steps, eta=0.0, sigma_max=1.0, **extra_args):
    """v-objective DDIM (sampling.py:254-307).  `model(x, t, **extra_args)` returns v."""
    ts = x.new_ones([x.shape[0]])
    t = torch.linspace(sigma_max, 0, steps + 1)[:-1]
    alphas, sigmas = get_alphas_sigmas(t)
    pred = x
    for i in range(steps):
        v = model(x, ts * t[i], **extra_args)
        pred = x * alphas[i] - v * sigmas[i]
        eps = x * sigmas[i] + v * alphas[i]
        if i < steps - 1:
            ddim_sigma = eta * (sigmas[i + 1] ** 2 / sigmas[i] ** 2).sqrt() * (1 - alphas[i] ** 2 / alphas[i + 1] ** 2).sqrt()
            adjusted_sigma = (sigmas[i + 1] ** 2 - ddim_sigma ** 2).sqrt()
            x = pred * alphas[i + 1] + eps * adjusted_sigma
            if eta:
                x = x + torch.randn_like(x) * ddim_sigma
    return pred


@torch.no_grad()
def sample_discrete_euler(model, x, steps, sigma_max=1.0, **extra_args):
    """Rectified-flow Euler (sampling.py:98-135)."""
    t = torch.linspace(sigma_max, 0, steps + 1)
    for t_curr, t_prev in zip(t[:-1], t[1:]):
        tc = t_curr * torch.ones((x.shape[0],), dtype=x.dtype, device=x.device)
        x = x + (t_prev - t_curr) * model(x, tc, **extra_args)
    return x
