"""Minimal in-tree samplers around the DiT call, restated from the reference so that the
"DiT sampling steps/s" metric can be measured without third-party k-diffusion
(stable_audio_tools/inference/sampling.py: get_alphas_sigmas :9-12, sample (v-DDIM) :254-307,
sample_discrete_euler :98-135).  The reference's own `generate_diffusion_cond`
(inference/generation.py:91) keeps working unchanged on top of the native modules — these loops
exist for bench.py and the parity tests.

Sampler-step fusion: with a native DiffusionTransformer the per-step update of x (DDIM / Euler) rides in
the guidance-combine kernel (csrc/dit_ops.hip sat_cfg_step) through the model's `fused_update=` extension — one launch for
chunk + guidance + channel-std rescale + update instead of ~12 elementwise launches."""
import math

import torch


class GraphedDenoiser:
    """One denoiser evaluation `model(x, t, **extra_args)` captured into a HIP graph (torch.cuda.CUDAGraph) and replayed:
    a sampler step is ~600 short launches, and at batch 1-2 the host cannot issue them as fast as the GPU retires them.
    Every kernel of the path launches on torch's current stream (the capture stream while capturing) and allocates
    only through torch, so the whole forward — CFG batch doubling included — is capturable.  Inputs are copied into
    static buffers; the conditioning tensors are captured by reference (keep them alive and unchanged).  With a native
    DiffusionTransformer the fused sampler update rides in the captured step too: its four coefficients live in a device buffer
    the guidance kernel reads (sat_cfg_step_dev), rewritten before every replay."""

    def __init__(self, model, x, t, **extra_args):
        self.fused = bool(getattr(model, "supports_fused_update", False)) and not torch.is_grad_enabled()
        self.x = x.clone()
        self.t = t.clone()
        self.coef = torch.tensor([0.0, 1.0, 0.0, 0.0], device=x.device, dtype=torch.float32) if self.fused else None
        kw = dict(extra_args, fused_update=self.coef) if self.fused else dict(extra_args)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                    # warm-up outside capture (lazy initialisations, caches)
            for _ in range(2):
                model(self.x, self.t, **kw)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = model(self.x, self.t, **kw)

    def __call__(self, x, t, fused_update=None):
        """fused_update: a DEVICE tensor (c0x, c0v, c1x, c1v) (a row of the per-run coefficient table)."""
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x)
        self.t.copy_(t)
        if self.fused:
            self.coef.copy_(fused_update)
        self.graph.replay()
        return self.out


def _denoiser(model, x, use_graph, extra_args):
    if not use_graph:
        f = lambda xx, tt, **kw: model(xx, tt, **extra_args, **kw)      # noqa: E731
        f.fused = bool(getattr(model, "supports_fused_update", False)) and not torch.is_grad_enabled()
        return f
    return GraphedDenoiser(model, x, x.new_ones([x.shape[0]]), **extra_args)


def get_alphas_sigmas(t):
    return torch.cos(t * math.pi / 2), torch.sin(t * math.pi / 2)


@torch.no_grad()
def sample_v_ddim(model, x, steps, eta=0.0, sigma_max=1.0, use_graph=False, **extra_args):
    """v-objective DDIM (sampling.py:254-307).  `model(x, t, **extra_args)` returns v.  use_graph: replay the denoiser
    evaluation from a HIP graph (same arithmetic, no per-launch host cost)."""
    f = _denoiser(model, x, use_graph, extra_args)
    ts = x.new_ones([x.shape[0]])
    t = torch.linspace(sigma_max, 0, steps + 1)[:-1]
    alphas, sigmas = get_alphas_sigmas(t)
    pred = x
    if use_graph and f.fused and eta != 0.0:
        raise NotImplementedError("the graphed sampler step carries the fused (eta = 0) update")
    if f.fused and eta == 0.0:
        # pred = a_i x - s_i v;  eps = s_i x + a_i v;  x' = a' pred + s' eps  — both as linear combinations of (x, v)
        an = torch.cat([alphas[1:], alphas.new_ones(1)])
        sn = torch.cat([sigmas[1:], sigmas.new_zeros(1)])
        table = torch.stack([an * alphas + sn * sigmas, -an * sigmas + sn * alphas, alphas, -sigmas], dim=1).float()
        if use_graph:
            table_dev, tsteps = table.to(x.device), (ts[:, None] * t.to(x.device)[None, :]).t().contiguous()
            for i in range(steps):
                x, pred = f(x, tsteps[i], fused_update=table_dev[i])
            return pred.clone()
        for i in range(steps):
            x, pred = f(x, ts * t[i], fused_update=tuple(float(v) for v in table[i]))
        return pred
    for i in range(steps):
        v = f(x, ts * t[i])
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
def sample_discrete_euler(model, x, steps, sigma_max=1.0, use_graph=False, **extra_args):
    """Rectified-flow Euler (sampling.py:98-135)."""
    f = _denoiser(model, x, use_graph, extra_args)
    t = torch.linspace(sigma_max, 0, steps + 1)
    for t_curr, t_prev in zip(t[:-1], t[1:]):
        tc = t_curr * torch.ones((x.shape[0],), dtype=x.dtype, device=x.device)
        if f.fused:
            coef = (1.0, float(t_prev - t_curr), 0.0, 1.0)
            x, _ = f(x, tc, fused_update=torch.tensor(coef, device=x.device) if use_graph else coef)
            continue
        x = x + (t_prev - t_curr) * f(x, tc)
    return x.clone() if use_graph else x
