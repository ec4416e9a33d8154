"""The reference's in-tree samplers (stable_audio_tools/inference/sampling.py) around the native DiT, with the per-step update
of x folded into the guidance kernel.

Same names, arguments and results as the reference functions:

    DistributionShift          :24-41     time-shift of the schedule by sequence length
    sample_discrete_euler      :98-135    rectified-flow Euler
    sample_rk4                 :138-177   4th-order Runge-Kutta (4 model evaluations per step)
    sample_flow_dpmpp          :179-219   DPM-Solver++(2M) for rectified-flow models
    sample_flow_pingpong       :222-250   ping-pong sampling for distilled models
    sample                     :254-307   v-objective DDIM (eta, cfg_pp)
    sample_rf                  :395-446   schedule + dispatch used by generate_diffusion_cond (inference/generation.py:206)
    sample_k (v-ddim types)    :334-391   the two sampler types that do not need third-party k-diffusion

`sigmas`, `callback`, `dist_shift`, `cfg_pp`, `disable_tqdm` mean what they mean there (progress bars are not drawn).  The
reference's own `generate_diffusion_cond` keeps working unchanged on top of the native modules (patch.patch_reference) — these
functions are the same arithmetic with the sampler-step fusion on:

Sampler-step fusion.  Every update rule above is LINEAR in (x, v, one more tensor, the unconditioned output), so with a native
DiffusionTransformer the whole step — chunk + guidance combine + channel-std rescale + update of x (+ the `denoised` / `pred`
the callback wants) — is ONE launch, csrc/dit_ops.hip sat_sampler_step, through the model's `fused_update=` extension:
    y0 = c0x*x + c0v*v + c0p*p + c0u*u        y1 = c1x*x + c1v*v + c1p*p + c1u*u
The coefficient tables are computed on the host with the reference's own expressions (same torch ops on the same fp32
schedule tensors), so only the final multiply-adds differ in rounding.  `use_graph=True` replays the denoiser evaluation from a
HIP graph (the coefficients then live in a device buffer the kernel reads).  Extra native arguments (never passed by reference
callers): `use_graph`, `noise_fn` (where the fresh noise of eta > 0 / ping-pong comes from; default torch.randn_like).
"""
import math

import torch


class DistributionShift:
    """inference/sampling.py:24-41."""

    def __init__(self, base_shift=0.5, max_shift=1.15, max_length=4096, min_length=256, use_sine=False):
        self.base_shift = base_shift
        self.max_shift = max_shift
        self.max_length = max_length
        self.min_length = min_length
        self.use_sine = use_sine

    def time_shift(self, t: torch.Tensor, seq_len: int):
        seq_len = min(max(seq_len, self.min_length), self.max_length)
        sigma = 1.0
        mu = -(self.base_shift + (self.max_shift - self.base_shift) * (seq_len - self.min_length) / (self.max_length - self.min_length))
        t_out = 1 - math.exp(mu) / (math.exp(mu) + (1 / (1 - t) - 1) ** sigma)
        if self.use_sine:
            t_out = torch.sin(t_out * math.pi / 2)
        return t_out


def get_alphas_sigmas(t):
    return torch.cos(t * math.pi / 2), torch.sin(t * math.pi / 2)


def t_to_alpha_sigma(t):
    return torch.cos(t * math.pi / 2), torch.sin(t * math.pi / 2)


def alpha_sigma_to_t(alpha, sigma):
    return torch.atan2(sigma, alpha) / math.pi * 2


# ---------------------------------------------------------------------------------------------------------------------------
# the denoiser evaluation: plain, fused (one launch for guidance + update), graphed
# ---------------------------------------------------------------------------------------------------------------------------
def _supports_fused(model):
    """The native DiffusionTransformer advertises `supports_fused_update`; the reference's DiTWrapper around it
    (models/diffusion.py:506-557 — what generate_diffusion_cond hands to the samplers as `model.model`) forwards unknown keyword
    arguments to the transformer unchanged, so the fused step works through it too."""
    if torch.is_grad_enabled():
        return False
    if getattr(model, "supports_fused_update", False):
        return True
    inner = getattr(model, "model", None)
    return type(model).__name__ == "DiTWrapper" and bool(getattr(inner, "supports_fused_update", False))


def _coef8(c0=(0.0, 1.0, 0.0, 0.0), c1=(0.0, 0.0, 0.0, 0.0)):
    return tuple(float(v) for v in c0) + tuple(float(v) for v in c1)


class GraphedDenoiser:
    """One denoiser evaluation `model(x, t, **extra_args)` captured into a HIP graph (torch.cuda.CUDAGraph) and replayed:
    a sampler step is ~600 short launches, and at batch 1-2 the host cannot issue them as fast as the GPU retires them.
    Every kernel of the path launches on torch's current stream (the capture stream while capturing) and allocates
    only through torch, so the whole forward — CFG batch doubling included — is capturable.  Inputs are copied into
    static buffers; the conditioning tensors are captured by reference (keep them alive and unchanged).  With a native
    DiffusionTransformer the fused sampler update rides in the captured step too: its eight coefficients live in a device buffer
    the guidance kernel reads (sat_sampler_step_dev), rewritten before every replay; `base` (the x the update is applied to when
    it is not the model input: RK4 stages) and `prev` (third operand) are static buffers as well."""

    def __init__(self, model, x, t, with_prev=False, with_base=False, **extra_args):
        self.fused = _supports_fused(model)
        self.x = x.clone()
        self.t = t.clone()
        self.coef = torch.tensor(_coef8(), device=x.device, dtype=torch.float32) if self.fused else None
        self.prev = torch.zeros_like(x) if (self.fused and with_prev) else None
        self.base = x.clone() if (self.fused and with_base) else None
        kw = dict(extra_args)
        if self.fused:
            kw.update(fused_update=self.coef, fused_prev=self.prev, fused_x=self.base)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                    # warm-up outside capture (lazy initialisations, caches)
            for _ in range(2):
                model(self.x, self.t, **kw)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = model(self.x, self.t, **kw)

    def __call__(self, x, t, fused_update=None, fused_prev=None, fused_x=None):
        """fused_update: 8 coefficients (host sequence or device tensor).  Returns the graph's static output buffers."""
        if x.data_ptr() != self.x.data_ptr():
            self.x.copy_(x)
        self.t.copy_(t)
        if self.fused:
            if isinstance(fused_update, torch.Tensor):
                self.coef.copy_(fused_update)
            else:
                self.coef.copy_(torch.tensor(fused_update, dtype=torch.float32))
            if self.prev is not None and fused_prev is not None and fused_prev.data_ptr() != self.prev.data_ptr():
                self.prev.copy_(fused_prev)
            if self.base is not None and fused_x is not None and fused_x.data_ptr() != self.base.data_ptr():
                self.base.copy_(fused_x)
        self.graph.replay()
        return self.out


def _denoiser(model, x, use_graph, extra_args, with_prev=False, with_base=False):
    if not use_graph:
        f = lambda xx, tt, **kw: model(xx, tt, **extra_args, **kw)      # noqa: E731
        f.fused = _supports_fused(model)
        f.graphed = False
        return f
    g = GraphedDenoiser(model, x, x.new_ones([x.shape[0]]), with_prev=with_prev, with_base=with_base, **extra_args)
    g.graphed = True
    return g


def _schedule(x, steps, sigma_max, sigmas, dist_shift):
    """The `t` of sample_discrete_euler / rk4 / dpmpp / pingpong (sampling.py:109-121 and twins)."""
    assert steps is not None or sigmas is not None, "Either steps or sigmas must be provided"
    if sigmas is not None:
        return sigmas
    t = torch.linspace(sigma_max, 0, steps + 1)
    if dist_shift is not None:
        t = dist_shift.time_shift(t, x.shape[-1])
    return t


def _tvec(x, t_scalar):
    """`t_curr * ones(B)` on x's device; the scalar stays on the host schedule tensor."""
    return torch.full((x.shape[0],), float(t_scalar), dtype=x.dtype, device=x.device)


def _table(f, rows, device):
    """rows: one 8-tuple per fused launch.  For a graphed denoiser: ONE (n, 8) device tensor built before the loop — a step then copies
    its row device-to-device into the buffer the captured kernel reads (no host->device copy, no sync inside the loop)."""
    if getattr(f, "graphed", False) and f.fused:
        return torch.tensor(rows, dtype=torch.float32, device=device)
    return rows


def _own(t, graphed):
    """Results handed to a callback / returned: a graphed denoiser's outputs are static buffers overwritten by the next replay."""
    return t.clone() if graphed else t


# ---------------------------------------------------------------------------------------------------------------------------
# rectified-flow samplers
# ---------------------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def sample_discrete_euler(model, x, steps=None, sigma_max=1, sigmas=None, callback=None, dist_shift=None, disable_tqdm=False,
                          use_graph=False, **extra_args):
    """Rectified-flow Euler (sampling.py:98-135): x <- x + (t_prev - t_curr) * v."""
    t = _schedule(x, steps, sigma_max, sigmas, dist_shift)
    f = _denoiser(model, x, use_graph, extra_args)
    # y0 = x + dt v;  y1 = denoised = x_new - t_prev v = x + (dt - t_prev) v
    rows = _table(f, [_coef8((1.0, tp - tc, 0, 0), (1.0, (tp - tc) - tp, 0, 0)) for tc, tp in zip(t[:-1], t[1:])], x.device)
    for i, (t_curr, t_prev) in enumerate(zip(t[:-1], t[1:])):
        dt = t_prev - t_curr
        tc = _tvec(x, t_curr)
        if f.fused:
            x, den = f(x, tc, fused_update=rows[i])
        else:
            v = f(x, tc)
            x = x + dt * v
            den = x - t_prev * v if callback is not None else None
        if callback is not None:
            callback({'x': _own(x, f.graphed), 't': t_curr, 'sigma': t_curr, 'i': i + 1, 'denoised': _own(den, f.graphed)})
    return _own(x, f.graphed)


@torch.no_grad()
def sample_rk4(model, x, steps=None, sigma_max=1, sigmas=None, callback=None, dist_shift=None, use_graph=False, **extra_args):
    """4th-order Runge-Kutta (sampling.py:138-177).  Fused: the stage inputs x + c*dt*k_j and the running sum k1 + 2 k2 + 2 k3 are
    the two outputs of the stage's own guidance launch (third operand = the running sum)."""
    t = _schedule(x, steps, sigma_max, sigmas, dist_shift)
    f = _denoiser(model, x, use_graph, extra_args, with_prev=True, with_base=True)
    rows = []
    for t_curr, t_prev in zip(t[:-1], t[1:]):
        dt = t_prev - t_curr
        h, s6 = dt / 2, dt / 6
        rows += [_coef8((1.0, h, 0, 0), (0, 1.0, 0, 0)),                   # x + dt/2 k1 ; k1
                 _coef8((1.0, h, 0, 0), (0, 2.0, 1.0, 0)),                 # x + dt/2 k2 ; acc + 2 k2
                 _coef8((1.0, dt, 0, 0), (0, 2.0, 1.0, 0)),                # x + dt k3   ; acc + 2 k3
                 _coef8((1.0, s6, s6, 0), (1.0, s6 - t_prev, s6, 0))]      # x' = x + dt/6 (acc + k4) ; denoised = x' - t_prev k4
    rows = _table(f, rows, x.device)
    for i, (t_curr, t_prev) in enumerate(zip(t[:-1], t[1:])):
        dt = t_prev - t_curr
        tc, tm, tp = _tvec(x, t_curr), _tvec(x, t_curr + dt / 2), _tvec(x, t_prev)
        if f.fused:
            x2, acc = f(x, tc, fused_update=rows[4 * i], fused_x=x)
            x2, acc = _own(x2, f.graphed), _own(acc, f.graphed)
            x3, acc = f(x2, tm, fused_update=rows[4 * i + 1], fused_x=x, fused_prev=acc)
            x3, acc = _own(x3, f.graphed), _own(acc, f.graphed)
            x4, acc = f(x3, tm, fused_update=rows[4 * i + 2], fused_x=x, fused_prev=acc)
            x4, acc = _own(x4, f.graphed), _own(acc, f.graphed)
            xn, den = f(x4, tp, fused_update=rows[4 * i + 3], fused_x=x, fused_prev=acc)
            x = _own(xn, f.graphed)
        else:
            k1 = f(x, tc)
            k2 = f(x + dt / 2 * k1, tm)
            k3 = f(x + dt / 2 * k2, tm)
            k4 = f(x + dt * k3, tp)
            x = x + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
            den = x - t_prev * k4 if callback is not None else None
        if callback is not None:
            callback({'x': x, 't': t_curr, 'sigma': t_curr, 'i': i + 1, 'denoised': _own(den, f.graphed)})
    return x


@torch.no_grad()
def sample_flow_dpmpp(model, x, steps=None, sigma_max=1, sigmas=None, callback=None, dist_shift=None, disable_tqdm=False,
                      use_graph=False, **extra_args):
    """DPM-Solver++(2M) for rectified-flow models (sampling.py:179-219).  denoised = x - t_curr v;
    x' = (t_next / t_curr) x - (1 - t_next) expm1(-h) D,  D = denoised (first / last step) or the 2M extrapolation
    (1 + 1/2r) denoised - (1/2r) old_denoised: linear in (x, v, old_denoised) -> one fused launch per step."""
    t = _schedule(x, steps, sigma_max, sigmas, dist_shift)
    f = _denoiser(model, x, use_graph, extra_args, with_prev=True)
    log_snr = lambda tt: ((1 - tt) / tt).log()      # noqa: E731
    plan = []                                        # per step: (ratio, gain, wd, wo) — the reference's own expressions on the schedule
    for i in range(len(t) - 1):
        t_curr, t_next = t[i], t[i + 1]
        alpha_t = 1 - t_next
        h = log_snr(t_next) - log_snr(t_curr)
        gain = -alpha_t * (-h).expm1()               # multiplies D
        if i == 0 or t_next == 0:
            wd, wo = 1.0, 0.0
        else:
            h_last = log_snr(t_curr) - log_snr(t[i - 1])
            r = h_last / h
            wd, wo = 1 + 1 / (2 * r), -(1 / (2 * r))
        plan.append((t_next / t_curr, gain, wd, wo))
    # D = wd (x - t_curr v) + wo old  ->  x' = (ratio + gain wd) x - gain wd t_curr v + gain wo old;  y1 = denoised = x - t_curr v
    rows = _table(f, [_coef8((ratio + gain * wd, -gain * wd * t[i], gain * wo, 0), (1.0, -t[i], 0, 0))
                      for i, (ratio, gain, wd, wo) in enumerate(plan)], x.device)
    old = None
    for i in range(len(t) - 1):
        t_curr = t[i]
        ratio, gain, wd, wo = plan[i]
        tc = _tvec(x, t_curr)
        if f.fused:
            xn, den = f(x, tc, fused_update=rows[i], fused_prev=old)
            den = _own(den, f.graphed)
            if callback is not None:
                callback({'x': x, 'i': i, 't': t_curr, 'sigma': t_curr, 'denoised': den})
            x = _own(xn, f.graphed)
        else:
            den = x - t_curr * f(x, tc)
            if callback is not None:
                callback({'x': x, 'i': i, 't': t_curr, 'sigma': t_curr, 'denoised': den})
            x = ratio * x + gain * (wd * den + (wo * old if (old is not None and float(wo) != 0.0) else 0.0))
        old = den
    return x


@torch.no_grad()
def sample_flow_pingpong(model, x, steps=None, sigma_max=1, sigmas=None, callback=None, dist_shift=None, use_graph=False, noise_fn=None,
                         **extra_args):
    """Ping-pong sampling for distilled models (sampling.py:222-250): denoised = x - t_i v; x' = (1 - t_next) denoised + t_next * noise.
    noise_fn(x) -> fresh noise (default torch.randn_like, as the reference draws it)."""
    t = _schedule(x, steps, sigma_max, sigmas, dist_shift)
    noise_fn = noise_fn or torch.randn_like
    f = _denoiser(model, x, use_graph, extra_args, with_prev=True)
    # x' = (1 - t_next) (x - t_i v) + t_next noise;  y1 = denoised
    rows = _table(f, [_coef8(((1 - t[i + 1]), -(1 - t[i + 1]) * t[i], t[i + 1], 0), (1.0, -t[i], 0, 0)) for i in range(len(t) - 1)], x.device)
    for i in range(len(t) - 1):
        t_i, t_next = t[i], t[i + 1]
        tc = _tvec(x, t_i)
        if f.fused:
            # the reference draws the noise AFTER the model call; nothing else consumes the stream in between, so the draw is the same
            noise = noise_fn(x)
            xn, den = f(x, tc, fused_update=rows[i], fused_prev=noise)
            if callback is not None:
                callback({'x': x, 'i': i, 't': t_i, 'sigma': t_i, 'sigma_hat': t_i, 'denoised': _own(den, f.graphed)})
            x = _own(xn, f.graphed)
            continue
        den = x - t_i * f(x, tc)
        if callback is not None:
            callback({'x': x, 'i': i, 't': t_i, 'sigma': t_i, 'sigma_hat': t_i, 'denoised': den})
        x = (1 - t_next) * den + t_next * noise_fn(x)
    return x


# ---------------------------------------------------------------------------------------------------------------------------
# v-objective DDIM
# ---------------------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def sample(model, x, steps, eta, callback=None, sigma_max=1.0, dist_shift=None, cfg_pp=False, use_graph=False, noise_fn=None, **extra_args):
    """v-diffusion DDIM (sampling.py:254-307).  `model(x, t, **extra_args)` returns v.
    pred = a_i x - s_i v;  eps = s_i x + a_i v_eps  (v_eps = the unconditioned output under cfg_pp, else v);
    x' = a' pred + adj eps (+ ddim_sigma * noise for eta > 0) — linear in (x, v, noise, u): one fused launch per step."""
    noise_fn = noise_fn or torch.randn_like
    t = torch.linspace(sigma_max, 0, steps + 1)[:-1]
    if dist_shift is not None:
        t = dist_shift.time_shift(t, x.shape[-1])
    alphas, sigmas = get_alphas_sigmas(t)
    f = _denoiser(model, x, use_graph, extra_args, with_prev=bool(eta))
    ts = x.new_ones([x.shape[0]])
    sched = []                     # per step: (ddim_sigma, adjusted_sigma) — sampling.py:289-292
    for i in range(steps - 1):
        ddim_sigma = eta * (sigmas[i + 1] ** 2 / sigmas[i] ** 2).sqrt() * (1 - alphas[i] ** 2 / alphas[i + 1] ** 2).sqrt()
        sched.append((ddim_sigma, (sigmas[i + 1] ** 2 - ddim_sigma ** 2).sqrt()))
    rows = []
    for i in range(steps):
        a, s = alphas[i], sigmas[i]
        if i == steps - 1:
            c0 = (1.0, 0.0, 0.0, 0.0)                  # x is not advanced on the last step (:286)
        else:
            an, (ddim_sigma, adj) = alphas[i + 1], sched[i]
            if cfg_pp:
                c0 = (an * a + adj * s, -an * s, ddim_sigma if eta else 0.0, adj * a)       # eps takes the unconditioned output
            else:
                c0 = (an * a + adj * s, -an * s + adj * a, ddim_sigma if eta else 0.0, 0.0)
        rows.append(_coef8(c0, (a, -s, 0, 0)))
    rows = _table(f, rows, x.device)
    pred = x
    for i in range(steps):
        last = i == steps - 1
        if f.fused:
            noise = noise_fn(x) if (eta and not last) else None
            xn, pred = f(x, _tvec(x, t[i]), fused_update=rows[i], fused_prev=noise)
            pred = _own(pred, f.graphed) if (callback is not None or last) else pred
            if not last:
                x = _own(xn, f.graphed) if callback is not None else xn
        else:
            if cfg_pp:
                v, info = f(x, ts * t[i], return_info=True)
                v_eps = info["uncond_output"] if "uncond_output" in info else v
            else:
                v = f(x, ts * t[i])
                v_eps = v
            pred = x * alphas[i] - v * sigmas[i]
            eps = x * sigmas[i] + v_eps * alphas[i]
            if not last:
                ddim_sigma, adjusted_sigma = sched[i]
                x = pred * alphas[i + 1] + eps * adjusted_sigma
                if eta:
                    x = x + noise_fn(x) * ddim_sigma
        if callback is not None:
            callback({'x': x, 't': t[i], 'sigma': sigmas[i], 'i': i, 'denoised': pred})
    return pred


def sample_v_ddim(model, x, steps, eta=0.0, sigma_max=1.0, use_graph=False, **extra_args):
    """`sample` under the name bench.py and earlier rounds' tests use."""
    return sample(model, x, steps, eta, sigma_max=sigma_max, use_graph=use_graph, **extra_args)


# ---------------------------------------------------------------------------------------------------------------------------
# dispatchers used by generate_diffusion_cond (inference/generation.py:200-206)
# ---------------------------------------------------------------------------------------------------------------------------
def sample_rf(model_fn, noise, init_data=None, steps=100, sampler_type="euler", sigma_max=1, device="cuda", callback=None, cond_fn=None,
              **extra_args):
    """sampling.py:395-446: the logSNR-uniform schedule of the rectified-flow samplers and the dispatch on sampler_type."""
    if sigma_max > 1:
        sigma_max = 1
    if cond_fn is not None:
        raise NotImplementedError("cond_fn (classifier guidance through k-diffusion utilities) is out of scope")
    if init_data is not None:
        x = init_data * (1 - sigma_max) + noise * sigma_max          # variation: interpolate init data and noise
    else:
        x = noise
    logsnr_max = math.log(((1 - sigma_max) / sigma_max) + 1e-6) if sigma_max < 1 else -6
    logsnr = torch.linspace(logsnr_max, 2, steps + 1)
    t = torch.sigmoid(-logsnr)
    t[0] = sigma_max
    t[-1] = 0
    if sampler_type == "euler":
        return sample_discrete_euler(model_fn, x, sigmas=t, sigma_max=sigma_max, callback=callback, **extra_args)
    elif sampler_type == "rk4":
        return sample_rk4(model_fn, x, steps, sigma_max, callback=callback, **extra_args)
    elif sampler_type == "dpmpp":
        return sample_flow_dpmpp(model_fn, x, sigmas=t, sigma_max=sigma_max, callback=callback, **extra_args)
    elif sampler_type == "pingpong":
        return sample_flow_pingpong(model_fn, x, sigmas=t, sigma_max=sigma_max, callback=callback, **extra_args)
    raise ValueError(f"Unknown sampler_type: {sampler_type}")


def sample_k(model_fn, noise, init_data=None, steps=100, sampler_type="v-ddim", sigma_min=0.01, sigma_max=100, rho=1.0, device="cuda",
             callback=None, cond_fn=None, **extra_args):
    """sampling.py:334-391 for the sampler types that live in the reference itself ("v-ddim", "v-ddim-cfgpp").  The k-diffusion
    types ("dpmpp-2m-sde", "k-heun", ...) call the third-party package around `model_fn` — use the reference's own sample_k for
    those (the native DiT is a drop-in `model_fn`); here they raise."""
    if sampler_type not in ("v-ddim", "v-ddim-cfgpp"):
        raise NotImplementedError(f"sampler_type {sampler_type!r} needs third-party k-diffusion (out of scope): call the reference's "
                                  "sample_k with the native model")
    if cond_fn is not None:
        raise NotImplementedError("cond_fn (classifier guidance through k-diffusion utilities) is out of scope")
    if sigma_max > 1:
        sigma_max = 1
    alpha, sigma = t_to_alpha_sigma(torch.tensor(sigma_max))
    x = init_data * alpha + noise * sigma if init_data is not None else noise
    return sample(model_fn, x, steps, eta=0.0, sigma_max=sigma_max, cfg_pp=(sampler_type == "v-ddim-cfgpp"), callback=callback, **extra_args)
