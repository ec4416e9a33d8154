"""Invalidation epochs for every DERIVED copy of a parameter this package keeps: bf16 / bf16x3 / fp8 / transposed weight
copies (linear._WeightCache), folded + packed conv weights (autoencoders._WNConvBase), fp32 LayerNorm parameters, and the
no-grad inference caches of the DiT (embedded conditioning, CFG batch, cross-attention K / V planes).

A cache entry is valid for one (storage, torch version counter, epoch) of its sources.  torch's version counter sees every
in-place op on the parameter itself; it does NOT see
  * the fused optimizer kernel writing the flat parameter buffer (training.FusedAdamW.step),
  * updates made through `.data` — ema_pytorch's `ma_params.data.lerp_()` / `.copy_()`, which the reference's training
    wrappers use for the EMA models they demo / validate with (training/diffusion.py:58, :240-247; training/autoencoders.py:262-270).
Both are covered by epochs, kept PER PARAMETER STORAGE (keyed on data_ptr): an optimizer step bumps the epochs of the
parameters THAT optimizer owns — a frozen pretransform next to a training DiT keeps its folded / packed weights, the DiT's own
copies are rebuilt — via FusedAdamW.step (its flat buffer's parameters), a global `register_optimizer_step_post_hook` (every
torch.optim optimizer: its param_groups), and `ema_pytorch.EMA.update` (the EMA model's parameters; wrapped at import time
when that package is importable, and again by patch.patch_reference).  `invalidate_weight_caches()` bumps a global epoch
for any other out-of-band edit of `.data`.  `epoch_of(*tensors)` is the key component caches use.
"""
import torch

_EPOCH = 0
_PARAM_EPOCH = {}        # data_ptr -> epoch of the storage that starts there (only ever grows: a recycled address costs one rebuild)
_HOOKED = False


def weight_epoch():
    return _EPOCH


def epoch_of(*tensors):
    """Cache-key component for copies derived from `tensors`: (global epoch, per-storage epochs)."""
    return (_EPOCH,) + tuple(_PARAM_EPOCH.get(t.data_ptr(), 0) for t in tensors if t is not None)


def bump_weight_epoch(*_args, **_kwargs):
    """Drop EVERY derived weight copy and inference cache (they are rebuilt on next use)."""
    global _EPOCH
    _EPOCH += 1


invalidate_weight_caches = bump_weight_epoch


def bump_params(params):
    """The storages of `params` were rewritten behind torch's version counters: drop what was derived from them (and only that)."""
    for p in params:
        if p is not None:
            k = p.data_ptr()
            _PARAM_EPOCH[k] = _PARAM_EPOCH.get(k, 0) + 1


def _optimizer_post_hook(optimizer, *_args, **_kwargs):
    """Per-storage bumps for plain leaf parameters; anything else in a group (a view or a tied alias whose data_ptr differs from the
    tensor the modules read) cannot be tracked per storage, so the GLOBAL epoch moves — the conservative behaviour of rounds 1–3."""
    plain = True
    for group in optimizer.param_groups:
        ps = group["params"]
        bump_params(ps)
        for p in ps:
            if p is not None and (p._base is not None or not p.is_leaf):
                plain = False
    if not plain:
        bump_weight_epoch()


def version_of(t):
    """torch's in-place version counter of `t`; inference tensors (torch.inference_mode) do not track one."""
    if t is None:
        return -1
    if t.is_inference():
        return -2
    return t._version


def trackable(*tensors):
    """False when one of the tensors is an inference tensor: it can be edited in place without any trace, so nothing derived
    from it may be cached."""
    return not any(t is not None and t.is_inference() for t in tensors)


def install_optimizer_hook():
    """Bump the epoch after every torch.optim.Optimizer.step() in this process (idempotent)."""
    global _HOOKED
    if _HOOKED:
        return
    from torch.optim.optimizer import register_optimizer_step_post_hook
    register_optimizer_step_post_hook(_optimizer_post_hook)
    _HOOKED = True


def wrap_ema_update(ema_cls):
    """Make `ema_cls.update` (ema_pytorch.EMA) bump the epoch after it rewrote the EMA model through `.data` (idempotent)."""
    if getattr(ema_cls.update, "_sat_wrapped", False):
        return
    orig = ema_cls.update

    def update(self, *a, **kw):
        out = orig(self, *a, **kw)
        model = getattr(self, "ema_model", None)
        if model is not None:
            bump_params(model.parameters())
            bump_params(model.buffers())
        else:
            bump_weight_epoch()
        return out

    update._sat_wrapped = True
    ema_cls.update = update


install_optimizer_hook()
try:        # outside patch_reference too: an EMA-model forward between optimizer.step and the `.data` EMA update must not see stale copies
    from ema_pytorch import EMA as _EMA
    wrap_ema_update(_EMA)
except Exception:      # noqa: BLE001 — an optional third-party package that fails to import (for whatever reason) must not break this one
    pass
