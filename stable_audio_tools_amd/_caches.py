"""One invalidation epoch for every DERIVED copy of a parameter this package keeps: bf16 / bf16x3 / fp8 / transposed weight
copies (linear._WeightCache), folded + packed conv weights (autoencoders._WNConvBase), fp32 LayerNorm parameters, and the
no-grad inference caches of the DiT (embedded conditioning, CFG batch, cross-attention K / V planes).

A cache entry is valid for one (storage, torch version counter, epoch) of its sources.  torch's version counter sees every
in-place op on the parameter itself; it does NOT see
  * the fused optimizer kernel writing the flat parameter buffer (training.FusedAdamW.step),
  * updates made through `.data` — ema_pytorch's `ma_params.data.lerp_()` / `.copy_()`, which the reference's training
    wrappers use for the EMA models they demo / validate with (training/diffusion.py:58, :240-247; training/autoencoders.py:262-270).
Both are covered by the epoch: it is bumped by FusedAdamW.step, by EVERY torch.optim optimizer step (a global
`register_optimizer_step_post_hook`: an EMA update always follows an optimizer step, with no forward of the EMA model in
between), by `ema_pytorch.EMA.update` itself when that package is importable (patch.patch_reference wraps it), and by
`invalidate_weight_caches()` for any other out-of-band edit of `.data`.
"""
import torch

_EPOCH = 0
_HOOKED = False


def weight_epoch():
    return _EPOCH


def bump_weight_epoch(*_args, **_kwargs):
    """Drop every derived weight copy and inference cache (they are rebuilt on next use)."""
    global _EPOCH
    _EPOCH += 1


invalidate_weight_caches = bump_weight_epoch


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
    register_optimizer_step_post_hook(bump_weight_epoch)
    _HOOKED = True


def wrap_ema_update(ema_cls):
    """Make `ema_cls.update` (ema_pytorch.EMA) bump the epoch after it rewrote the EMA model through `.data` (idempotent)."""
    if getattr(ema_cls.update, "_sat_wrapped", False):
        return
    orig = ema_cls.update

    def update(self, *a, **kw):
        out = orig(self, *a, **kw)
        bump_weight_epoch()
        return out

    update._sat_wrapped = True
    ema_cls.update = update


install_optimizer_hook()
