"""MI355X-native denoising hot path of stable-audio-tools (DiT, Oobleck VAE, MR-STFT, DDP) behind the reference's
module API.  `patch_reference()` plugs it into an importable reference checkout (see patch.py)."""


def patch_reference(*args, **kwargs):
    from .patch import patch_reference as _p
    return _p(*args, **kwargs)


def invalidate_weight_caches():
    """Call after editing parameters of a native module through `.data` (or any other way torch's version counters do not see)
    outside an optimizer step / ema_pytorch update: drops every derived weight copy and inference cache (_caches.py)."""
    from ._caches import invalidate_weight_caches as _i
    return _i()
