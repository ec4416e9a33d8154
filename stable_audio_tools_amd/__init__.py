"""MI355X-native denoising hot path of stable-audio-tools (DiT, Oobleck VAE, MR-STFT, DDP) behind the reference's
module API.  `patch_reference()` plugs it into an importable reference checkout (see patch.py)."""


def patch_reference(*args, **kwargs):
    from .patch import patch_reference as _p
    return _p(*args, **kwargs)
