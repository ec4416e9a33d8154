"""Registry that swaps the MI355X-native hot-path modules into the reference package, so that the
reference's own factories, samplers and training wrappers run unchanged on top of them.

    import stable_audio_tools                     # the reference (Stability-AI/stable-audio-tools)
    import stable_audio_tools_amd
    stable_audio_tools_amd.patch_reference()
    model = stable_audio_tools.models.factory.create_model_from_config(model_config)   # native DiT / Oobleck inside
    model.load_state_dict(reference_checkpoint)                                        # same keys and shapes
    audio = stable_audio_tools.inference.generation.generate_diffusion_cond(model, ...)

The reference has no plugin interface: its factories resolve class names in their own
module namespace at call time —
  models/diffusion.py:507-519   DiTWrapper.__init__            -> DiffusionTransformer
  models/autoencoders.py:783-865 create_encoder/decoder_from_config -> OobleckEncoder / OobleckDecoder
  models/factory.py:32-54        create_pretransform_from_config    -> pretransforms.AutoencoderPretransform
  models/factory.py:89-99        create_bottleneck_from_config      -> bottleneck.VAEBottleneck
  training/losses/auraloss.py    STFTLoss / MultiResolutionSTFTLoss / SumAndDifferenceSTFTLoss (training/autoencoders.py:142-146)
  inference/generation.py:6, :200-206  sample_k / sample_rf   -> the in-tree samplers with the fused sampler step (sampling.py)
— so rebinding those names is the whole integration.  Everything else (ConditionedDiffusionModelWrapper,
conditioners, AudioAutoencoder's chunking glue, samplers, Lightning wrappers) stays the reference's code and
only ever calls the module API the native classes mirror.
"""
import importlib
import sys


def _registry():
    from . import auraloss, autoencoders, bottleneck, discriminators, dit, pretransforms, sampling, transformer
    return [
        # the in-tree samplers with the fused sampler step: what generate_diffusion_cond[_inpaint] call
        # (inference/generation.py:200-206); sample_k is rebound separately below (its k-diffusion types stay the reference's)
        ("inference.generation", "sample_rf", sampling.sample_rf),
        ("inference.sampling", "sample_rf", sampling.sample_rf),
        ("inference.sampling", "sample_discrete_euler", sampling.sample_discrete_euler),
        ("inference.sampling", "sample_rk4", sampling.sample_rk4),
        ("inference.sampling", "sample_flow_dpmpp", sampling.sample_flow_dpmpp),
        ("inference.sampling", "sample_flow_pingpong", sampling.sample_flow_pingpong),
        ("inference.sampling", "sample", sampling.sample),
        # (reference module, attribute, native object)
        ("models.dit", "DiffusionTransformer", dit.DiffusionTransformer),
        ("models.diffusion", "DiffusionTransformer", dit.DiffusionTransformer),
        ("models.transformer", "ContinuousTransformer", transformer.ContinuousTransformer),
        ("models.dit", "ContinuousTransformer", transformer.ContinuousTransformer),
        ("models.autoencoders", "OobleckEncoder", autoencoders.OobleckEncoder),
        ("models.autoencoders", "OobleckDecoder", autoencoders.OobleckDecoder),
        ("models.bottleneck", "VAEBottleneck", bottleneck.VAEBottleneck),
        ("models.pretransforms", "AutoencoderPretransform", pretransforms.AutoencoderPretransform),
        ("models.autoencoders", "AutoencoderPretransform", pretransforms.AutoencoderPretransform),
        ("training.losses.auraloss", "STFTLoss", auraloss.STFTLoss),
        ("training.losses.auraloss", "MultiResolutionSTFTLoss", auraloss.MultiResolutionSTFTLoss),
        ("training.losses.auraloss", "SumAndDifferenceSTFTLoss", auraloss.SumAndDifferenceSTFTLoss),
        ("training.losses", "MultiResolutionSTFTLoss", auraloss.MultiResolutionSTFTLoss),
        ("training.losses", "SumAndDifferenceSTFTLoss", auraloss.SumAndDifferenceSTFTLoss),
        ("training.autoencoders", "MultiResolutionSTFTLoss", auraloss.MultiResolutionSTFTLoss),
        ("training.autoencoders", "SumAndDifferenceSTFTLoss", auraloss.SumAndDifferenceSTFTLoss),
        # the MS-STFT discriminator of the autoencoder training step
        ("models.discriminators", "EncodecDiscriminator", discriminators.EncodecDiscriminator),
        ("training.autoencoders", "EncodecDiscriminator", discriminators.EncodecDiscriminator),
        ("models.encodec", "MultiScaleSTFTDiscriminator", discriminators.MultiScaleSTFTDiscriminator),
    ]


class PatchHandle:
    """What patch_reference() returns: `.applied` lists (module, attribute) pairs that were rebound, `.skipped` the
    ones whose reference module is not importable in this environment (e.g. the training package without
    pytorch_lightning); `.undo()` restores the reference's own classes."""

    def __init__(self):
        self.applied, self.skipped, self._saved = [], [], []

    def undo(self):
        for mod, attr, old in reversed(self._saved):
            setattr(mod, attr, old)
        self._saved.clear()
        self.applied.clear()


def _sample_k_dispatch(ref_sample_k):
    """generation.sample_k while patched: the two sampler types that live in the reference itself ("v-ddim", "v-ddim-cfgpp") run on
    the native loop with the fused step; the k-diffusion types stay the reference's own code around the (native) model."""
    from . import sampling

    def sample_k(model_fn, noise, init_data=None, steps=100, sampler_type="dpmpp-2m-sde", *args, **kwargs):
        if sampler_type in ("v-ddim", "v-ddim-cfgpp") and not args and kwargs.get("cond_fn") is None:
            return sampling.sample_k(model_fn, noise, init_data, steps, sampler_type, **kwargs)
        return ref_sample_k(model_fn, noise, init_data, steps, sampler_type, *args, **kwargs)
    return sample_k


def patch_reference(package="stable_audio_tools", import_missing=("models", "inference")):
    """Rebind the hot-path class names inside the reference package `package` to the native classes.

    Modules of the reference that are already imported are always patched; those under the sub-packages named
    in `import_missing` are imported first (the models package is safe to import; `training` drags in
    pytorch_lightning and is only patched when the host program has imported it)."""
    handle = PatchHandle()
    for sub, attr, native in _registry():
        name = f"{package}.{sub}"
        mod = sys.modules.get(name)
        if mod is None and sub.split(".")[0] in import_missing:
            try:
                mod = importlib.import_module(name)
            except Exception:   # noqa: BLE001 — optional third-party imports of the reference
                mod = None
        if mod is None or not hasattr(mod, attr):
            handle.skipped.append((name, attr))
            continue
        handle._saved.append((mod, attr, getattr(mod, attr)))
        setattr(mod, attr, native)
        handle.applied.append((name, attr))
    gen = sys.modules.get(f"{package}.inference.generation")
    if gen is not None and hasattr(gen, "sample_k"):
        handle._saved.append((gen, "sample_k", gen.sample_k))
        gen.sample_k = _sample_k_dispatch(gen.sample_k)
        handle.applied.append((f"{package}.inference.generation", "sample_k"))
    # the reference's training wrappers keep EMA copies of the native models and update them through `.data` (ema_pytorch:
    # training/diffusion.py:58, :240-247; training/autoencoders.py:262-270) — invisible to torch's version counters, so every
    # EMA update also bumps the invalidation epoch of the derived-weight / inference caches (_caches.py)
    try:
        from ema_pytorch import EMA
        from . import _caches
        _caches.wrap_ema_update(EMA)
        handle.applied.append(("ema_pytorch", "EMA.update"))
    except ImportError:
        handle.skipped.append(("ema_pytorch", "EMA.update"))
    if not [a for a in handle.applied if a[0] != "ema_pytorch"]:
        raise RuntimeError(f"patch_reference: nothing to patch — is the reference package {package!r} importable?")
    return handle
