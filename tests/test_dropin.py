"""Drop-in test: the REFERENCE's own factories and sampler running on top of the native hot-path modules.

    stable_audio_tools_amd.patch_reference()          # rebinds DiffusionTransformer / Oobleck* / VAEBottleneck / ...
    create_model_from_config(diffusion_cond JSON)     # reference models/factory.py:3 -> models/diffusion.py:629
    model.load_state_dict(<state_dict of an all-reference model>)
    generate_diffusion_cond(model, ...)               # reference inference/generation.py:91 (rectified-flow Euler, CFG)

and the result must equal the all-reference run (same seed) to 1e-3.  The reference imports only in the build container
(/root/reference); elsewhere the test is skipped.  CPU run = the native modules executing on the host-side simulator.
"""
import contextlib
import copy
import sys
from unittest import mock

import pytest
import torch

import refimport
from golden_util import rel_err

pytestmark = pytest.mark.skipif(not refimport.available(), reason="no reference tree (/root/reference, or oracle/_ref staged by build())")

MODEL_CONFIG = {
    "model_type": "diffusion_cond", "sample_size": 512, "sample_rate": 16000, "audio_channels": 2,
    "model": {
        "pretransform": {"type": "autoencoder", "iterate_batch": False, "config": {
            "encoder": {"type": "oobleck", "config": {"in_channels": 2, "channels": 8, "c_mults": [1, 2], "strides": [2, 4],
                                                        "latent_dim": 8, "use_snake": True}},
            "decoder": {"type": "oobleck", "config": {"out_channels": 2, "channels": 8, "c_mults": [1, 2], "strides": [2, 4],
                                                        "latent_dim": 4, "use_snake": True, "final_tanh": False}},
            "bottleneck": {"type": "vae"}, "latent_dim": 4, "downsampling_ratio": 8, "io_channels": 2}},
        "conditioning": {"cond_dim": 64, "configs": [
            {"id": "seconds_start", "type": "number", "config": {"min_val": 0, "max_val": 512}},
            {"id": "seconds_total", "type": "number", "config": {"min_val": 0, "max_val": 512}}]},
        "diffusion": {"cross_attention_cond_ids": ["seconds_start", "seconds_total"], "global_cond_ids": ["seconds_total"],
                      "diffusion_objective": "rectified_flow", "type": "dit",
                      "config": {"io_channels": 4, "embed_dim": 128, "depth": 2, "num_heads": 2, "cond_token_dim": 64,
                                 "global_cond_dim": 64, "project_cond_tokens": False, "transformer_type": "continuous_transformer"}},
        "io_channels": 4},
}


def _randomise(model, seed):
    """De-zero the branch outputs the reference zero-initialises so that the comparison is not vacuous."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("to_out.weight") or ".ff.ff.2." in n or "process_conv" in n:
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 / p.shape[-1] ** 0.5 if p.dim() > 1 else 0.1))


@contextlib.contextmanager
def _cpu_noise():
    """generate_diffusion_cond draws its initial noise with torch.randn(..., device=device) right after torch.manual_seed(seed)
    (inference/generation.py:141-143): the CPU and the HIP generators give different streams for the same seed, so — for the
    comparison only — every torch.randn inside the call draws from the CPU stream and is moved to the requested device."""
    real = torch.randn

    def randn(*size, device=None, **kw):
        return real(*size, **kw).to(device) if device is not None else real(*size, **kw)
    with mock.patch.object(torch, "randn", randn):
        yield


def _dropin(device):
    """device: where the NATIVE model lives ("cpu" = the simulator, "cuda" = the gfx950 library); the all-reference model always
    runs the reference's CPU path."""
    import stable_audio_tools_amd
    from stable_audio_tools_amd import autoencoders as n_ae, bottleneck as n_bn, dit as n_dit, pretransforms as n_pt
    with contextlib.redirect_stdout(sys.stderr):
        refimport.import_reference()
    from stable_audio_tools.inference.generation import generate_diffusion_cond
    from stable_audio_tools.models.factory import create_model_from_config

    torch.manual_seed(0)
    ref_model = create_model_from_config(copy.deepcopy(MODEL_CONFIG)).train(False)
    _randomise(ref_model, 1)
    sd = {k: v.clone() for k, v in ref_model.state_dict().items()}

    handle = stable_audio_tools_amd.patch_reference()
    try:
        assert ("stable_audio_tools.models.diffusion", "DiffusionTransformer") in handle.applied
        assert ("stable_audio_tools.inference.generation", "sample_rf") in handle.applied
        nat_model = create_model_from_config(copy.deepcopy(MODEL_CONFIG)).train(False)
        # the reference factory really built the native classes
        assert isinstance(nat_model.model.model, n_dit.DiffusionTransformer)
        assert isinstance(nat_model.pretransform, n_pt.AutoencoderPretransform)
        assert isinstance(nat_model.pretransform.model.encoder, n_ae.OobleckEncoder)
        assert isinstance(nat_model.pretransform.model.decoder, n_ae.OobleckDecoder)
        assert isinstance(nat_model.pretransform.model.bottleneck, n_bn.VAEBottleneck)
        assert sorted(nat_model.state_dict().keys()) == sorted(sd.keys())
        nat_model.load_state_dict(sd)                       # strict: same keys, same shapes
        nat_model = nat_model.to(device)

        cond = [{"seconds_start": 0, "seconds_total": 3}, {"seconds_start": 1, "seconds_total": 7}]
        kw = dict(steps=4, cfg_scale=3.0, conditioning=cond, batch_size=2, sample_size=MODEL_CONFIG["sample_size"], seed=1234)
        out = {}
        with torch.no_grad(), _cpu_noise(), contextlib.redirect_stdout(sys.stderr):
            # while patched, the reference's generate_diffusion_cond samples through the native sample_rf (fused sampler step)
            for sampler in ("euler", "dpmpp"):
                out[sampler] = (generate_diffusion_cond(nat_model, device=device, sampler_type=sampler, **kw).cpu(),
                                generate_diffusion_cond(nat_model, device=device, sampler_type=sampler, return_latents=True, **kw).cpu())
    finally:
        handle.undo()
    with torch.no_grad(), contextlib.redirect_stdout(sys.stderr):
        for sampler in ("euler", "dpmpp"):
            ref_audio = generate_diffusion_cond(ref_model, device="cpu", sampler_type=sampler, **kw)
            ref_lat = generate_diffusion_cond(ref_model, device="cpu", sampler_type=sampler, return_latents=True, **kw)
            nat_audio, nat_lat = out[sampler]
            assert ref_audio.shape == nat_audio.shape == (2, 2, MODEL_CONFIG["sample_size"])
            assert rel_err(nat_lat, ref_lat) < 1e-3, sampler
            assert rel_err(nat_audio, ref_audio) < 1e-3, sampler
        # unpatched again: the reference's own sampler loop around the native model (class swap only)
        with _cpu_noise():
            nat_lat2 = generate_diffusion_cond(nat_model, device=device, sampler_type="euler", return_latents=True, **kw).cpu()
        assert rel_err(nat_lat2, generate_diffusion_cond(ref_model, device="cpu", sampler_type="euler", return_latents=True, **kw)) < 1e-3
    # pretransform.encode through the reference wrapper API (VAE draw from the CPU stream on both sides)
    audio = 0.3 * torch.randn(2, 2, 512, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        torch.manual_seed(77)
        zr = ref_model.pretransform.encode(audio)
        torch.manual_seed(77)
        noise = torch.randn(zr.shape)                        # VAEBottleneck.encode: torch.randn_like(stdev) * stdev + mean (models/bottleneck.py:118)
        zn = nat_model.pretransform.model.encode(audio.to(device), noise=noise.to(device)) / nat_model.pretransform.scale
    assert rel_err(zn.cpu(), zr) < 1e-3


def test_reference_factories_and_sampler_run_on_native_modules(emu_modules):
    _dropin("cpu")


@pytest.mark.gpu
def test_reference_factories_and_sampler_run_on_native_modules_gpu(hip):
    """The same on the MI355X: patch_reference() -> create_model_from_config -> load_state_dict -> generate_diffusion_cond
    (inference/generation.py:91) with the native modules on cuda:0, against the all-reference run on this box's host cores.
    The reference tree comes from oracle/_ref (staged by __graft_entry__.build(): /root/reference does not exist on the GPU box)."""
    _dropin("cuda")


def test_patch_reference_is_reversible():
    import stable_audio_tools_amd
    refimport.import_reference()
    import stable_audio_tools.models.dit as ref_dit
    original = ref_dit.DiffusionTransformer
    handle = stable_audio_tools_amd.patch_reference()
    assert ref_dit.DiffusionTransformer is not original
    handle.undo()
    assert ref_dit.DiffusionTransformer is original
