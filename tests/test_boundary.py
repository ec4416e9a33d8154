"""Drop-in boundary behaviour beyond plain forward parity:

  * `remove_weight_norm_from_model` (reference models/utils.py:31-37, used by train.py:73-81) on native convs, and both checkpoint
    flavours (weight_g / weight_v, or the folded `weight` written after removal) into both module forms;
  * derived-weight caches of frozen / no-grad passes: a frozen pretransform launches no sat_wn_fold / sat_pack / sat_snake_consts
    after its first call, and every cache follows weight updates the version counters do not see (fused AdamW kernel, `.data`
    edits as ema_pytorch makes them, torch optimizers);
  * the optional activation recompute of ResidualUnit / TransformerBlock gives the same results as keeping activations;
  * the native DiT under torch.inference_mode().
The bodies run on the simulator here and on the gfx950 library with `-m gpu`.
"""
import copy

import pytest
import torch

import seeded
from golden_util import build_native_ae, rel_err


def _ae_inputs(device, batch=2, n=512, seed=100):
    cfg = seeded.AE_CONFIGS["tiny"]
    audio = torch.from_numpy(seeded.seeded_array((batch, 2, n), seed + 1, scale=0.5)).to(device)
    noise = torch.from_numpy(seeded.seeded_array((batch, cfg["model"]["latent_dim"], n // 8), seed + 2)).to(device)
    return audio, noise


def _remove_weight_norm_from_model(model):
    """The reference helper when the checkout is importable, else the same loop (models/utils.py:31-37)."""
    import refimport
    if refimport.available():
        import contextlib
        import sys
        with contextlib.redirect_stdout(sys.stderr):
            refimport.import_reference()
            from stable_audio_tools.models.utils import remove_weight_norm_from_model
            return remove_weight_norm_from_model(model)
    from torch.nn.utils import remove_weight_norm
    for module in model.modules():
        if hasattr(module, "weight"):
            remove_weight_norm(module)
    return model


def _weight_norm_removal(device):
    from stable_audio_tools_amd.autoencoders import _WNConvBase
    model = build_native_ae("tiny", 100, device)
    audio, noise = _ae_inputs(device)
    wn_sd = copy.deepcopy(model.state_dict())
    with torch.no_grad():
        z0 = model.encode(audio, noise=noise)
        d0 = model.decode(z0)
    convs = [m for m in model.modules() if isinstance(m, _WNConvBase)]
    assert all(hasattr(m, "weight") and tuple(m.weight.shape) == tuple(m.weight_v.shape) for m in convs)
    _remove_weight_norm_from_model(model)                   # must not be a silent no-op
    assert all(m.is_folded for m in convs)
    keys = set(model.state_dict().keys())
    assert not any(k.endswith("weight_g") or k.endswith("weight_v") for k in keys)
    assert sum(k.endswith(".weight") for k in keys) == len(convs)
    with torch.no_grad():
        z1 = model.encode(audio, noise=noise)
        d1 = model.decode(z1)
    assert rel_err(z1, z0) < 1e-5 and rel_err(d1, d0) < 1e-5
    folded_sd = copy.deepcopy(model.state_dict())
    # (a) folded checkpoint into a fresh weight-normed model ("post-removal checkpoint")
    fresh = build_native_ae("tiny", 999, device)
    fresh.load_state_dict(folded_sd)
    with torch.no_grad():
        assert rel_err(fresh.decode(fresh.encode(audio, noise=noise)), d0) < 1e-5
    # (b) weight-normed checkpoint into a model whose weight norm was removed first (train.py "pre_load" order)
    pre = build_native_ae("tiny", 998, device)
    _remove_weight_norm_from_model(pre)
    pre.load_state_dict(wn_sd)
    with torch.no_grad():
        assert rel_err(pre.decode(pre.encode(audio, noise=noise)), d0) < 1e-5
    # folded parameters train: the gradient reaches `weight`
    loss = model.decode(model.encode(audio, noise=noise)).square().mean()
    loss.backward()
    assert all(m.weight.grad is not None and float(m.weight.grad.abs().sum()) > 0 for m in convs)


def test_weight_norm_removal_simulator(emu_modules):
    _weight_norm_removal("cpu")


@pytest.mark.gpu
def test_weight_norm_removal_gpu(hip):
    _weight_norm_removal("cuda")


class _Counter:
    """Counts calls of SatOps methods (weight fold / pack / snake constants)."""

    NAMES = ("wn_fold", "pack_bf16x3", "pack", "snake_consts")

    def __init__(self, ops):
        self.ops, self.n, self._orig = ops, {k: 0 for k in self.NAMES}, {}
        for k in self.NAMES:
            self._orig[k] = getattr(ops, k)
            setattr(ops, k, self._wrap(k))

    def _wrap(self, k):
        def f(*a, **kw):
            self.n[k] += 1
            return self._orig[k](*a, **kw)
        return f

    def total(self):
        return sum(self.n.values())

    def reset(self):
        self.n = {k: 0 for k in self.NAMES}

    def close(self):
        for k in self.NAMES:
            delattr(self.ops, k)        # instance attribute shadows the method: remove it


def _frozen_caches(device, ops):
    from stable_audio_tools_amd import _caches
    from stable_audio_tools_amd.pretransforms import AutoencoderPretransform
    model = build_native_ae("tiny", 100, device)
    pt = AutoencoderPretransform(model, scale=1.0)
    audio, noise = _ae_inputs(device)
    cnt = _Counter(ops)
    try:
        z = pt.encode(audio, noise=noise)
        d0 = pt.decode(z).clone()
        assert cnt.total() > 0
        cnt.reset()
        d1 = pt.decode(z)
        pt.encode(audio, noise=noise)
        assert cnt.total() == 0, cnt.n                      # frozen: folded + packed weights and snake constants are kept
        assert torch.equal(d0, d1)
        # an in-place edit through torch bumps the version counter -> refold
        conv = model.decoder.layers[0]
        with torch.no_grad():
            conv.weight_g.mul_(1.25)
        d2 = pt.decode(z)
        assert cnt.n["wn_fold"] == 1 and not torch.equal(d2, d0)
        # a `.data` edit (ema_pytorch: ma_params.data.lerp_/copy_) is invisible to the version counter: the epoch covers it
        v_before = conv.weight_g._version
        conv.weight_g.data.mul_(0.8)
        assert conv.weight_g._version == v_before
        _caches.invalidate_weight_caches()
        cnt.reset()
        d3 = pt.decode(z)
        assert cnt.n["wn_fold"] > 0 and rel_err(d3, d0) < 1e-5        # 1.25 * 0.8 = 1
        # any torch optimizer step invalidates (global post-step hook)
        e0 = _caches.weight_epoch()
        p = torch.nn.Parameter(torch.zeros(3))
        p.grad = torch.ones(3)
        torch.optim.SGD([p], lr=0.1).step()
        assert _caches.weight_epoch() > e0
        # training pass (grad enabled, trainable parameters): no caching, gradients flow to weight_g / weight_v
        model.requires_grad_(True)
        loss = model.decode(model.encode(audio, noise=noise)).square().mean()
        loss.backward()
        assert conv.weight_v.grad is not None and conv.weight_g.grad is not None
    finally:
        cnt.close()


def test_frozen_pretransform_caches_simulator(emu_modules):
    _frozen_caches("cpu", emu_modules)


@pytest.mark.gpu
def test_frozen_pretransform_caches_gpu(hip):
    _frozen_caches("cuda", hip)


def _residual_recompute(device):
    from stable_audio_tools_amd.autoencoders import ResidualUnit
    model = build_native_ae("tiny", 100, device)
    audio, noise = _ae_inputs(device)

    def grads():
        model.zero_grad(set_to_none=True)
        dec = model.decode(model.encode(audio, noise=noise))
        (dec * dec).mean().backward()
        return dec.detach().clone(), [p.grad.clone() for p in model.parameters()]

    d0, g0 = grads()
    ResidualUnit.checkpointing = True
    try:
        d1, g1 = grads()
    finally:
        ResidualUnit.checkpointing = False
    assert torch.equal(d0, d1)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)            # same kernels on the same inputs: bit-identical


def test_residual_unit_recompute_simulator(emu_modules):
    _residual_recompute("cpu")


@pytest.mark.gpu
def test_residual_unit_recompute_gpu(hip):
    _residual_recompute("cuda")


# ------------------------------------------------------------------------------------------------ DiT
def _dit(name, seed, device, dtype=torch.float32):
    from test_dit_parity import _build
    return _build(name, seed, device, dtype=dtype)


def _dit_inputs(name, device, dtype=None):
    from gen_golden import dit_inputs
    inp = {k: v.to(device) for k, v in dit_inputs(name).items()}
    if dtype is not None:
        inp = {k: (v.to(dtype) if v.is_floating_point() and k != "t" else v) for k, v in inp.items()}
    return inp


def _dit_cache_follows_training(device):
    """no_grad forward, one DiTTrainStep (the fused AdamW kernel rewrites the weights without touching torch's version
    counters), no_grad forward with the SAME conditioning tensor: must equal a fresh-cache evaluation."""
    from stable_audio_tools_amd.dit import clear_inference_caches
    from stable_audio_tools_amd.training import DiTTrainStep
    name = "tiny_prepend"
    model, _ = _dit(name, 720, device)
    inp = _dit_inputs(name, device)
    x, t, g, cond = inp["x"], inp["t"], inp["global_embed"], inp["cross_attn_cond"]
    step = DiTTrainStep(model, lr=1e-2, use_ema=False, autocast_dtype=torch.bfloat16)

    def demo(fresh=False):
        if fresh:
            clear_inference_caches(model)
        with torch.no_grad(), torch.autocast("cuda" if device == "cuda" else "cpu", dtype=torch.bfloat16):
            return model(x, t, cross_attn_cond=cond, global_embed=g, cfg_scale=3.0).float().clone()

    a = demo(fresh=True)
    step(x, cross_attn_cond=cond, global_embed=g)
    b = demo()                       # same cond object, weights changed by the HIP optimizer kernel
    c = demo(fresh=True)
    assert torch.equal(b, c)
    assert not torch.equal(a, b)


def test_dit_caches_follow_fused_optimizer_simulator(emu_modules):
    _dit_cache_follows_training("cpu")


@pytest.mark.gpu
def test_dit_caches_follow_fused_optimizer_gpu(hip):
    _dit_cache_follows_training("cuda")


def _dit_data_edit_and_inference_mode(device):
    from stable_audio_tools_amd import _caches
    name = "tiny_adaln"
    model, _ = _dit(name, 730, device, dtype=torch.bfloat16)
    inp = _dit_inputs(name, device)
    x, t, g = inp["x"], inp["t"], inp["global_embed"]
    cond = inp["cross_attn_cond"].to(torch.bfloat16)

    def run():
        with torch.no_grad():
            return model(x, t, cross_attn_cond=cond, global_embed=g, cfg_scale=4.0).float().clone()

    a = run()
    # ema_pytorch-style update of every parameter through .data (version counters do not move)
    with torch.no_grad():
        for p in model.parameters():
            p.data.mul_(1.01)
    _caches.invalidate_weight_caches()           # what the wrapped EMA.update / the optimizer post-step hook do
    b = run()
    from stable_audio_tools_amd.dit import clear_inference_caches
    clear_inference_caches(model)
    for m in model.modules():
        if hasattr(m, "_cache"):
            m._cache.items.clear()
        if hasattr(m, "_f32_cache"):
            m._f32_cache.clear()
    c = run()
    assert torch.equal(b, c) and not torch.equal(a, b)
    # torch.inference_mode: conditioning tensors are inference tensors (no version counter) -> caches are bypassed, not a crash
    with torch.inference_mode():
        ci = cond.clone()
        d1 = model(x.clone(), t.clone(), cross_attn_cond=ci, global_embed=g.clone(), cfg_scale=4.0).float()
        ci.mul_(2.0)                            # untracked in-place edit must be seen
        d2 = model(x.clone(), t.clone(), cross_attn_cond=ci, global_embed=g.clone(), cfg_scale=4.0).float()
    assert rel_err(d1, b) < 1e-6
    assert not torch.equal(d1, d2)


def test_dit_data_edit_and_inference_mode_simulator(emu_modules):
    _dit_data_edit_and_inference_mode("cpu")


@pytest.mark.gpu
def test_dit_data_edit_and_inference_mode_gpu(hip):
    _dit_data_edit_and_inference_mode("cuda")


def _transformer_checkpointing(device):
    name = "tiny_adaln"
    model, _ = _dit(name, 740, device)
    inp = _dit_inputs(name, device)
    x, t, g, cond = inp["x"], inp["t"], inp["global_embed"], inp["cross_attn_cond"]

    def grads():
        model.zero_grad(set_to_none=True)
        out = model(x, t, cross_attn_cond=cond, global_embed=g)
        out.square().mean().backward()
        return out.detach().clone(), [p.grad.clone() for p in model.parameters() if p.grad is not None]

    o0, g0 = grads()
    model.transformer.checkpointing = True
    o1, g1 = grads()
    model.transformer.checkpointing = False
    assert torch.equal(o0, o1) and len(g0) == len(g1)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)


def test_transformer_checkpointing_simulator(emu_modules):
    _transformer_checkpointing("cpu")


@pytest.mark.gpu
def test_transformer_checkpointing_gpu(hip):
    _transformer_checkpointing("cuda")
