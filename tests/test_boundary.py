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
    # (a') train.py `--remove-pretransform-weight-norm post_load` on that model: it is folded already (by the load) — nothing left to
    # fold, no crash, same outputs; a second removal finds no weight norm, as torch reports for its own modules
    _remove_weight_norm_from_model(fresh)
    assert all(m.is_folded for m in fresh.modules() if isinstance(m, _WNConvBase))
    with torch.no_grad():
        assert rel_err(fresh.decode(fresh.encode(audio, noise=noise)), d0) < 1e-5
    with pytest.raises(ValueError, match="weight_norm of 'weight' not found"):
        torch.nn.utils.remove_weight_norm(next(m for m in fresh.modules() if isinstance(m, _WNConvBase)))
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


def _vae_inference_mode(device):
    """torch.inference_mode on the conv stack and the discriminator: derived weights made inside inference_mode
    are inference tensors without a version counter — they must bypass the caches / plane side channels, not crash, and give the
    no_grad results; calls before and after (ordinary tensors) keep working."""
    from stable_audio_tools_amd.autoencoders import WNConv1d
    from stable_audio_tools_amd.discriminators import EncodecDiscriminator
    model = build_native_ae("tiny", 100, device)
    audio, noise = _ae_inputs(device)
    fresh_conv = WNConv1d(4, 4, kernel_size=7, padding=3).to(device)       # a layer whose FIRST call is under inference_mode
    xs = torch.from_numpy(seeded.seeded_array((1, 4, 64), 5)).to(device)
    with torch.inference_mode():
        za = model.encode(audio, noise=noise)
        da = model.decode(za)
        za2 = model.encode(audio, noise=noise)                              # second call: whatever was kept must still be valid
        da2 = model.decode(za2)
        ya = fresh_conv(xs)
        ya2 = fresh_conv(xs)
    with torch.no_grad():
        zb = model.encode(audio, noise=noise)
        db = model.decode(zb)
        yb = fresh_conv(xs)
    assert rel_err(za, zb) < 1e-6 and rel_err(da, db) < 1e-6 and rel_err(za2, zb) < 1e-6 and rel_err(da2, db) < 1e-6
    assert rel_err(ya, yb) < 1e-6 and rel_err(ya2, yb) < 1e-6
    torch.manual_seed(3)
    disc = EncodecDiscriminator(in_channels=2, filters=4, n_ffts=[64, 32], hop_lengths=[16, 8], win_lengths=[64, 32]).to(device)
    with torch.inference_mode():
        la = [float(v) for v in disc.loss(audio, db.clone())]
        la2 = [float(v) for v in disc.loss(audio, db.clone())]
    with torch.no_grad():
        lb = [float(v) for v in disc.loss(audio, db)]
    for a, a2, b in zip(la, la2, lb):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)) and abs(a2 - b) <= 1e-5 * max(1.0, abs(b))


def test_vae_inference_mode_simulator(emu_modules):
    _vae_inference_mode("cpu")


@pytest.mark.gpu
def test_vae_inference_mode_gpu(hip):
    _vae_inference_mode("cuda")


def _emitted_planes_follow_in_place_edits(device, ops):
    """The producer -> consumer plane side channel (ops._note_emitted / _take_emitted) is valid for ONE version of the producer's
    output: a forward hook that edits it in place must make the consuming k7 conv rebuild its planes (it used to
    multiply the OLD values while the residual added the new ones)."""
    from stable_audio_tools_amd.autoencoders import EncoderBlock
    torch.manual_seed(0)
    blk = EncoderBlock(128, 256, stride=2, use_snake=True).to(device)     # units at C = 128: the k7q + emission path
    x = torch.from_numpy(seeded.seeded_array((1, 128, 512), 21, scale=0.5)).to(device)
    h = blk.layers[0].register_forward_hook(lambda m, i, o: o.add_(0.5))
    try:
        prev = ops.k7_emit
        with torch.no_grad():
            ops.k7_emit = True
            y_emit = blk(x)
            ops.k7_emit = False
            y_plain = blk(x)
    finally:
        ops.k7_emit = prev
        h.remove()
    assert rel_err(y_emit, y_plain) < 1e-6


def test_emitted_planes_follow_in_place_edits_simulator(emu_modules):
    _emitted_planes_follow_in_place_edits("cpu", emu_modules)


@pytest.mark.gpu
def test_emitted_planes_follow_in_place_edits_gpu(hip):
    _emitted_planes_follow_in_place_edits("cuda", hip)


def _pretransform_model_half(device):
    """AutoencoderPretransform(model_half=True) (models/pretransforms.py:48-71): fp16 parameter storage and fp16-rounded inputs /
    outputs as the reference, fp32-accurate arithmetic in between.  Checked three ways: (1) state_dict dtypes are fp16; (2) equal —
    up to the output rounding — to the fp32 native model whose weights and inputs were rounded to fp16 by hand; (3) when the
    reference is importable: within fp16 resolution of the reference's own half pretransform on CPU, and closer to the fp32 result."""
    from stable_audio_tools_amd.pretransforms import AutoencoderPretransform
    audio, noise = _ae_inputs(device)
    full = AutoencoderPretransform(build_native_ae("tiny", 100, device), scale=0.7)
    half = AutoencoderPretransform(build_native_ae("tiny", 100, device), scale=0.7, model_half=True)
    assert half.model_half and all(v.dtype == torch.float16 for v in half.model.state_dict().values() if v.is_floating_point())
    hand = build_native_ae("tiny", 100, device)
    with torch.no_grad():
        for p in hand.parameters():
            p.copy_(p.half().float())
    hand = AutoencoderPretransform(hand, scale=0.7)
    with torch.no_grad():
        z_half = half.encode(audio, noise=noise)
        z_hand = hand.encode(audio.half().float(), noise=noise)
        z_full = full.encode(audio, noise=noise)
        d_half = half.decode(z_full)
        d_hand = hand.decode(((z_full * 0.7).half().float()) / 0.7)
        d_full = full.decode(z_full)
    assert z_half.dtype == torch.float32 and d_half.dtype == torch.float32
    assert rel_err(z_half, z_hand) < 1e-3 and rel_err(d_half, d_hand) < 1e-3          # fp16 output rounding: 2^-11
    e_z, e_d = rel_err(z_half, z_full), rel_err(d_half, d_full)
    assert 0 < e_z < 2e-2 and 0 < e_d < 2e-2, (e_z, e_d)
    import refimport
    if not refimport.available():
        return
    import contextlib
    import sys
    with contextlib.redirect_stdout(sys.stderr):
        refimport.import_reference()
    from stable_audio_tools.models.autoencoders import create_autoencoder_from_config
    from stable_audio_tools.models.pretransforms import AutoencoderPretransform as RefPretransform
    cfg = seeded.AE_CONFIGS["tiny"]
    ref = create_autoencoder_from_config(copy.deepcopy(cfg))
    ref.load_state_dict({k: v.float().cpu() for k, v in full.model.state_dict().items()})
    ref_half = RefPretransform(ref, scale=0.7, model_half=True)
    try:
        with torch.no_grad():
            d_ref = ref_half.decode(z_full.cpu())
    except RuntimeError as e:        # a CPU build without half convolutions
        pytest.skip(f"reference half path not runnable on this host: {e}")
    assert d_ref.dtype == torch.float32
    e_ref = rel_err(d_ref, d_full)
    assert rel_err(d_half, d_ref) < max(2e-2, 3 * e_ref)        # both are fp16-level approximations of the same fp32 decode
    assert e_d <= 1.5 * e_ref + 1e-3, (e_d, e_ref)              # ... and the native one is not the worse of the two


def test_pretransform_model_half_simulator(emu_modules):
    _pretransform_model_half("cpu")


@pytest.mark.gpu
def test_pretransform_model_half_gpu(hip):
    _pretransform_model_half("cuda")


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
        # a torch optimizer step invalidates what was derived from ITS parameters (global post-step hook), and only that: an
        # optimizer of some other model (the DiT being trained next to this frozen pretransform) leaves the pretransform's copies alone
        cnt.reset()
        p = torch.nn.Parameter(torch.zeros(3, device=device))
        p.grad = torch.ones(3, device=device)
        e_p = _caches.epoch_of(p)
        torch.optim.SGD([p], lr=0.1).step()
        assert _caches.epoch_of(p) != e_p
        assert torch.equal(pt.decode(z), d3) and cnt.total() == 0, cnt.n
        conv.weight_g.grad = torch.zeros_like(conv.weight_g)
        torch.optim.SGD([conv.weight_g], lr=0.1).step()               # ... while an optimizer that owns one of ITS parameters does
        conv.weight_g.grad = None
        pt.decode(z)
        assert cnt.n["wn_fold"] == 1, cnt.n
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
