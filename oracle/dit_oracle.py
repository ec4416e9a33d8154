"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product package).

CPU restatement (plain torch functional ops on a reference-format state_dict) of the DiT path:
stable_audio_tools/models/dit.py (DiffusionTransformer._forward :125-229, forward :231-431) and
stable_audio_tools/models/transformer.py (ContinuousTransformer.forward :796-865, TransformerBlock.forward
:659-713, Attention.forward :445-543, apply_rotary_pos_emb :155-174, LayerNorm :236-241, GLU :263-275),
models/blocks.py:85-94 (FourierFeatures).  Pinned against golden vectors generated from the reference
(oracle/gen_golden.py -> tests/golden/dit_*.npz).  dtype-generic (float32 / float64).
"""
import math

import torch
import torch.nn.functional as F


def _lin(sd, prefix, x):
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


def _mlp(sd, prefix, x):
    """nn.Sequential(Linear, SiLU, Linear) — dit.py:49-53, :62-66, :73-77, :82-86; transformer.py:769-773."""
    return _lin(sd, prefix + ".2", F.silu(_lin(sd, prefix + ".0", x)))


def fourier_features(sd, prefix, t):
    f = 2 * math.pi * t @ sd[prefix + ".weight"].T
    return torch.cat([f.cos(), f.sin()], dim=-1)


def layer_norm(sd, prefix, x, eps=1e-5):
    return F.layer_norm(x, x.shape[-1:], weight=sd[prefix + ".gamma"], bias=sd[prefix + ".beta"], eps=eps)


def rotary_freqs(inv_freq, n):
    """RotaryEmbedding.forward (transformer.py:125-138): freqs = cat(t*inv_freq, t*inv_freq), fp32 in the reference."""
    t = torch.arange(n, dtype=inv_freq.dtype)
    f = torch.einsum("i,j->ij", t, inv_freq)
    return torch.cat((f, f), dim=-1)


def apply_rotary(t, freqs):
    """apply_rotary_pos_emb (transformer.py:155-174): first rot_dim dims, NeoX halves, freqs[-seq_len:]."""
    rot_dim, n = freqs.shape[-1], t.shape[-2]
    freqs = freqs[-n:, :]
    tr, tu = t[..., :rot_dim], t[..., rot_dim:]
    x1, x2 = tr[..., : rot_dim // 2], tr[..., rot_dim // 2:]
    rot = torch.cat((-x2, x1), dim=-1)
    return torch.cat((tr * freqs.cos() + rot * freqs.sin(), tu), dim=-1)


def attention(sd, prefix, x, dim_heads, context=None, freqs=None):
    """Attention.forward (transformer.py:445-543) + apply_attn (:406-441): softmax(q k^T / sqrt(d)) v, dense,
    unmasked; kv heads repeated (repeat_interleave) when the context is narrower than the model."""
    b, n, d = x.shape
    h = d // dim_heads
    if (prefix + ".to_q.weight") in sd:
        kv_in = context if context is not None else x
        q = _lin(sd, prefix + ".to_q", x)
        k, v = _lin(sd, prefix + ".to_kv", kv_in).chunk(2, dim=-1)
    else:
        q, k, v = _lin(sd, prefix + ".to_qkv", x).chunk(3, dim=-1)
    kv_h = k.shape[-1] // dim_heads
    q = q.view(b, n, h, dim_heads).transpose(1, 2)
    k = k.reshape(b, -1, kv_h, dim_heads).transpose(1, 2)
    v = v.reshape(b, -1, kv_h, dim_heads).transpose(1, 2)
    if freqs is not None:
        q, k = apply_rotary(q, freqs), apply_rotary(k, freqs)
    if kv_h != h:
        k = k.repeat_interleave(h // kv_h, dim=1)
        v = v.repeat_interleave(h // kv_h, dim=1)
    att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dim_heads), dim=-1)
    out = (att @ v).transpose(1, 2).reshape(b, n, d)
    return _lin(sd, prefix + ".to_out", out)


def feed_forward(sd, prefix, x):
    """FeedForward/GLU (transformer.py:263-275, :290-326): Linear(d, 2*4d) -> x*silu(gate) -> Linear(4d, d)."""
    a, g = _lin(sd, prefix + ".ff.0.proj", x).chunk(2, dim=-1)
    return _lin(sd, prefix + ".ff.2", a * F.silu(g))


def transformer_block(sd, prefix, x, dim_heads, context, global_cond, freqs):
    """TransformerBlock.forward (transformer.py:659-713)."""
    has_cross = (prefix + ".cross_attn.to_q.weight") in sd
    if (prefix + ".to_scale_shift_gate") in sd and global_cond is not None:
        sc_s, sh_s, g_s, sc_f, sh_f, g_f = (sd[prefix + ".to_scale_shift_gate"] + global_cond).unsqueeze(1).chunk(6, dim=-1)
        res = x
        h = layer_norm(sd, prefix + ".pre_norm", x) * (1 + sc_s) + sh_s
        h = attention(sd, prefix + ".self_attn", h, dim_heads, freqs=freqs)
        x = h * torch.sigmoid(1 - g_s) + res
        if context is not None and has_cross:
            x = x + attention(sd, prefix + ".cross_attn", layer_norm(sd, prefix + ".cross_attend_norm", x), dim_heads, context=context)
        res = x
        h = layer_norm(sd, prefix + ".ff_norm", x) * (1 + sc_f) + sh_f
        h = feed_forward(sd, prefix + ".ff", h)
        x = h * torch.sigmoid(1 - g_f) + res
    else:
        x = x + attention(sd, prefix + ".self_attn", layer_norm(sd, prefix + ".pre_norm", x), dim_heads, freqs=freqs)
        if context is not None and has_cross:
            x = x + attention(sd, prefix + ".cross_attn", layer_norm(sd, prefix + ".cross_attend_norm", x), dim_heads, context=context)
        x = x + feed_forward(sd, prefix + ".ff", layer_norm(sd, prefix + ".ff_norm", x))
    return x


def continuous_transformer(sd, prefix, x, depth, dim_heads, prepend_embeds=None, context=None, global_cond=None):
    """ContinuousTransformer.forward (transformer.py:796-865), rotary_pos_emb=True, no memory tokens."""
    x = _lin(sd, prefix + ".project_in", x)
    if prepend_embeds is not None:
        x = torch.cat((prepend_embeds, x), dim=-2)
    freqs = rotary_freqs(sd[prefix + ".rotary_pos_emb.inv_freq"].to(x.dtype), x.shape[1])
    if global_cond is not None and (prefix + ".global_cond_embedder.0.weight") in sd:
        global_cond = _mlp(sd, prefix + ".global_cond_embedder", global_cond)
    for i in range(depth):
        x = transformer_block(sd, f"{prefix}.layers.{i}", x, dim_heads, context, global_cond, freqs)
    return _lin(sd, prefix + ".project_out", x)


def dit_inner_forward(sd, cfg, x, t, cross_attn_cond=None, global_embed=None, prepend_cond=None):
    """DiffusionTransformer._forward (dit.py:125-229), patch_size 1, no input_concat."""
    embed_dim, depth = cfg["embed_dim"], cfg["depth"]
    dim_heads = embed_dim // cfg["num_heads"]
    gtype = cfg.get("global_cond_type", "prepend")
    if cross_attn_cond is not None:
        cross_attn_cond = _mlp(sd, "to_cond_embed", cross_attn_cond)
    if global_embed is not None:
        global_embed = _mlp(sd, "to_global_embed", global_embed)
    prepend_inputs, prepend_length = None, 0
    if prepend_cond is not None:
        prepend_inputs = _mlp(sd, "to_prepend_embed", prepend_cond)
        prepend_length = prepend_inputs.shape[1]
    timestep_embed = _mlp(sd, "to_timestep_embed", fourier_features(sd, "timestep_features", t[:, None]))
    global_embed = global_embed + timestep_embed if global_embed is not None else timestep_embed
    if gtype == "prepend":
        g = global_embed.unsqueeze(1)
        prepend_inputs = g if prepend_inputs is None else torch.cat([prepend_inputs, g], dim=1)
        prepend_length = prepend_inputs.shape[1]
    x = F.conv1d(x, sd["preprocess_conv.weight"]) + x
    x = x.transpose(1, 2)
    out = continuous_transformer(sd, "transformer", x, depth, dim_heads, prepend_embeds=prepend_inputs, context=cross_attn_cond,
                                 global_cond=global_embed if gtype == "adaLN" else None)
    out = out.transpose(1, 2)[:, :, prepend_length:]
    return F.conv1d(out, sd["postprocess_conv.weight"]) + out


def dit_forward(sd, cfg, x, t, cross_attn_cond=None, global_embed=None, prepend_cond=None, cfg_scale=1.0, scale_phi=0.0,
                negative_cross_attn_cond=None):
    """DiffusionTransformer.forward (dit.py:231-431) for cfg_dropout_prob = 0, cfg_interval = (0, 1)."""
    if cfg_scale == 1.0 or (cross_attn_cond is None and prepend_cond is None):
        return dit_inner_forward(sd, cfg, x, t, cross_attn_cond, global_embed, prepend_cond)
    null = torch.zeros_like(cross_attn_cond) if cross_attn_cond is not None else None
    neg = negative_cross_attn_cond if negative_cross_attn_cond is not None else null
    bc = torch.cat([cross_attn_cond, neg], dim=0) if cross_attn_cond is not None else None
    bp = torch.cat([prepend_cond, torch.zeros_like(prepend_cond)], dim=0) if prepend_cond is not None else None
    bg = torch.cat([global_embed, global_embed], dim=0) if global_embed is not None else None
    out = dit_inner_forward(sd, cfg, torch.cat([x, x], dim=0), torch.cat([t, t], dim=0), bc, bg, bp)
    cond, uncond = out.chunk(2, dim=0)
    cfg_out = uncond + (cond - uncond) * cfg_scale
    if scale_phi != 0.0:
        cfg_out = scale_phi * (cfg_out * (cond.std(dim=1, keepdim=True) / cfg_out.std(dim=1, keepdim=True))) + (1 - scale_phi) * cfg_out
    return cfg_out
