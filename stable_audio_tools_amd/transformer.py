"""MI355X-native ContinuousTransformer — drop-in mirror of
stable_audio_tools/models/transformer.py (LayerNorm :215, RotaryEmbedding :92, GLU :252,
FeedForward :277, Attention :328, TransformerBlock :582, ContinuousTransformer :715): same class
names, constructor kwargs and state_dict keys (gamma/beta, to_qkv / to_q / to_kv / to_out,
ff.ff.0.proj / ff.ff.2, to_scale_shift_gate, project_in/out, rotary_pos_emb.inv_freq,
global_cond_embedder).

Execution: LayerNorm(+adaLN modulate), partial rotary, attention (self and GQA cross), SwiGLU and the
gate/residual epilogue are HIP kernels (csrc/dit_ops.hip, csrc/attention.hip); every projection is
linear.Linear on the native MFMA GEMM (csrc/gemm.hip) with the head split / rotary / SwiGLU / gate /
residual work of the block fused into the GEMM epilogues — no library GEMM.  Options of the reference that
the Stable Audio DiT configs never enable (qk_norm, differential attention, conformer, layer_scale,
memory tokens, sliding window, causal, flex-attention masks, abs/sinusoidal position embeddings)
raise NotImplementedError.

Forward AND backward run on the HIP kernels: every fused operator is a torch.autograd.Function whose
backward calls the matching kernel (LayerNorm+adaLN, rotary transpose, attention dQ / dK,dV with
probabilities recomputed from the saved log-sum-exp, SwiGLU, gate/residual); the projections' data and
weight gradients run on the same native GEMM (linear.LinearFn.backward).  Per-layer activation checkpointing
of the reference (transformer.py:28-30, :840-845) is optional here: off by default (288 GB of HBM), on with
`use_checkpointing=True` / `ContinuousTransformer.checkpointing = True` when the batch does not fit.
"""

import torch
from torch import nn

from . import _caches
from . import functional as _fn
from . import linear as _linear
from .linear import Linear, _lowp

# LayerNorm -> fp8 rows for an fp8 consumer (round 4); False (set by an A/B script): normalise to bf16, then quantise
fp8_ln_fused = True


def _ops():
    return _fn._ops(None)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, scale, shift, eps, g32=None, b32=None):
        """g32 / b32: optional fp32 copies of gamma / beta prepared by the caller (the kernels read fp32 parameters)."""
        ops = _ops()
        x = x.contiguous()
        if g32 is None:
            g32 = gamma.float().contiguous()
        if b32 is None and beta is not None:
            b32 = beta.float().contiguous()
        if any(ctx.needs_input_grad):   # (torch.is_grad_enabled() is always False inside Function.forward)
            y, mean, rstd = ops.layernorm(x, g32, b32, scale, shift, eps, save_stats=True)
            ctx.save_for_backward(x, g32, b32, scale, mean, rstd)
            ctx.ops = ops
            ctx.gdtype = gamma.dtype
        else:
            y = ops.layernorm(x, g32, b32, scale, shift, eps)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, g32, b32, scale, mean, rstd = ctx.saved_tensors
        dx, dgamma, dscale, dshift = ctx.ops.layernorm_bwd(dy.contiguous(), x, g32, b32, scale, mean, rstd)
        if dscale is not None:
            dscale, dshift = dscale.to(scale.dtype), dshift.to(scale.dtype)
        return dx, dgamma.to(ctx.gdtype), None, dscale, dshift, None, None, None


class LayerNormResFn(torch.autograd.Function):
    """(LayerNorm(x), x) for a pre-norm residual branch x = x + f(LN(x)) (transformer.py:703-712): the second output is x itself, to be
    handed to the branch's output projection as `res`.  Autograd then delivers BOTH gradients of x to this node — through the normalised
    branch and along the residual path — and the LayerNorm backward kernel adds the latter in its own pass (sat_layernorm_bwd_res) instead
    of autograd launching an elementwise add of two (B, N, D) tensors per norm (round 6)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, g32, b32):
        ops = _ops()
        x = x.contiguous()
        y, mean, rstd = ops.layernorm(x, g32, b32, None, None, eps, save_stats=True)
        ctx.save_for_backward(x, g32, b32, mean, rstd)
        ctx.ops, ctx.gdtype = ops, gamma.dtype
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, g32, b32, mean, rstd = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None, None, None
        if dres is not None and dres.dtype != dy.dtype:
            dres = dres.to(dy.dtype)
        dx, dgamma, _, _ = ctx.ops.layernorm_bwd(dy.contiguous(), x, g32, b32, None, mean, rstd,
                                                 dres=dres.contiguous() if dres is not None else None)
        return dx, dgamma.to(ctx.gdtype), None, None, None, None


class LayerNorm(nn.Module):
    def __init__(self, dim, bias=False, fix_scale=False, force_fp32=False, eps=1e-5):
        super().__init__()
        if fix_scale:
            self.register_buffer("gamma", torch.ones(dim))
        else:
            self.gamma = nn.Parameter(torch.ones(dim))
        if bias:
            # the backward kernel does not produce d(beta); no Stable Audio config sets norm_kwargs.bias
            raise NotImplementedError("LayerNorm(bias=True) is not on the HIP path (the reference default is a zero buffer)")
        else:
            self.register_buffer("beta", torch.zeros(dim))
        self.eps = eps
        self.force_fp32 = force_fp32   # statistics are always fp32 in the HIP kernel
        self._f32_cache = {}

    def _as_f32(self, name):
        """fp32 view of a parameter for the kernels; for a bf16/fp16 module the converted copy is cached until the
        parameter changes (in-place version counter / storage)."""
        t = getattr(self, name)
        if t.dtype == torch.float32:
            return t.detach()
        if not _caches.trackable(t):
            return t.detach().float().contiguous()
        key = (t.data_ptr(), _caches.version_of(t), t.device, _caches.epoch_of(t))
        hit = self._f32_cache.get(name)
        if hit is None or hit[0] != key:
            hit = (key, t.detach().float().contiguous())
            self._f32_cache[name] = hit
        return hit[1]

    def forward(self, x, scale=None, shift=None, fp8_for=None):
        """scale/shift: optional (B, D) adaLN modulation fused into the same pass: LN(x)*(1+scale)+shift.
        fp8_for: the ONE projection that consumes the result.  In inference, when that Linear runs in fp8 with per-row activation scales, the
        normalised rows leave the kernel as fp8 + row scales (linear.Fp8Rows): no bf16 LayerNorm output, no quantisation pass."""
        if (fp8_for is not None and fp8_for.fp8 and _linear.fp8_row_scales and not torch.is_grad_enabled() and x.dim() == 3
                and x.dtype == torch.bfloat16 and fp8_for.in_features % 16 == 0 and fp8_for.out_features % 8 == 0
                and fp8_for.fp8_weight() is not None and fp8_ln_fused):
            out = _ops().layernorm_fp8(x.contiguous(), self._as_f32("gamma"), self._as_f32("beta"), scale, shift, self.eps)
            if out is not None:
                return _linear.Fp8Rows(out[0], out[1], x.shape)
        return LayerNormFn.apply(x, self.gamma, self.beta, scale, shift, self.eps, self._as_f32("gamma"), self._as_f32("beta"))

    def with_residual(self, x):
        """(LN(x), x) — see LayerNormResFn; training path of the un-modulated pre-norm branches."""
        return LayerNormResFn.apply(x, self.gamma, self.beta, self.eps, self._as_f32("gamma"), self._as_f32("beta"))


class RotaryEmbedding(nn.Module):
    def __init__(self, dim, use_xpos=False, scale_base=512, interpolation_factor=1., base=10000, base_rescale_factor=1.):
        super().__init__()
        if use_xpos:
            raise NotImplementedError("use_xpos is dead code in the reference (transformer.py:143) and not implemented")
        base *= base_rescale_factor ** (dim / (dim - 2))
        inv_freq = 1. / (base ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq)
        assert interpolation_factor >= 1.
        self.interpolation_factor = interpolation_factor
        self.register_buffer("scale", None)

    def tables(self, seq_len):
        """(seq_len, dim/2, 2) fp32 cos/sin table (csrc/dit_ops.hip sat_rope_tables)."""
        inv = self.inv_freq.float().contiguous()
        if self.interpolation_factor != 1.:
            inv = inv / self.interpolation_factor
        return _ops().rope_tables(inv, seq_len)

    def forward_from_seq_len(self, seq_len):
        return self.tables(seq_len), 1.


class _RopeQKFn(torch.autograd.Function):
    """In-place rotary on the q and k slices of a fused (B, N, 3*H*dh) projection."""

    @staticmethod
    def forward(ctx, qkv, cs, heads, dh):
        ops = _ops()
        b, n, _ = qkv.shape
        hd = heads * dh
        ops.rope_apply_(qkv[..., 0:hd].unflatten(-1, (heads, dh)), cs)
        ops.rope_apply_(qkv[..., hd:2 * hd].unflatten(-1, (heads, dh)), cs)
        ctx.mark_dirty(qkv)
        ctx.save_for_backward(cs)
        ctx.meta = (heads, dh, ops)
        return qkv

    @staticmethod
    def backward(ctx, g):
        (cs,) = ctx.saved_tensors
        heads, dh, ops = ctx.meta
        g = g.clone()
        hd = heads * dh
        ops.rope_apply_(g[..., 0:hd].unflatten(-1, (heads, dh)), cs, transpose=True)
        ops.rope_apply_(g[..., hd:2 * hd].unflatten(-1, (heads, dh)), cs, transpose=True)
        return g, None, None, None


class _AttentionCoreFn(torch.autograd.Function):
    """o = softmax(q k^T * scale) v  on csrc/attention.hip; backward = sat_attention_bwd (dQ and dK/dV kernels
    that recompute the probabilities from the saved log-sum-exp — nothing N x N is ever stored)."""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        ops = _ops()
        if any(ctx.needs_input_grad[:3]):
            o, lse, planes = ops.attention(q, k, v, scale, return_planes=True)
            ctx.ops, ctx.scale, ctx.planes = ops, scale, planes
            ctx.shapes = (q.shape, k.shape)
            ctx.save_for_backward(o, lse)
            return o
        return ops.attention(q, k, v, scale)

    @staticmethod
    def backward(ctx, g):
        o, lse = ctx.saved_tensors
        (_, _, _, _), (_, hkv, nk, _) = ctx.shapes
        dq, dk, dv = ctx.ops.attention_bwd(ctx.planes, o, g.contiguous(), lse, ctx.scale, hkv, nk)
        return dq, dk, dv, None


class _SelfAttnFn(torch.autograd.Function):
    """The training path of a fused-projection self-attention as ONE autograd node (round 6): in-place rotary on the q / k thirds of
    the (B, N, 3*H*dh) projection, head split, attention core (transformer.py:155-174, :389-441 of the reference).  As separate nodes
    (_RopeQKFn, three slices, three permutes, _AttentionCoreFn) autograd rebuilt the projection's gradient from dq / dk / dv with three
    zero-filled (B, N, 3*H*dh) tensors, three strided copies into their thirds, two adds and a clone for the in-place rotary — nine
    activation-sized launches per layer; here dq / dk / dv leave the backward kernels as one (3, B, H, N, dh) buffer that ONE strided
    copy turns into the projection's layout, and the inverse rotary runs in place on that copy.
    `qkv` is rotated in place: it is the output of the projection GEMM, which nothing else reads (autograd raises if something saved it)."""

    @staticmethod
    def forward(ctx, qkv, cs, heads, dh, scale):
        ops = _ops()
        b, n, _ = qkv.shape
        if not qkv.is_contiguous():
            qkv = qkv.contiguous()
        if cs is not None:
            ops.rope_apply_(qkv[..., 0:2 * heads * dh].unflatten(-1, (2 * heads, dh)), cs)     # q and k heads in one launch
        q5 = qkv.view(b, n, 3, heads, dh)
        q, k, v = (q5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        if ctx.needs_input_grad[0]:
            o, lse, planes = ops.attention(q, k, v, scale, return_planes=True)
            ctx.ops, ctx.scale, ctx.planes, ctx.meta = ops, scale, planes, (b, n, heads, dh)
            ctx.save_for_backward(o, lse, cs)
            return o
        return ops.attention(q, k, v, scale)

    @staticmethod
    def backward(ctx, g):
        o, lse, cs = ctx.saved_tensors
        b, n, heads, dh = ctx.meta
        buf = torch.empty(3, b, heads, n, dh, dtype=o.dtype, device=o.device)
        ctx.ops.attention_bwd(ctx.planes, o, g.contiguous(), lse, ctx.scale, heads, n, out=(buf[0], buf[1], buf[2]))
        dqkv = torch.empty(b, n, 3, heads, dh, dtype=o.dtype, device=o.device)
        dqkv.copy_(buf.permute(1, 3, 0, 2, 4))
        dqkv = dqkv.view(b, n, 3 * heads * dh)
        if cs is not None:
            ctx.ops.rope_apply_(dqkv[..., 0:2 * heads * dh].unflatten(-1, (2 * heads, dh)), cs, transpose=True)
        return dqkv, None, None, None, None


class _CrossAttnFn(torch.autograd.Function):
    """The training path of the cross-attention core as one autograd node: q (B, N, H*dh) and the fused key / value projection
    kv (B, M, 2*Hkv*dh) in, merged heads out; the backward assembles dkv with one strided copy (see _SelfAttnFn)."""

    @staticmethod
    def forward(ctx, q2, kv, heads, kv_heads, dh, scale):
        ops = _ops()
        b, n, _ = q2.shape
        m = kv.shape[1]
        q = q2.view(b, n, heads, dh).permute(0, 2, 1, 3)
        kv5 = kv.contiguous().view(b, m, 2, kv_heads, dh)
        k, v = kv5[:, :, 0].permute(0, 2, 1, 3), kv5[:, :, 1].permute(0, 2, 1, 3)
        if any(ctx.needs_input_grad[:2]):
            o, lse, planes = ops.attention(q, k, v, scale, return_planes=True)
            ctx.ops, ctx.scale, ctx.planes, ctx.meta = ops, scale, planes, (b, n, m, heads, kv_heads, dh)
            ctx.save_for_backward(o, lse)
            return o
        return ops.attention(q, k, v, scale)

    @staticmethod
    def backward(ctx, g):
        o, lse = ctx.saved_tensors
        b, n, m, heads, kv_heads, dh = ctx.meta
        buf = torch.empty(2, b, kv_heads, m, dh, dtype=o.dtype, device=o.device)
        dq = torch.empty(b, heads, n, dh, dtype=o.dtype, device=o.device)
        ctx.ops.attention_bwd(ctx.planes, o, g.contiguous(), lse, ctx.scale, kv_heads, m, out=(dq, buf[0], buf[1]))
        dq2 = torch.empty(b, n, heads, dh, dtype=o.dtype, device=o.device)
        dq2.copy_(dq.permute(0, 2, 1, 3))
        dkv = torch.empty(b, m, 2, kv_heads, dh, dtype=o.dtype, device=o.device)
        dkv.copy_(buf.permute(1, 3, 0, 2, 4))
        return dq2.view(b, n, heads * dh), dkv.view(b, m, 2 * kv_heads * dh), None, None, None, None


class _SwiGLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xin):
        xin = xin.contiguous()
        ctx.save_for_backward(xin)
        ctx.ops = _ops()
        return ctx.ops.swiglu(xin)

    @staticmethod
    def backward(ctx, g):
        (xin,) = ctx.saved_tensors
        return ctx.ops.swiglu_bwd(xin, g.contiguous())


class _GateResidualFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gate, res):
        ops = _ops()
        x = x.contiguous()
        ctx.ops = ops
        ctx.save_for_backward(x, gate)
        return ops.gate_residual(x, gate, res.contiguous())

    @staticmethod
    def backward(ctx, g):
        x, gate = ctx.saved_tensors
        g = g.contiguous()
        dx, dgate = ctx.ops.gate_residual_bwd(g, x, gate)
        return dx, dgate.to(gate.dtype), g


class GLU(nn.Module):
    def __init__(self, dim_in, dim_out, activation=None, use_conv=False, conv_kernel_size=3):
        super().__init__()
        if use_conv:
            raise NotImplementedError("GLU(use_conv=True) is not used by the DiT and not implemented")
        self.proj = Linear(dim_in, dim_out * 2)

    def forward(self, x):
        return self.proj(x, mode="swiglu")       # x * silu(gate) in the GEMM epilogue (transformer.py:274-275)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, no_bias=False, glu=True, use_conv=False, conv_kernel_size=3,
                 zero_init_output=True):
        super().__init__()
        if not glu or use_conv:
            raise NotImplementedError("only the SwiGLU feed-forward (glu=True, use_conv=False) is implemented")
        inner_dim = int(dim * mult)
        dim_out = dim if dim_out is None else dim_out
        linear_in = GLU(dim, inner_dim)
        linear_out = Linear(inner_dim, dim_out, bias=not no_bias)
        if zero_init_output:
            nn.init.zeros_(linear_out.weight)
            if not no_bias:
                nn.init.zeros_(linear_out.bias)
        self.ff = nn.Sequential(linear_in, nn.Identity(), linear_out, nn.Identity())

    def forward(self, x, res=None):
        """res: the residual stream — added in the output projection's epilogue (x = x + ff(...), transformer.py:711)."""
        return self.ff[2](self.ff[0](x), res=res)


class Attention(nn.Module):
    def __init__(self, dim, dim_heads=64, dim_context=None, causal=False, zero_init_output=True, qk_norm="none",
                 differential=False, feat_scale=False):
        super().__init__()
        if causal or qk_norm != "none" or differential or feat_scale:
            raise NotImplementedError("causal / qk_norm / differential / feat_scale attention are not on the HIP path")
        if dim_heads != 64:
            raise NotImplementedError("the HIP attention kernel is specialised for head dim 64 (every Stable Audio DiT config)")
        self.dim, self.dim_heads = dim, dim_heads
        dim_kv = dim_context if dim_context is not None else dim
        self.num_heads = dim // dim_heads
        self.kv_heads = dim_kv // dim_heads
        if dim_context is not None:
            self.to_q = Linear(dim, dim, bias=False)
            self.to_kv = Linear(dim_kv, dim_kv * 2, bias=False)
        else:
            self.to_qkv = Linear(dim, dim * 3, bias=False)
        self.to_out = Linear(dim, dim, bias=False)
        if zero_init_output:
            nn.init.zeros_(self.to_out.weight)
        self.causal = False
        self.scale = dim_heads ** -0.5

    @staticmethod
    def _heads(ops, lin, x2, cs, heads, nb, ntok, sec0, nsec, tag):
        """Input projection straight into attention operand planes (bf16, or fp8 operands when the layer is switched to fp8)."""
        if isinstance(x2, _linear.Fp8Rows):
            qw, sw = lin.fp8_weight()
            alpha, calpha = _linear.fp8_scales(None, sw)
            return ops.gemm_heads_fp8(x2.q, qw, alpha, cs, heads, nb, ntok, sec0, nsec, reuse=tag, row_alpha=x2.scale, col_alpha=calpha)
        if lin.fp8 and lin.fp8_weight() is not None:
            qw, sw = lin.fp8_weight()
            if _linear.fp8_row_scales and x2.shape[1] % 8 == 0 and x2.stride(0) % 8 == 0 and x2.shape[1] <= 8192:
                qx, rs = ops.quant_fp8_rows(x2)
                alpha, calpha = _linear.fp8_scales(None, sw)
                return ops.gemm_heads_fp8(qx, qw, alpha, cs, heads, nb, ntok, sec0, nsec, reuse=tag, row_alpha=rs, col_alpha=calpha)
            qx, sx = ops.quant_fp8(x2)
            alpha, calpha = _linear.fp8_scales(sx, sw)
            return ops.gemm_heads_fp8(qx, qw, alpha, cs, heads, nb, ntok, sec0, nsec, reuse=tag, col_alpha=calpha)
        return ops.gemm_heads_bf16(x2, lin.lowp_weight(), cs, heads, nb, ntok, sec0, nsec, reuse=tag)

    def forward(self, x, context=None, rotary_pos_emb=None, causal=None, res=None, **unsupported):
        """res: the residual stream, added in the output projection's epilogue (x = x + attn(...), transformer.py:703-707)."""
        for k, v in unsupported.items():
            if v is not None:
                raise NotImplementedError(f"Attention.forward({k}=...) is not on the HIP path")
        h, kv_h, dh = self.num_heads, self.kv_heads, self.dim_heads
        b, n, _ = x.shape
        cross = hasattr(self, "to_q")
        if cross and rotary_pos_emb is not None:
            raise NotImplementedError("rotary on a separate-projection attention is never used by the DiT")
        kv_input = context if (cross and context is not None) else x
        first = self.to_q if cross else self.to_qkv
        if not torch.is_grad_enabled() and _lowp(x, first.weight) and first.lowp_weight() is not None \
                and (not cross or self.to_kv.lowp_weight() is not None):
            # inference, bf16: head split, rotary and the attention kernel's operand planes come straight out of the
            # projection GEMM's epilogue (no qkv tensor, no rotary pass, no plane-preparation pass)
            ops = _ops()
            if isinstance(x, _linear.Fp8Rows):
                x2 = x
            else:
                x2 = x.reshape(b * n, -1)
                x2 = x2 if x2.dtype == torch.bfloat16 else ops.cast_bf16(x2.contiguous())
            if cross:
                m = kv_input.shape[1]
                pq = self._heads(ops, self.to_q, x2, None, h, b, n, 0, 1, "cross")
                # the conditioning is the same tensor at every sampler step (dit.py caches its embedding): its K / V planes are
                # computed once per (context object, version, weight version) and kept in this layer's own buffers
                w = self.to_kv.weight
                can_cache = _caches.trackable(kv_input, w)    # an inference tensor carries no version counter: project it every call
                key = (_caches.version_of(kv_input), _caches.version_of(w), w.data_ptr(), _caches.epoch_of(w), bool(self.to_kv.fp8),
                       tuple(kv_input.shape), kv_input.dtype)
                hit = can_cache and getattr(self, "_kv_ctx", None) is kv_input and self._kv_key == key
                if hit:
                    pkv = self._kv_planes
                else:
                    c2 = kv_input.reshape(b * m, -1)
                    c2 = c2 if c2.dtype == torch.bfloat16 else ops.cast_bf16(c2.contiguous())
                    pkv = self._heads(ops, self.to_kv, c2, None, kv_h, b, m, 1, 2, ("crosskv", id(self)))
                    if can_cache:
                        self._kv_ctx, self._kv_key, self._kv_planes = kv_input, key, pkv
                    else:
                        self.__dict__.pop("_kv_ctx", None)
                out = ops.attention_planes(pq["q"], pkv["k"], pkv["v_tr"], n, m, self.scale)
            else:
                cs = rotary_pos_emb[0] if rotary_pos_emb is not None else None
                pl = self._heads(ops, self.to_qkv, x2, cs, h, b, n, 0, 3, "self")
                out = ops.attention_planes(pl["q"], pl["k"], pl["v_tr"], n, n, self.scale)
            return self.to_out(out, res=res)
        if _ops().train_fused_nodes and dh == 64:
            if cross:
                out = _CrossAttnFn.apply(self.to_q(x), self.to_kv(kv_input), h, kv_h, dh, self.scale)
            else:
                cs = rotary_pos_emb[0] if rotary_pos_emb is not None else None
                out = _SelfAttnFn.apply(self.to_qkv(x), cs, h, dh, self.scale)     # (B, N, H*dh): heads already merged
            return self.to_out(out, res=res)
        if cross:
            q = self.to_q(x).view(b, n, h, dh).permute(0, 2, 1, 3)
            kv = self.to_kv(kv_input)
            k = kv[..., :kv_h * dh].unflatten(-1, (kv_h, dh)).permute(0, 2, 1, 3)
            v = kv[..., kv_h * dh:].unflatten(-1, (kv_h, dh)).permute(0, 2, 1, 3)
        else:
            qkv = self.to_qkv(x)
            if rotary_pos_emb is not None:
                cs, _ = rotary_pos_emb
                qkv = _RopeQKFn.apply(qkv, cs, h, dh)
            hd = h * dh
            q = qkv[..., 0:hd].unflatten(-1, (h, dh)).permute(0, 2, 1, 3)
            k = qkv[..., hd:2 * hd].unflatten(-1, (h, dh)).permute(0, 2, 1, 3)
            v = qkv[..., 2 * hd:].unflatten(-1, (h, dh)).permute(0, 2, 1, 3)
        out = _AttentionCoreFn.apply(q, k, v, self.scale)       # (B, N, H*dh): heads already merged
        return self.to_out(out, res=res)


class TransformerBlock(nn.Module):
    def __init__(self, dim, dim_heads=64, cross_attend=False, dim_context=None, global_cond_dim=None, causal=False,
                 zero_init_branch_outputs=True, conformer=False, layer_ix=-1, remove_norms=False, add_rope=False,
                 layer_scale=False, attn_kwargs={}, ff_kwargs={}, norm_kwargs={}):
        super().__init__()
        if conformer or remove_norms or add_rope or layer_scale or causal:
            raise NotImplementedError("conformer / remove_norms / add_rope / layer_scale / causal are not on the HIP path")
        self.dim = dim
        self.dim_heads = min(dim_heads, dim)
        self.cross_attend = cross_attend
        self.dim_context = dim_context
        self.pre_norm = LayerNorm(dim, **norm_kwargs)
        self.self_attn = Attention(dim, dim_heads=self.dim_heads, zero_init_output=zero_init_branch_outputs, **attn_kwargs)
        if cross_attend:
            self.cross_attend_norm = LayerNorm(dim, **norm_kwargs)
            self.cross_attn = Attention(dim, dim_heads=self.dim_heads, dim_context=dim_context,
                                        zero_init_output=zero_init_branch_outputs, **attn_kwargs)
        self.ff_norm = LayerNorm(dim, **norm_kwargs)
        self.ff = FeedForward(dim, zero_init_output=zero_init_branch_outputs, **ff_kwargs)
        self.layer_ix = layer_ix
        self.global_cond_dim = global_cond_dim
        if global_cond_dim is not None:
            self.to_scale_shift_gate = nn.Parameter(torch.randn(6 * dim) / dim ** 0.5)

    def forward(self, x, context=None, global_cond=None, rotary_pos_emb=None, **unsupported):
        for k, v in unsupported.items():
            if v is not None:
                raise NotImplementedError(f"TransformerBlock.forward({k}=...) is not on the HIP path")
        d = self.dim
        if self.global_cond_dim is not None and self.global_cond_dim > 0 and global_cond is not None:
            # adaLN: (to_scale_shift_gate + global_cond).chunk(6)   (transformer.py:677)
            # cast to the activation dtype: under bf16 autocast the fp32 parameter promotes the sum to fp32 while x is bf16
            mod = (self.to_scale_shift_gate + global_cond).to(x.dtype).contiguous()   # (B, 6D)
            scale_self, shift_self, gate_self = mod[:, 0:d], mod[:, d:2 * d], mod[:, 2 * d:3 * d]
            scale_ff, shift_ff, gate_ff = mod[:, 3 * d:4 * d], mod[:, 4 * d:5 * d], mod[:, 5 * d:6 * d]
            h = self.self_attn(self.pre_norm(x, scale_self, shift_self, fp8_for=self.self_attn.to_qkv), rotary_pos_emb=rotary_pos_emb)
            x = _GateResidualFn.apply(h, gate_self, x)                           # h*sigmoid(1-gate)+x  (:684-686)
            if context is not None and self.cross_attend:
                x = self.cross_attn(self.cross_attend_norm(x, fp8_for=self.cross_attn.to_q), context=context, res=x)   # never modulated (:688-689)
            h = self.ff(self.ff_norm(x, scale_ff, shift_ff, fp8_for=self.ff.ff[0].proj))
            x = _GateResidualFn.apply(h, gate_ff, x)
        else:
            # (fp8_for: each norm's only consumer — in fp8 inference the rows leave the LayerNorm kernel quantised)
            if torch.is_grad_enabled() and x.requires_grad and _ops().train_fused_nodes and _ops().ln_residual:
                # training: each norm also hands x through, so the residual path's gradient is added inside the LayerNorm backward kernel
                h, xr = self.pre_norm.with_residual(x)
                x = self.self_attn(h, rotary_pos_emb=rotary_pos_emb, res=xr)
                if context is not None and self.cross_attend:
                    h, xr = self.cross_attend_norm.with_residual(x)
                    x = self.cross_attn(h, context=context, res=xr)
                h, xr = self.ff_norm.with_residual(x)
                return self.ff(h, res=xr)
            x = self.self_attn(self.pre_norm(x, fp8_for=self.self_attn.to_qkv), rotary_pos_emb=rotary_pos_emb, res=x)   # residual adds live in
            if context is not None and self.cross_attend:                                                                 # the output projections' epilogues
                x = self.cross_attn(self.cross_attend_norm(x, fp8_for=self.cross_attn.to_q), context=context, res=x)
            x = self.ff(self.ff_norm(x, fp8_for=self.ff.ff[0].proj), res=x)
        return x


class ContinuousTransformer(nn.Module):
    def __init__(self, dim, depth, *, dim_in=None, dim_out=None, dim_heads=64, cross_attend=False, cond_token_dim=None,
                 final_cross_attn_ix=-1, global_cond_dim=None, causal=False, rotary_pos_emb=True,
                 zero_init_branch_outputs=True, conformer=False, use_sinusoidal_emb=False, use_abs_pos_emb=False,
                 abs_pos_emb_max_length=10000, num_memory_tokens=0, sliding_window=None, **kwargs):
        super().__init__()
        if causal or conformer or use_sinusoidal_emb or use_abs_pos_emb or num_memory_tokens or sliding_window is not None:
            raise NotImplementedError("causal / conformer / abs-pos-emb / memory tokens / sliding window are not on the HIP path")
        self.dim, self.depth, self.causal = dim, depth, False
        self.layers = nn.ModuleList([])
        self.project_in = Linear(dim_in, dim, bias=False) if dim_in is not None else nn.Identity()
        self.project_out = Linear(dim, dim_out, bias=False) if dim_out is not None else nn.Identity()
        self.rotary_pos_emb = RotaryEmbedding(max(dim_heads // 2, 32)) if rotary_pos_emb else None
        self.num_memory_tokens = 0
        self.global_cond_embedder = None
        if global_cond_dim is not None:
            self.global_cond_embedder = nn.Sequential(Linear(global_cond_dim, dim), nn.SiLU(), Linear(dim, dim * 6))
        self.final_cross_attn_ix = final_cross_attn_ix
        self.sliding_window = None
        self.checkpointing = False      # opt-in activation recompute per layer (see forward)
        for i in range(depth):
            should_cross_attend = cross_attend and (final_cross_attn_ix == -1 or i <= final_cross_attn_ix)
            self.layers.append(TransformerBlock(dim, dim_heads=dim_heads, cross_attend=should_cross_attend,
                                                dim_context=cond_token_dim, global_cond_dim=global_cond_dim,
                                                zero_init_branch_outputs=zero_init_branch_outputs, layer_ix=i, **kwargs))

    def forward(self, x, prepend_embeds=None, global_cond=None, return_info=False, use_checkpointing=True,
                exit_layer_ix=None, **kwargs):
        """Per-layer activation checkpointing (reference: transformer.py:28-30, :840-845, always on in training) is OPTIONAL here:
        `self.checkpointing` (default False — a depth-24 N = 6145 step keeps 36 GiB at batch 1 on a 288 GB part) AND the reference's
        `use_checkpointing` kwarg must both be true; results are identical either way (tests/test_boundary.py)."""
        model_dtype = next(self.parameters()).dtype
        x = x.to(model_dtype)
        info = {"hidden_states": []}
        x = self.project_in(x)
        if prepend_embeds is not None:
            assert prepend_embeds.shape[-1] == x.shape[-1], "prepend dimension must match sequence dimension"
            x = torch.cat((prepend_embeds, x), dim=-2)
        rotary = self.rotary_pos_emb.forward_from_seq_len(x.shape[1]) if self.rotary_pos_emb is not None else None
        if global_cond is not None and self.global_cond_embedder is not None:
            global_cond = self.global_cond_embedder(global_cond)
        x = x.contiguous()
        ckpt = self.checkpointing and use_checkpointing and torch.is_grad_enabled()
        for layer_ix, layer in enumerate(self.layers):
            if ckpt:
                x = torch.utils.checkpoint.checkpoint(layer, x, rotary_pos_emb=rotary, global_cond=global_cond, use_reentrant=False, **kwargs)
            else:
                x = layer(x, rotary_pos_emb=rotary, global_cond=global_cond, **kwargs)
            if return_info:
                info["hidden_states"].append(x)
            if exit_layer_ix is not None and layer_ix == exit_layer_ix:
                return (x, info) if return_info else x
        x = self.project_out(x)
        return (x, info) if return_info else x
