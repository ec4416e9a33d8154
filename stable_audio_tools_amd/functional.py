"""Autograd boundary of the Oobleck conv stack.

Each ``torch.autograd.Function`` is one *fusion unit* of the reference graph
(stable_audio_tools/models/autoencoders.py):

  WeightNormFn       weight_norm reparametrisation  (:23-27)
  SnakeConv1dFn      SnakeBeta -> WNConv1d [-> +residual] [-> tanh]   (:58-83, :233-250, :298-312, :333-354)
  SnakeConvTr1dFn    SnakeBeta -> WNConvTranspose1d                   (:266-271)
  ResidualUnitFn     x + conv1(snake(conv7_dil(snake(x))))            (:58-83)
  VaeSampleFn        VAEBottleneck.encode                             (models/bottleneck.py:105-133)

Forward and backward both run on the HIP kernels (ops.SatOps); the SnakeBeta activations are never
materialised — backward recomputes them inside the wgrad / dgrad kernels from the saved
pre-activation.  Where the reference wraps ResidualUnit in torch.utils.checkpoint
(autoencoders.py:78-79) we instead keep the one intermediate (h) resident: 288 GB of HBM makes
recompute the worse trade on MI355X.
"""
import torch

from . import ops as _ops_mod
from .ops import PACK_CONV_DGRAD, PACK_CONV_FWD, PACK_POLYPHASE


def _ops(ops):
    """An explicitly passed SatOps, else the product singleton (gfx950 library, or raises).  There is no test hook in this dispatch:
    the CPU test-suite substitutes `ops.get_ops` itself from its fixtures (tests/emu_util.use_emu_ops)."""
    return ops if ops is not None else _ops_mod.get_ops()


class SumAllFn(torch.autograd.Function):
    """sum over every element -> 0-d tensor on ops.sum_all (deterministic two-pass row sums; no torch multi-block reduction — see its
    docstring for why the training step avoids those).  Backward: the incoming scalar broadcast to the input's shape, as torch's.
    `ops`: the SatOps handle of the caller (None = the product singleton), as every other Function of this module takes it."""

    @staticmethod
    def forward(ctx, x, ops=None):
        ctx.in_shape, ctx.in_dtype = x.shape, x.dtype
        return _ops(ops).sum_all(x.detach())

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.in_dtype).expand(ctx.in_shape), None


def sum_all(x, ops=None):
    return SumAllFn.apply(x, ops)


def mean_all(x, ops=None):
    return SumAllFn.apply(x, ops) / x.numel()


class DerivedCache:
    """Derived copies of one conv layer's parameters (folded weight, packed bf16x3 planes, SnakeBeta constants) for the passes in
    which nothing is trained: each entry is valid for one (storage, torch version counter, invalidation epoch) of its sources
    (_caches.py explains the epoch).  A frozen pretransform (pretransforms.AutoencoderPretransform: requires_grad_(False).eval())
    therefore launches no sat_wn_fold / sat_pack / sat_snake_consts after its first call."""

    def __init__(self):
        self.items = {}

    def get(self, name, sources, make):
        from . import _caches
        if not _caches.trackable(*sources):
            # inference tensors (parameters created / loaded under torch.inference_mode) carry no version counter: an in-place edit
            # would leave no trace, so nothing derived from them is kept
            return make()
        key = tuple((t.data_ptr(), _caches.version_of(t), t.device) for t in sources if t is not None) + _caches.epoch_of(*sources)
        hit = self.items.get(name)
        if hit is None or hit[0] != key:
            hit = (key, make())
            self.items[name] = hit
        return hit[1]


def _cached(cache, name, sources, make):
    return make() if cache is None else cache.get(name, sources, make)


class WeightNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, g, ops=None):
        ops = _ops(ops)
        v = v.contiguous()
        gf = g.contiguous().view(-1)
        w, norm = ops.wn_fold(v, gf)
        ctx.save_for_backward(v, gf, norm)
        ctx.ops = ops
        ctx.g_shape = g.shape
        return w

    @staticmethod
    def backward(ctx, dw):
        v, gf, norm = ctx.saved_tensors
        dv, dg = ctx.ops.wn_grad(v, gf, norm, dw.contiguous())
        return dv, dg.view(ctx.g_shape), None


def _conv_dgrad(ops, dy, w, k, stride, dil, pad, cin, tin, dsnake, res=None, out=None, emit=None):
    """dL/d(conv input pre-activation) of a Conv1d with torch weight w (Cout, Cin, K).  out= (stride-1 kernels only): write into
    the caller's tensor (it may alias `res`: accumulation in place).  emit: {"snake": None | (la, lb)} — also write the result as
    the activation planes of the k7 conv that consumes it next (ops.emit_ok decides)."""
    if stride == 1:
        if (res is None and out is None and emit is None and ops.edge_ok(w.shape[0], cin, k, 1, dil, (k - 1) * dil - pad) and tin == dy.shape[2]
                and (dsnake is None or w.shape[0] <= 2)):
            # the data-gradient of a conv with a two-channel side = that side's edge conv on the transposed, tap-flipped weight
            return ops.edge_conv(dy, w, (k - 1) * dil - pad, mode=1, dsnake=dsnake)
        if ops.bf16x3_ok(k, 1, dil):
            q = ops.k7q_applicable(w.shape[0], k, 1, dil, (k - 1) * dil - pad, cin)  # the data-gradient's input channels = Cout
            return ops.conv1d_bf16x3(dy, ops.pack_bf16x3(w, mode=1, q=q), cin, k, 1, dil, (k - 1) * dil - pad, tout=tin,
                                     dsnake=dsnake, res=res, out=out, emit=emit)
        wpb = ops.pack(w, PACK_CONV_DGRAD)
        return ops.conv1d(dy, wpb, cin, k, 1, dil, (k - 1) * dil - pad, tout=tin, dsnake=dsnake, res=res, out=out)
    if out is not None:
        raise NotImplementedError("_conv_dgrad(out=...) is only plumbed through the stride-1 kernels")
    if ops.bf16x3_ok(k, stride, dil, transposed=True):   # (Cout, Cin, K) is the [in][out][K] weight of the transposed conv
        return ops.convtr1d_bf16x3(dy, ops.pack_bf16x3(w, mode=2, stride=stride), cin, k, stride, pad, tout=tin,
                                   dsnake=dsnake, res=res)
    wpb = ops.pack(w, PACK_POLYPHASE, stride)
    return ops.convtr1d(dy, wpb, cin, k, stride, pad, tout=tin, dsnake=dsnake, res=res)


def _conv_wgrad(ops, dy, x, k, stride, dil, pad, snake, bias_grad=False, raw=False):
    """dL/dW (Cout, Cin, K) of conv1d(snake(x)): k7 (dilation 1/3/9) -> the 7-tap bf16x3 kernels; k1 and K = 2*stride ->
    the short-kernel bf16x3 wgrad (chosen inside ops.conv_wgrad); anything else -> the fp32-MFMA kernel.
    bias_grad=True returns (dW, dbias): the bf16x3 kernels sum the dy rows they stream anyway.
    raw=True: dW stays the kernel's split slabs (ops.WgradSlabs), and dbias the per-split sums (C, R) one reduction short of the
    gradient — what _wn_backward takes."""
    if ops.edge_ok(x.shape[1], dy.shape[1], k, stride, dil, pad) and dy.shape[2] == x.shape[2] and (snake is None or dy.shape[1] <= 2):
        return ops.edge_conv_wgrad(dy, x, k, pad, snake=snake, dy_rowsum=bias_grad, raw=raw)      # a two-channel side: fp32 FMA stream
    if ops.wgrad7_bf16x3_ok(x.shape[1], k, stride, dil):
        return ops.conv_wgrad7_bf16x3(dy, x, dil, pad, snake=snake, dy_rowsum=bias_grad, raw=raw)
    return ops.conv_wgrad(dy, x, k, stride, dil, pad, snake=snake, snake_on=2, lo_rowsum=bias_grad, raw=raw)


def _wn_forward(ops, v, g):
    """Weight norm INSIDE a conv's autograd unit (round 5): the unit takes (weight_v, weight_g) in place of the folded weight, folds in
    its forward (sat_wn_fold) and, in its backward, turns the weight-gradient kernel's split slabs straight into (dv, dg)
    (sat_wn_grad_splits) — no separate WeightNormFn node, no sat_reduce_splits / permute / sat_wn_grad launches, no dW in HBM.
    Returns (w, (v, g_flat, norm))."""
    v = v.contiguous()
    gf = g.contiguous().view(-1)
    w, norm = ops.wn_fold(v, gf)
    return w, (v, gf, norm)


def _wn_backward(ops, slabs, saved, g_shape, bias_partial=None):
    """(dv, dg[, dbias]) from the slabs (+ the per-split sums of dy a raw _conv_wgrad(bias_grad=True) returned beside them)."""
    v, gf, norm = saved
    out = ops.wn_grad_splits(slabs, v, gf, norm, bias_partial=bias_partial)
    return (out[0], out[1].view(g_shape)) + tuple(out[2:])


def _conv_fwd(ops, x, w, stride, dil, pad, bias=None, snake=None, res=None, tanh_out=False, dsnake=None, tout=None, cache=None, emit=None):
    """conv1d(snake(x), w) [+bias] [+res]: the bf16x3 split-MFMA kernel (fp32-accurate, csrc/conv1d_bf16x3.hip) where
    its shape rules allow, else the fp32-MFMA kernel (csrc/conv1d.hip).  cache: DerivedCache of the layer (no-grad / frozen
    passes only) for the packed planes and the SnakeBeta constants."""
    cout, cin, k = w.shape
    if (res is None and ops.edge_ok(cin, cout, k, stride, dil, pad) and (tout is None or tout == x.shape[2])
            and (snake is None or cout <= 2) and (dsnake is None or cin <= 2) and (emit is None or cin <= 2)):
        # a two-channel end of the stack (the encoder's first conv, the decoder's last): fp32 FMA stream straight from the folded weight
        return ops.edge_conv(x, w, pad, bias=bias, snake=snake, tanh_out=tanh_out, dsnake=dsnake, emit=emit)
    if ops.bf16x3_ok(k, stride, dil):
        q = ops.k7q_applicable(cin, k, stride, dil, pad, cout)
        planes = _cached(cache, "pack_fwd_q" if q else "pack_fwd", (w,), lambda: ops.pack_bf16x3(w, stride=stride, q=q))
        sconsts = _cached(cache, "snake", snake, lambda: ops.snake_consts(snake[0], snake[1])) if snake is not None else None
        return ops.conv1d_bf16x3(x, planes, cout, k, stride, dil, pad, tout=tout, bias=bias,
                                 snake=snake, res=res, tanh_out=tanh_out, dsnake=dsnake, sconsts=sconsts, emit=emit)
    return ops.conv1d(x, _cached(cache, "pack_fwd32", (w,), lambda: ops.pack(w, PACK_CONV_FWD)), cout, k, stride, dil, pad, tout=tout,
                      bias=bias, snake=snake, res=res, tanh_out=tanh_out, dsnake=dsnake)


def _convtr_fwd(ops, x, w, stride, pad, bias=None, snake=None, cache=None):
    """conv_transpose1d(snake(x), w (Cin, Cout, K)) [+bias]."""
    cin, cout, k = w.shape
    if ops.bf16x3_ok(k, stride, 1, transposed=True):
        planes = _cached(cache, "pack_tr", (w,), lambda: ops.pack_bf16x3(w, mode=2, stride=stride))
        sconsts = _cached(cache, "snake", snake, lambda: ops.snake_consts(snake[0], snake[1])) if snake is not None else None
        return ops.convtr1d_bf16x3(x, planes, cout, k, stride, pad, bias=bias, snake=snake, sconsts=sconsts)
    return ops.convtr1d(x, _cached(cache, "pack_tr32", (w,), lambda: ops.pack(w, PACK_POLYPHASE, stride)), cout, k, stride, pad,
                        bias=bias, snake=snake)


class SnakeConv1dFn(torch.autograd.Function):
    """y = tanh?( conv1d(snake(x; alpha, beta), w, bias, stride, dil, pad) + res )."""

    @staticmethod
    def forward(ctx, x, alpha, beta, w, bias, res, stride, dil, pad, tanh_out, ops=None, cache=None, next_snake=None, g=None):
        """next_snake = (log-alpha, log-beta, dilation) of the ResidualUnit that reads y next (its k7 conv): the conv also emits
        snake(y) as that conv's activation planes (no autograd through them: the consumer's own backward recomputes from y).
        g: weight-norm magnitudes — `w` is then weight_v and the unit folds / un-folds itself (_wn_forward)."""
        ops = _ops(ops)
        x = x.contiguous()
        wn = ()
        if g is not None:
            w, wn = _wn_forward(ops, w, g)
        w = w.contiguous()
        cout, cin, k = w.shape
        snake = (alpha.contiguous(), beta.contiguous()) if alpha is not None else None
        emit = None
        if next_snake is not None and not tanh_out:
            tout_ = (x.shape[2] + 2 * pad - dil * (k - 1) - 1) // stride + 1
            if ((ops.emit_ok(cout, k, stride, tout_, next_snake[2]) and ops.bf16x3_ok(k, stride, dil))
                    or (res is None and snake is None and ops.edge_emit_ok(cin, cout, k, stride, dil, pad, next_snake[2]))):
                emit = {"snake": (next_snake[0].detach(), next_snake[1].detach())}
        y = _conv_fwd(ops, x, w, stride, dil, pad, bias=bias, snake=snake,
                      res=res.contiguous() if res is not None else None, tanh_out=tanh_out, cache=cache, emit=emit)
        ctx.ops = ops
        ctx.cfg = (stride, dil, pad, tanh_out, bias is not None, res is not None, alpha is not None)
        ctx.g_shape = g.shape if g is not None else None
        ctx.save_for_backward(x, alpha, beta, w, y if tanh_out else None, *wn)
        return y

    @staticmethod
    def backward(ctx, dy):
        ops = ctx.ops
        stride, dil, pad, tanh_out, has_bias, has_res, has_snake = ctx.cfg
        x, alpha, beta, w, y = ctx.saved_tensors[:5]
        wn = ctx.saved_tensors[5:]
        cout, cin, k = w.shape
        dy = dy.contiguous()
        if tanh_out:
            dy = dy * (1.0 - y * y)
        snake = (alpha, beta) if has_snake else None
        dres = dy if has_res else None
        dw = dbias = dg = None
        if ctx.needs_input_grad[3] or (wn and ctx.needs_input_grad[13]):
            dw = _conv_wgrad(ops, dy, x, k, stride, dil, pad, snake, bias_grad=has_bias, raw=bool(wn))
            if has_bias:
                dw, dbias = dw
            if wn:
                dw, dg, *rest = _wn_backward(ops, dw, wn, ctx.g_shape, bias_partial=dbias)
                if has_bias:
                    dbias = rest[0]
        elif has_bias:
            dbias = ops.rowsum(dy)
        dx = da = db = None
        if has_snake:
            dx, da, db = _conv_dgrad(ops, dy, w, k, stride, dil, pad, cin, x.shape[2], (x, alpha, beta))
        elif ctx.needs_input_grad[0]:
            dx = _conv_dgrad(ops, dy, w, k, stride, dil, pad, cin, x.shape[2], None)
        return dx, da, db, dw, dbias, dres, None, None, None, None, None, None, None, dg


class SnakeConvTr1dFn(torch.autograd.Function):
    """y = conv_transpose1d(snake(x), w (Cin, Cout, K=2*stride), bias, stride, pad)."""

    @staticmethod
    def forward(ctx, x, alpha, beta, w, bias, stride, pad, ops=None, cache=None, g=None):
        ops = _ops(ops)
        x = x.contiguous()
        wn = ()
        if g is not None:           # w is weight_v: fold here, un-fold in the backward (_wn_forward)
            w, wn = _wn_forward(ops, w, g)
        w = w.contiguous()
        cin, cout, k = w.shape
        snake = (alpha.contiguous(), beta.contiguous()) if alpha is not None else None
        y = _convtr_fwd(ops, x, w, stride, pad, bias=bias, snake=snake, cache=cache)
        ctx.ops = ops
        ctx.cfg = (stride, pad, bias is not None, alpha is not None)
        ctx.g_shape = g.shape if g is not None else None
        ctx.save_for_backward(x, alpha, beta, w, *wn)
        return y

    @staticmethod
    def backward(ctx, dy):
        ops = ctx.ops
        stride, pad, has_bias, has_snake = ctx.cfg
        x, alpha, beta, w = ctx.saved_tensors[:4]
        wn = ctx.saved_tensors[4:]
        cin, cout, k = w.shape
        dy = dy.contiguous()
        snake = (alpha, beta) if has_snake else None
        dbias = ops.rowsum(dy) if has_bias else None
        dw = ops.conv_wgrad(x, dy, k, stride, 1, pad, snake=snake, snake_on=1, raw=bool(wn))
        dg = None
        if wn:
            dw, dg = _wn_backward(ops, dw, wn, ctx.g_shape)
        # dgrad of a transposed conv is the strided conv with in=Cout, out=Cin: w (Cin, Cout, K) is its [out][in][K] weight
        dx = da = db = None
        if has_snake:
            dx, da, db = _conv_fwd(ops, dy, w, stride, 1, pad, dsnake=(x, alpha, beta), tout=x.shape[2])
        elif ctx.needs_input_grad[0]:
            dx = _conv_fwd(ops, dy, w, stride, 1, pad, tout=x.shape[2])
        return dx, da, db, dw, dbias, None, None, None, None, dg


class ResidualUnitFn(torch.autograd.Function):
    """y = x + conv1x1(snake2(conv7_dil(snake1(x))))  — one unit, one saved intermediate (h = the k7 conv's output).
    recompute=True: h is NOT kept; the backward runs the k7 conv again (what the reference's torch.utils.checkpoint around
    `self.layers` does, autoencoders.py:78-79, minus the k1 conv it also repeats) — half the activation memory of the unit for
    +1/3 of its forward flops.  caches = (DerivedCache of the k7 conv, of the k1 conv) or None."""

    @staticmethod
    def forward(ctx, x, a1, b1, w1, bias1, a2, b2, w2, bias2, dil, ops=None, recompute=False, caches=None, next_snake=None, fuse=False,
                g1=None, g2=None):
        """next_snake = (log-alpha, log-beta, dilation) of the ResidualUnit that follows: the k1 conv's epilogue then also writes
        snake(y) as that unit's k7 activation planes (its sat_conv1d_k7_planes pre-pass disappears).
        fuse: run the unit as ONE launch where the kernel allows (C <= 128).  The caller asks for it when no backward will follow
        (inference: h is not kept and never touches HBM — 2.30 vs 2.65 ms at C = 128, T = 2 097 152); with h kept the fused launch
        is slower than the two launches (3.06 ms: its extra 1-GB store burst is not overlapped, profiles/EXPERIMENTS.md).
        g1 / g2: weight-norm magnitudes of the two convs — w1 / w2 are then their weight_v (_wn_forward)."""
        ops = _ops(ops)
        x = x.contiguous()
        wn1 = wn2 = ()
        if g1 is not None:
            w1, wn1 = _wn_forward(ops, w1, g1)
        if g2 is not None:
            w2, wn2 = _wn_forward(ops, w2, g2)
        w1 = w1.contiguous()
        w2 = w2.contiguous()
        c = x.shape[1]
        k1 = w1.shape[2]
        pad = dil * (k1 - 1) // 2
        c1, c2 = caches if caches is not None else (None, None)
        emit = None
        if next_snake is not None and ops.emit_ok(c, w2.shape[2], 1, x.shape[2], next_snake[2]):
            emit = {"snake": (next_snake[0].detach(), next_snake[1].detach())}
        if fuse and w2.shape[2] == 1 and w1.shape[0] == c and ops.ru_fused_ok(c, k1, dil, x.shape[2]):
            # C <= 128: the whole unit in ONE launch (csrc/conv1d_bf16x3_k7q.h, FUSED) — the k1 launch and its read of h disappear
            w7q = _cached(c1, "pack_q7", (w1,), lambda: ops.pack_k7q(w1))
            w1q = _cached(c2, "pack_q1", (w2,), lambda: ops.pack_k7q(w2))
            sc = None
            if c1 is not None and c2 is not None:
                sc = (_cached(c1, "snake", (a1, b1), lambda: ops.snake_consts(a1, b1)), _cached(c2, "snake", (a2, b2), lambda: ops.snake_consts(a2, b2)))
            h, y = ops.residual_unit_fwd(x, (a1, b1), w7q, bias1, (a2, b2), w1q, bias2, k1, dil, keep_h=ctx is not None and not recompute and fuse != "nokeep",
                                         emit=emit, sconsts=sc)
        else:
            h = _conv_fwd(ops, x, w1, 1, dil, pad, bias=bias1, snake=(a1, b1), cache=c1)
            y = _conv_fwd(ops, h, w2, 1, 1, 0, bias=bias2, snake=(a2, b2), res=x, cache=c2, emit=emit)
        ctx.ops = ops
        ctx.dil = dil
        ctx.recompute = bool(recompute)
        ctx.g_shapes = (g1.shape if g1 is not None else None, g2.shape if g2 is not None else None)
        ctx.save_for_backward(x, None if recompute else h, a1, b1, w1, a2, b2, w2, bias1 if recompute else None, *wn1, *wn2)
        return y

    @staticmethod
    def backward(ctx, dy):
        ops = ctx.ops
        x, h, a1, b1, w1, a2, b2, w2, bias1 = ctx.saved_tensors[:9]
        gs1, gs2 = ctx.g_shapes
        rest = ctx.saved_tensors[9:]
        wn1 = rest[:3] if gs1 is not None else ()
        wn2 = rest[len(wn1):] if gs2 is not None else ()
        dil = ctx.dil
        c = x.shape[1]
        k1, k2 = w1.shape[2], w2.shape[2]
        pad1 = dil * (k1 - 1) // 2
        t = x.shape[2]
        dy = dy.contiguous()
        if ctx.recompute:
            h = _conv_fwd(ops, x, w1, 1, dil, pad1, bias=bias1, snake=(a1, b1))
        # the k1 data-gradient also writes dh as the planes its consumer — the k7 data-gradient two launches below — reads
        want_emit = ops.emit_ok(w1.shape[0], k2, 1, t, dil) and ops.k7q_applicable(w1.shape[0], k1, 1, dil, pad1, c)
        if k2 == 1 and w2.shape[0] == c and w1.shape[0] == c and ops.ru_k1_bwd_ok(dy.shape[0], c, t):
            # C == 128 (the widest levels): weight gradient, data gradient, both bias gradients and the snake gradients of the 1x1 conv in ONE pass
            # over dy and h (csrc/ru_k1_bwd.hip) instead of three kernels that each stream them from HBM
            dh, da2, db2, dw2, dbias2, dbias1 = ops.ru_k1_bwd(dy, h, w2, (a2, b2), emit=want_emit, raw=bool(wn2))
            dw1 = _conv_wgrad(ops, dh, x, k1, 1, dil, pad1, (a1, b1), bias_grad=False, raw=bool(wn1))
        else:
            dw2, dbias2 = ops.conv_wgrad(dy, h, k2, 1, 1, 0, snake=(a2, b2), snake_on=2, lo_rowsum=True, raw=bool(wn2))
            dh, da2, db2 = _conv_dgrad(ops, dy, w2, k2, 1, 1, 0, c, t, (h, a2, b2), emit={"snake": None} if want_emit else None)
            dw1, dbias1 = _conv_wgrad(ops, dh, x, k1, 1, dil, pad1, (a1, b1), bias_grad=True, raw=bool(wn1))
        dg1 = dg2 = None
        if wn1:         # a raw weight gradient brings its bias gradient as per-split sums (C, R): finished in the same launch
            if dbias1.dim() == 2:
                dw1, dg1, dbias1 = _wn_backward(ops, dw1, wn1, gs1, bias_partial=dbias1)
            else:
                dw1, dg1 = _wn_backward(ops, dw1, wn1, gs1)
        if wn2:
            if dbias2.dim() == 2:
                dw2, dg2, dbias2 = _wn_backward(ops, dw2, wn2, gs2, bias_partial=dbias2)
            else:
                dw2, dg2 = _wn_backward(ops, dw2, wn2, gs2)
        dx, da1, db1 = _conv_dgrad(ops, dh, w1, k1, 1, dil, pad1, c, t, (x, a1, b1), res=dy)
        return dx, da1, db1, dw1, dbias1, da2, db2, dw2, dbias2, None, None, None, None, None, None, dg1, dg2


class VaeSampleFn(torch.autograd.Function):
    """(z, kl) = vae_sample(chunk(pre, 2, dim=1)) with the N(0,1) draw supplied by the caller."""

    @staticmethod
    def forward(ctx, pre, noise, ops=None):
        ops = _ops(ops)
        pre = pre.contiguous()
        noise = noise.contiguous()
        z, kl = ops.vae_sample_fwd(pre, noise)
        ctx.ops = ops
        ctx.save_for_backward(pre, noise)
        return z, kl

    @staticmethod
    def backward(ctx, dz, dkl):
        pre, noise = ctx.saved_tensors
        dpre = ctx.ops.vae_sample_bwd(pre, noise, dz.contiguous() if dz is not None else None,
                                      dkl.contiguous() if dkl is not None else None)
        return dpre, None, None
