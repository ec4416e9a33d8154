"""nn.Linear of the DiT on the native MFMA GEMM (csrc/gemm.hip) — forward, data gradient and weight gradient — with the
elementwise work that follows a projection fused into its epilogue.

Reference call sites (stable_audio_tools/models/transformer.py): to_qkv :362/:481, to_out :364/:534, to_q / to_kv :356-357,
GLU.proj + x*silu(gate) :263-275, FeedForward linear_out :308, the residual / gate updates of TransformerBlock :684-712;
models/dit.py: to_timestep_embed / to_cond_embed / to_global_embed :49-77.

`Linear` keeps nn.Linear's parameter names (`weight` (out, in), `bias`) so state_dict keys are the reference's.

Precision modes
  * bf16 tensors, or fp32 master weights under torch.autocast(bf16): one bf16 MFMA per product, fp32 accumulation; weights
    are cast once per optimizer step (cache keyed on the parameter version and the optimizer epoch, `bump_weight_epoch`).
  * fp32 tensors: the same kernel on bf16x3-split operands ([hi|hi|lo] x [hi|lo|hi] along K): three MFMAs per product,
    ~2^-17 relative per product — float32-class results for the 1e-3 parity bar.
One kernel computes all three GEMMs of a layer: it is "NT" (both operands K-contiguous), which is nn.Linear's forward as
stored; the data gradient runs on the transposed weight copy and the weight gradient on transposed activations
(sat_cast_bf16 with transpose, zero padded along the reduction dim).
"""

import torch
from torch import nn

from . import _caches
from . import functional as _fn
from ._caches import bump_weight_epoch  # noqa: F401  (re-export: training.FusedAdamW, tests)


def _ops():
    return _fn._ops(None)


def _pad8(t):
    """Pad the last dim of a 2-D tensor to a multiple of 8 with zeros (K of the GEMM must be a multiple of 8)."""
    k = t.shape[1]
    if k % 8 == 0:
        return t
    return torch.nn.functional.pad(t, (0, 8 - k % 8))


def _pad_rows8(t):
    n = t.shape[0]
    if n % 8 == 0:
        return t
    return torch.nn.functional.pad(t, (0, 0, 0, 8 - n % 8))


class _WeightCache:
    """Low-precision / transposed / split copies of a module's parameters; each entry is valid for one
    (storage, version, dtype, invalidation epoch) of the tensor it was made from — the epoch (_caches.py) covers the writers
    torch's version counter does not see: the fused optimizer kernel and `.data` updates (ema_pytorch)."""

    def __init__(self):
        self.items = {}

    def get(self, w, name, make):
        if not _caches.trackable(w):          # a parameter created under torch.inference_mode: no version counter, nothing is kept
            return make()
        key = (w.data_ptr(), _caches.version_of(w), w.dtype, w.device, _caches.epoch_of(w))
        hit = self.items.get(name)
        if hit is None or hit[0] != key:
            hit = (key, make())
            self.items[name] = hit
        return hit[1]

    def put(self, w, name, value):
        """Store an entry made together with another one (the transposed weight copy that rides in the forward's cast launch)."""
        if _caches.trackable(w):
            self.items[name] = ((w.data_ptr(), _caches.version_of(w), w.dtype, w.device, _caches.epoch_of(w)), value)


# activations quantised per ROW in one pass (round 4); False (set by an A/B script): round 3's per-tensor scale (absmax pass + quant pass)
fp8_row_scales = True


class Fp8Rows:
    """An activation that already left its producer as fp8 e4m3 rows + one dynamic scale per row (LayerNorm -> fp8, round 4): what an
    fp8 `Linear` / the attention input projection take instead of a tensor in inference.  Carries only what those two consumers read."""

    def __init__(self, q, scale, shape):
        self.q, self.scale, self.shape = q, scale, tuple(shape)       # q (rows, K) uint8, scale (rows,) fp32, logical shape (..., K)
        self.dtype = torch.bfloat16                                   # (the precision mode the consumer runs in)
        self.device = q.device
        self.is_cuda = q.is_cuda


# weights quantised per OUTPUT CHANNEL (round 6: one dynamic scale per weight row — `col_alpha` of the fp8 GEMM epilogues — instead of
# one per tensor: a channel's largest weight no longer sets every other channel's step size); False (A/B scripts): round 3's per-tensor scale
fp8_channel_scales = True


def _quant_weight_fp8(ops, w):
    """(uint8 e4m3 weight rows padded to a multiple of 8, de-quantisation scale): a (rows,) vector with fp8_channel_scales (the K <= 8192
    shapes sat_quant_fp8_rows takes), else a 0-dim per-tensor scale."""
    wb = _pad_rows8(w.detach() if w.dtype == torch.bfloat16 else ops.cast_bf16(w.detach()))
    if fp8_channel_scales and wb.shape[1] % 8 == 0 and wb.stride(0) % 8 == 0 and wb.shape[1] <= 8192:
        return ops.quant_fp8_rows(wb)
    return ops.quant_fp8(wb)


def fp8_scales(sa, sb):
    """(alpha, col_alpha) of an fp8 GEMM from the activation's per-tensor scale `sa` (None when it is quantised per row: row_alpha) and
    the weight's scale `sb` (0-dim: per tensor; 1-dim: per output channel)."""
    if sb.dim() == 0:
        return (sb if sa is None else sa * sb), None
    return sa, sb


def _fp8_operands(ops, x2, w, cache):
    """(A, B, alpha, row_alpha, col_alpha) in fp8 e4m3 with dynamic scales: weights per output channel (or per tensor), once per version;
    activations per call and per ROW (one pass, a token's outliers do not set the other tokens' step size) — or per tensor with
    `fp8_row_scales = False`."""
    xb = x2 if x2.dtype == torch.bfloat16 else ops.cast_bf16(x2)
    b, sb = cache.get(w, "fp8", lambda: _quant_weight_fp8(ops, w))
    if fp8_row_scales and xb.shape[1] % 8 == 0 and xb.stride(0) % 8 == 0 and xb.shape[1] <= 8192:
        a, ra = ops.quant_fp8_rows(xb)
        alpha, ca = fp8_scales(None, sb)
        return a, b, alpha, ra, ca
    a, sa = ops.quant_fp8(xb)
    alpha, ca = fp8_scales(sa, sb)
    return a, b, alpha, None, ca


def _lowp_weight_pair(ops, w, cache):
    """The bf16 weight of a training forward; its transposed copy for the data-gradient GEMM (LinearFn.backward's "bf16_t" entry) comes
    out of the same launch when the shape allows (round 6: one pass over the fp32 master weight per optimizer step instead of two)."""
    if w.dtype == torch.float32 and w.shape[0] % 8 == 0 and w.shape[1] % 8 == 0:
        pair = ops.cast_bf16_dual(w.detach())
        if pair is not None:
            cache.put(w, "bf16_t", pair[1])
            return pair[0]
    return _pad_rows8(_pad8(w.detach() if w.dtype == torch.bfloat16 else ops.cast_bf16(w.detach())))


def _operands(ops, x2, w, cache, lowp, want_t=False):
    """GEMM operands (A, B) for y = x2 · w^T in the chosen precision mode.  want_t: the caller's backward will ask for the transposed weight."""
    if lowp:
        a = x2 if x2.dtype == torch.bfloat16 else ops.cast_bf16(x2)
        a = _pad8(a)
        if want_t and ops.train_fused_nodes:
            b = cache.get(w, "bf16", lambda: _lowp_weight_pair(ops, w, cache))
        else:
            b = cache.get(w, "bf16", lambda: _pad_rows8(_pad8(w.detach() if w.dtype == torch.bfloat16 else ops.cast_bf16(w.detach()))))
        return a, b
    a = ops.split_bf16x3(_pad8(x2.float()).contiguous(), 0)
    b = cache.get(w, "x3", lambda: ops.split_bf16x3(_pad_rows8(_pad8(w.detach().float())).contiguous(), 1))
    return a, b


_bias_operands = {}


def _bias_operand(device, k, k8, mp, m):
    """The (k8 + 8, Mp) transposed-activation operand of a weight-gradient GEMM that also yields the bias gradient (row k8 = ones over
    the m tokens): rows k .. k8 + 7 never change, so the buffer is kept per shape and only rows [:k] are rewritten by each call's cast
    (two fills per call before).  Reuse is stream-ordered: the GEMM that read the previous contents is enqueued before the next cast."""
    key = (device, k, mp, m)
    buf = _bias_operands.get(key)
    if buf is None:
        if len(_bias_operands) >= 16:
            _bias_operands.clear()
        buf = torch.zeros(k8 + 8, mp, dtype=torch.bfloat16, device=device)
        buf[k8, :m] = 1.0
        _bias_operands[key] = buf
    return buf


def _wgrad_splits(n, k, red):
    tiles = ((n + 127) // 128) * ((k + 127) // 128)
    return max(1, min(8, 256 // max(tiles, 1), (red + 511) // 512))


class LinearFn(torch.autograd.Function):
    """y = epilogue(x · W^T + b).   mode: 'plain' | 'res' (y += res) | 'swiglu' (y = v * silu(g), W rows = [v | g])."""

    @staticmethod
    def forward(ctx, x, weight, bias, res, mode, lowp, cache, fp8=False):
        ops = _ops()
        shp = x.shape
        k = shp[-1]
        n = weight.shape[0]
        x2 = x.reshape(-1, k)
        if x2.stride(1) != 1:
            x2 = x2.contiguous()
        out_dtype = torch.bfloat16 if lowp else torch.float32
        fp8 = bool(fp8) and lowp and k % 16 == 0
        if fp8:
            a, b, alpha, ralpha, calpha = _fp8_operands(ops, x2, weight, cache)
        else:
            a, b = _operands(ops, x2, weight, cache, lowp, want_t=ctx.needs_input_grad[0])
        bias32 = None
        if bias is not None:
            bias32 = cache.get(bias, "bias32", lambda: torch.nn.functional.pad(bias.detach().float(), (0, (-n) % 8)).contiguous())
        need_grad = any(ctx.needs_input_grad[:4])
        r2 = None
        if mode == "res":
            r2 = res.reshape(-1, n).to(out_dtype)
            if r2.stride(1) != 1:
                r2 = r2.contiguous()
            if n % 8:
                r2 = _pad8(r2)
        pre = None
        nout = n // 2 if mode == "swiglu" else n
        npad = b.shape[0] // 2 if mode == "swiglu" else b.shape[0]
        # the result is allocated in its final shape (an output that is a *view* of a tensor made inside a custom Function
        # cannot be modified in place later, e.g. by the in-place rotary)
        yfull = torch.empty(*shp[:-1], nout, dtype=out_dtype, device=x.device)
        out2 = yfull.view(-1, nout) if npad == nout else None
        if mode == "swiglu":
            if n % 16:
                raise ValueError("SwiGLU projection needs 2F output rows with F % 8 == 0")
            if fp8:
                y = ops.gemm_fp8(a, b, alpha, bias=bias32, epilogue=ops.EPI_SWIGLU, out_dtype=out_dtype, want_pre=need_grad, out=out2, row_alpha=ralpha,
                                 col_alpha=calpha)
            else:
                y = ops.gemm_bf16(a, b, bias=bias32, epilogue=ops.EPI_SWIGLU, out_dtype=out_dtype, want_pre=need_grad, out=out2)
            if need_grad:
                y, pre = y
        elif fp8:
            y = ops.gemm_fp8(a, b, alpha, bias=bias32, res=r2, epilogue=ops.EPI_RES if mode == "res" else ops.EPI_STORE, out_dtype=out_dtype,
                             out=out2, row_alpha=ralpha, col_alpha=calpha)
        elif lowp and out2 is not None and ops.splitk_for(a.shape[0], b.shape[0], a.shape[1]) > 1:
            y = ops.gemm_bf16_splitk(a, b, ops.splitk_for(a.shape[0], b.shape[0], a.shape[1]), bias=bias32, res=r2, out_dtype=out_dtype, out=out2)
        else:
            y = ops.gemm_bf16(a, b, bias=bias32, res=r2, epilogue=ops.EPI_RES if mode == "res" else ops.EPI_STORE, out_dtype=out_dtype,
                              out=out2)
        if out2 is None:
            yfull.view(-1, nout).copy_(y[:, :nout])
        if need_grad:
            ctx.save_for_backward(x2, weight, pre)
            ctx.meta = (ops, shp, mode, lowp, cache, bias is not None, res is not None, x.dtype, weight.dtype,
                        bias.dtype if bias is not None else None, res.dtype if res is not None else None, res.shape if res is not None else None)
        return yfull

    @staticmethod
    def backward(ctx, dy):
        x2, weight, pre = ctx.saved_tensors
        ops, shp, mode, lowp, cache, has_bias, has_res, xdt, wdt, bdt, rdt, rshape = ctx.meta
        n, k = weight.shape
        m = x2.shape[0]
        dy2 = dy.reshape(m, -1)
        if dy2.stride(1) != 1:
            dy2 = dy2.contiguous()
        dres = dy.to(rdt).reshape(rshape) if has_res and ctx.needs_input_grad[3] else None
        if mode == "swiglu":
            dz = ops.swiglu_bwd(pre, dy2.to(pre.dtype).contiguous())          # (M, 2F): d/d[v | g]
        else:
            dz = dy2
        dx = dw = db = None
        if lowp:
            dzb = dz if dz.dtype == torch.bfloat16 else ops.cast_bf16(dz.contiguous())
            if ctx.needs_input_grad[0]:
                # dx (M, K) = dz (M, N) · W (N, K): NT form with B = W^T (K, N)
                wt = cache.get(weight, "bf16_t", lambda: _pad8(ops.cast_bf16(weight.detach(), transpose=True, row_pad=8)))
                dx = ops.gemm_bf16(_pad8(dzb), _pad_rows8(wt), out_dtype=torch.bfloat16)[:, :k]
            if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
                # dW (N, K) = dz^T (N, Mp) · x^T (K, Mp)^T, reduction over the zero-padded token dim.  The bias gradient rides in
                # the same GEMM: one extra "activation" row of ones makes column K of the result the column sum of dz.
                k8 = (k + 7) // 8 * 8
                if ops.train_fused_nodes:
                    # both transposed operands from ONE launch (sat_cast_bf16_tpair); the rows of x^T past k (zero padding, the bias row of
                    # ones) live in a cached buffer
                    mp = (m + 7) // 8 * 8
                    if has_bias:
                        xt = _bias_operand(dzb.device, k, k8, mp, m)
                    else:
                        xt = torch.empty(k8, mp, dtype=torch.bfloat16, device=dzb.device)
                        if k8 != k:
                            xt[k:].zero_()
                    if ops.cast_pair:
                        dzt, _ = ops.cast_bf16_tpair(dzb, x2, row_pad=8, out_b=xt[:k])
                    else:
                        dzt = ops.cast_bf16(dzb, transpose=True, row_pad=8)
                        ops.cast_bf16(x2, transpose=True, row_pad=8, out=xt[:k])
                    dzt = _pad_rows8(dzt)                                                    # (N8, Mp)
                else:
                    dzt = _pad_rows8(ops.cast_bf16(dzb, transpose=True, row_pad=8))          # (N8, Mp)
                    xt = torch.empty(k8 + (8 if has_bias else 0), dzt.shape[1], dtype=torch.bfloat16, device=dzt.device)
                    ops.cast_bf16(x2, transpose=True, row_pad=8, out=xt[:k])
                    xt[k:].zero_()
                    if has_bias:
                        xt[k8, :m] = 1.0
                dwb = ops.gemm_bf16(dzt, xt, out_dtype=torch.float32, splits=_wgrad_splits(n, k, m))
                if ctx.needs_input_grad[1]:
                    dw = dwb[:n, :k]
                if has_bias and ctx.needs_input_grad[2]:
                    db = dwb[:n, k8]
        else:
            dz32 = dz.float()
            if ctx.needs_input_grad[0]:
                wt = cache.get(weight, "x3_t", lambda: ops.split_bf16x3(_pad_rows8(_pad8(weight.detach().float().t().contiguous())).contiguous(), 1))
                dx = ops.gemm_bf16(ops.split_bf16x3(_pad8(dz32).contiguous(), 0), wt, out_dtype=torch.float32)[:, :k]
            if ctx.needs_input_grad[1]:
                dzt = _pad_rows8(_pad8(dz32.t().contiguous())).contiguous()              # (N, Mp)
                xt = _pad_rows8(_pad8(x2.float().t().contiguous())).contiguous()         # (K, Mp)
                dw = ops.gemm_bf16(ops.split_bf16x3(dzt, 0), ops.split_bf16x3(xt, 1), out_dtype=torch.float32,
                                   splits=_wgrad_splits(n, k, 3 * m))[:n, :k]
            if has_bias and ctx.needs_input_grad[2]:
                db = ops.rowsum(dz32.t().contiguous().unsqueeze(0))
        if dx is not None:
            dx = dx.to(xdt).reshape(shp)
        if dw is not None:
            dw = dw.to(wdt)
        if db is not None:
            db = db.to(bdt)
        return dx, dw, db, dres, None, None, None, None


def _lowp(x, weight):
    """bf16 MFMA path? (bf16 tensors, or bf16 autocast over fp32 master weights)"""
    if x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16:
        return True
    dev = "cuda" if x.is_cuda else "cpu"
    if torch.is_autocast_enabled(dev):
        if torch.get_autocast_dtype(dev) != torch.bfloat16:
            raise NotImplementedError("only bf16 autocast is on the HIP path")
        return True
    if x.dtype != weight.dtype:
        raise TypeError(f"Linear: input {x.dtype} vs weight {weight.dtype} outside autocast")
    if x.dtype != torch.float32:
        raise TypeError(f"Linear: unsupported dtype {x.dtype}")
    return False


class Linear(nn.Module):
    """Drop-in for nn.Linear (same parameters, same init) running on csrc/gemm.hip."""

    def __init__(self, in_features, out_features, bias=True, device=None, dtype=None):
        super().__init__()
        ref = nn.Linear(in_features, out_features, bias=bias, device=device, dtype=dtype)   # the reference's init
        self.in_features, self.out_features = in_features, out_features
        self.weight = ref.weight
        self.bias = ref.bias
        self._cache = _WeightCache()
        self.fp8 = False          # forward products in fp8 e4m3 (set_fp8): BASELINE.json configs[4]; the backward stays bf16

    def extra_repr(self):
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}"

    def lowp_weight(self):
        """bf16 copy of the weight (cached), or None when the fused projection epilogues do not apply (K % 8 != 0)."""
        if self.in_features % 8:
            return None
        w = self.weight
        return self._cache.get(w, "bf16", lambda: _pad_rows8(w.detach() if w.dtype == torch.bfloat16 else _ops().cast_bf16(w.detach())))

    def forward(self, x, res=None, mode=None):
        """mode None: x W^T + b [+ res];  'swiglu': value * silu(gate) over W rows = [value | gate]."""
        if mode is None:
            mode = "res" if res is not None else "plain"
        if isinstance(x, Fp8Rows):
            return self._forward_fp8_rows(x, res, mode)
        return LinearFn.apply(x, self.weight, self.bias, res, mode, _lowp(x, self.weight), self._cache, self.fp8)

    def _forward_fp8_rows(self, xq, res, mode):
        """Inference on an input that arrives quantised (Fp8Rows): the fp8 branch of LinearFn.forward without the quantisation pass."""
        if torch.is_grad_enabled() and (self.weight.requires_grad or (res is not None and res.requires_grad)):
            raise RuntimeError("a pre-quantised input is an inference-only path")
        ops = _ops()
        b, sb = self.fp8_weight()
        n = self.weight.shape[0]
        bias32 = None
        if self.bias is not None:
            bias32 = self._cache.get(self.bias, "bias32", lambda: torch.nn.functional.pad(self.bias.detach().float(), (0, (-n) % 8)).contiguous())
        nout = n // 2 if mode == "swiglu" else n
        if b.shape[0] != n or (mode == "swiglu" and n % 16):
            raise ValueError("pre-quantised input: the projection needs out_features % 8 == 0 (SwiGLU: % 16)")
        y = torch.empty(*xq.shape[:-1], nout, dtype=torch.bfloat16, device=xq.device)
        r2 = None
        if mode == "res":
            r2 = res.reshape(-1, n).to(torch.bfloat16)
            if r2.stride(1) != 1:
                r2 = r2.contiguous()
        epi = ops.EPI_SWIGLU if mode == "swiglu" else (ops.EPI_RES if mode == "res" else ops.EPI_STORE)
        alpha, calpha = fp8_scales(None, sb)
        ops.gemm_fp8(xq.q, b, alpha, bias=bias32, res=r2, epilogue=epi, out_dtype=torch.bfloat16, out=y.view(-1, nout), row_alpha=xq.scale,
                     col_alpha=calpha)
        return y

    def fp8_weight(self):
        """(uint8 e4m3 weight, dequant scale: per output channel (rows,) or 0-dim per tensor — fp8_scales) cached per weight version, or
        None when K % 16 != 0."""
        if self.in_features % 16:
            return None
        w = self.weight
        return self._cache.get(w, "fp8", lambda: _quant_weight_fp8(_ops(), w))


def set_fp8(module, enabled=True, min_features=256, policy="all"):
    """Switch the forward GEMMs of the native Linears under `module` with in/out features >= min_features to fp8 e4m3 (dynamic scales per
    activation row and per weight output channel, MX MFMA at twice the bf16 rate).  The tiny conditioning MLPs stay bf16.
    policy "all":    every such projection (BASELINE.json configs[4] as timed: 7 GEMMs per layer);
    policy "attn":   the attention projections only (to_qkv, to_out, to_q, to_kv: 39 % of a layer's projection flops), the feed-forward
                     pair in bf16 — the accuracy-first setting.  e4m3's 3-bit mantissa on both operands costs each GEMM ~4 % relative L2
                     whatever the scale granularity (a dot product of zero-mean terms does not average the terms' relative rounding error
                     away), the roundings of different GEMMs add in quadrature, and the feed-forward pair carries most of it (its input
                     projection feeds v * silu(g), a product of two rounded values, into a K = 6144 reduction): depth 24, N = 6145,
                     final output vs fp32 — all 0.135, feed-forward pair alone 0.130, attention projections alone 0.043, bf16 0.021
                     (profiles/r06_experiments/fp8_policy/);
    policy "no_ff1": everything but the feed-forward INPUT projection (59 % of the flops; 0.043 and FF2's 0.078 in quadrature);
    policy "ff":     the feed-forward pair only (the experiment's other arm).
    Returns the number of Linears switched."""
    if policy not in ("all", "attn", "no_ff1", "ff"):
        raise ValueError("set_fp8: policy is 'all', 'attn', 'no_ff1' or 'ff'")
    n = 0
    for name, m in module.named_modules():
        if isinstance(m, Linear) and min(m.in_features, m.out_features) >= min_features and m.in_features % 16 == 0:
            dotted = "." + name + "."
            in_ff, ff1 = ".ff." in dotted, ".ff.ff.0." in dotted
            on = bool(enabled) and {"all": True, "attn": not in_ff, "no_ff1": not ff1, "ff": in_ff}[policy]
            m.fp8 = on
            n += int(on) if enabled else 1
    return n
