"""MS-STFT discriminator of the autoencoder training step on the native kernels.

Mirror of stable_audio_tools/models/encodec.py (NormConv2d :19-28, get_2d_padding :34-35, DiscriminatorSTFT :37-106,
MultiScaleSTFTDiscriminator :108-138) and stable_audio_tools/models/discriminators.py (get_hinge_losses :13-16,
EncodecDiscriminator :18-63): same class names, constructor kwargs and state_dict keys
(`discriminators.discriminators.{i}.convs.{j}.conv.{weight_g,weight_v,bias}`, `...conv_post.conv.*`).

Execution
  * spectrogram: csrc/stft.hip `sat_spec_fwd/bwd` — LDS FFT per frame (both stereo channels in one complex transform), normalised
    by ||w||, real/imag planes written directly in the (B, 2C, frames, freq) layout the convs read; backward without atomics.
    (torchaudio's Spectrogram is not in the reference tree: restated from its published algorithm, torch.stft(center=False)/||w||.)
  * Conv2d (3 x 9 / 3 x 3 kernels, dilation along frames, stride 1 — DiscriminatorSTFT's default, the only one the reference
    configures), round 3: csrc/disc_conv.hip on the "pitched rows" layout — every activation of a scale stays (B, C, frames * P)
    (frame = [4 zeros | freq bins | zeros]) from the spectrogram to the logits, the frame taps are virtual channels read from the
    same buffer (no shifted copies, no un-pitching between layers), each layer's epilogue applies bias + LeakyReLU + the pad mask and
    writes the next layer's bf16 hi / lo operand planes; data-gradient = the same kernel on transposed weights, weight-gradient =
    sat_disc_wgrad (_DiscConvFn).  The public forward returns strided (B, C, frames, freq) views of those buffers.
    Fallback (fp32-MFMA mode, other kernel shapes; round 2's path):
    run as stride-1 Conv1d over "virtual channels" on the conv stack's bf16x3 MFMA kernels (csrc/conv1d_bf16x3*.hip,
    conv_wgrad*): the kh frame taps become channels (time-shifted copies), and the (frames x freq) plane is laid out as one long
    sequence of zero-separated rows, so the 1-D kernels see the same regime as the VAE convs (C' = 192 channels, millions of
    steps) instead of thousands of short rows; a 9-tap kernel is taps 0..7 in one launch of the k7 kernels + tap 8 as a 1-tap conv on
    an offset view of the same buffer (_Conv9Fn: one autograd unit); 3-tap kernels are zero-padded to 7 taps in training (the
    pipelined 7-tap weight-gradient kernel).
    The rearrangement is two kernels each way (sat_rows_pack / sat_rows_unpack with LeakyReLU, and their adjoints).
  * weight norm: functional.WeightNormFn (sat_wn_fold / sat_wn_grad), as for the 1-D convs.
"""
import typing as tp

import torch
from torch import nn
from torch.nn import functional as F

from . import functional as _fn
from .functional import SnakeConv1dFn, WeightNormFn


def get_2d_padding(kernel_size, dilation=(1, 1)):
    return (((kernel_size[0] - 1) * dilation[0]) // 2, ((kernel_size[1] - 1) * dilation[1]) // 2)


class _SpecFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n_fft, hop):
        ops = _fn._ops(None)
        x = x.contiguous()
        ctx.meta = (ops, x.shape[1], x.shape[2], n_fft, hop)
        return ops.spec_fwd(x, n_fft, hop)

    @staticmethod
    def backward(ctx, dz):
        ops, c, t, n_fft, hop = ctx.meta
        return ops.spec_bwd(dz.contiguous(), c, t, n_fft, hop), None, None


class _PackRowsFn(torch.autograd.Function):
    """(B, C, T, W) -> the flat virtual-channel sequence buffer (ops.rows_pack); backward = its adjoint, one kernel each way."""

    @staticmethod
    def forward(ctx, x, kh, dil_t, pad_t, pad_w, pitch, lead):
        ops = _fn._ops(None)
        x = x.contiguous()
        ctx.meta = (ops, tuple(x.shape), kh, dil_t, pad_t, pad_w, pitch, lead)
        return ops.rows_pack(x, kh, dil_t, pad_t, pad_w, pitch, lead)

    @staticmethod
    def backward(ctx, dbuf):
        ops, shape, kh, dil_t, pad_t, pad_w, pitch, lead = ctx.meta
        return ops.rows_pack_bwd(dbuf.contiguous(), shape, kh, dil_t, pad_t, pad_w, pitch, lead), None, None, None, None, None, None


class _UnpackActFn(torch.autograd.Function):
    """conv output (B, C, T*pitch) -> LeakyReLU(slope) of its sample positions as (B, C, T, W); slope 1 = no activation."""

    @staticmethod
    def forward(ctx, y, t, wd, pad_w, pitch, slope):
        ops = _fn._ops(None)
        out = ops.rows_unpack(y.contiguous(), t, wd, pad_w, pitch, slope)
        ctx.meta = (ops, pad_w, pitch, slope)
        if slope != 1.0:
            ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        ops, pad_w, pitch, slope = ctx.meta
        out = ctx.saved_tensors[0] if slope != 1.0 else None
        return ops.rows_unpack_bwd(dout.contiguous(), out, pad_w, pitch, slope), None, None, None, None, None


class _Conv9Fn(torch.autograd.Function):
    """The 9-tap row conv on the packed sequence buffer as ONE autograd unit: taps 0..7 as an 8-tap conv on the sequence (storage offset
    4) + tap 8 as a 1-tap conv on the view at offset 8, chained through the residual input.  The backward writes both data-gradients
    into ONE buffer of the packed layout — the 8-tap one with plain stores at offset 4, the 1-tap one accumulated in place at offset 8
    (out = res) — so no autograd of offset views (zero-fill + copy + add per view) is left."""

    @staticmethod
    def forward(ctx, buf, w1, bias, b, cp, L):
        ops = _fn._ops(None)
        strides = (cp * L, L, 1)
        v0, v4, v8 = (buf.as_strided((b, cp, L), strides, o) for o in (0, 4, 8))
        w1 = w1.contiguous()
        # forward and data-gradient: taps 0..7 in ONE launch (the k7 kernels process 8 tap groups per chunk — the 8th is a zero pad
        # for a 7-tap conv) + tap 8 as a 1-tap conv on the view at offset 8; the weight-gradient keeps the 7 + 1 + 1 split (its
        # pipelined kernel is 7-tap)
        y = _fn._conv_fwd(ops, v4, w1[..., 0:8].contiguous(), 1, 1, 4, bias=bias, tout=L)
        y = _fn._conv_fwd(ops, v8, w1[..., 8:9].contiguous(), 1, 1, 0, res=y)
        ctx.meta = (ops, b, cp, L, bias is not None)
        ctx.save_for_backward(buf, w1)
        return y

    @staticmethod
    def backward(ctx, dy):
        ops, b, cp, L, has_bias = ctx.meta
        buf, w1 = ctx.saved_tensors
        dy = dy.contiguous()
        strides = (cp * L, L, 1)
        v0, v4, v8 = (buf.as_strided((b, cp, L), strides, o) for o in (0, 4, 8))
        dw = dbias = None
        if ctx.needs_input_grad[1]:
            dw7 = _fn._conv_wgrad(ops, dy, v4, 7, 1, 1, 3, None, bias_grad=has_bias)
            if has_bias:
                dw7, dbias = dw7
            dw0 = _fn._conv_wgrad(ops, dy, v0, 1, 1, 1, 0, None)
            dw8 = _fn._conv_wgrad(ops, dy, v8, 1, 1, 1, 0, None)
            dw = torch.cat([dw0, dw7, dw8], dim=2)
        elif has_bias:
            dbias = ops.rowsum(dy)
        dbuf = None
        if ctx.needs_input_grad[0]:
            dbuf = torch.empty_like(buf)
            dbuf[:4].zero_()                           # the slack before / after the sequence: read by the in-place accumulations
            dbuf[-4:].zero_()
            d4, d8 = (dbuf.as_strided((b, cp, L), strides, o) for o in (4, 8))
            _fn._conv_dgrad(ops, dy, w1[..., 0:8].contiguous(), 8, 1, 1, 4, cp, L, None, out=d4)
            _fn._conv_dgrad(ops, dy, w1[..., 8:9].contiguous(), 1, 1, 1, 0, cp, L, None, res=d8, out=d8)
        return dbuf, dw, dbias, None, None, None


class _PitchFn(torch.autograd.Function):
    """(B, C, frames, W) -> the pitched sequence (B, C, frames * P) (csrc/disc_conv.hip layout) + its operand planes for the first
    conv; backward = un-pitching."""

    @staticmethod
    def forward(ctx, z):
        ops = _fn._ops(None)
        z = z.contiguous()
        frames, wd = z.shape[2], z.shape[3]
        h, pl = ops.disc_planes(z, frames, wd, want_dst=True, slot=0)
        ops.disc_register(h, pl, z.shape[1], frames, wd, 0)
        ctx.meta = (ops, frames, wd)
        return h

    @staticmethod
    def backward(ctx, g):
        ops, frames, wd = ctx.meta
        return ops.rows_unpack(g.contiguous(), frames, wd, 4, ops.disc_geom(frames, wd)[0], 1.0)


class _DiscConvFn(torch.autograd.Function):
    """One NormConv2d + LeakyReLU of DiscriminatorSTFT on the pitched layout: h (B, Cin, L) -> y = LeakyReLU_slope(conv2d(h, w) + bias)
    (B, Cout, L), pad positions zero.  Forward: sat_disc_conv on the planes the producer emitted for h (or sat_disc_planes), emitting
    the planes of its own output unless `last`.  Backward: dpre = g * LeakyReLU'(y) as pitched fp32 + planes in one pass
    (sat_disc_planes), dW = sat_disc_wgrad(dpre, h), dbias = row sums, dh = sat_disc_conv(planes of dpre, transposed weights).

    fm_ref (pitched, no gradient): the layer also returns sum |y - fm_ref| (the L1 feature-matching distance to the other signal's
    feature map, models/discriminators.py:52-56); its gradient enters the same dpre pass (+ g_fm * sign(y - fm_ref) before the
    LeakyReLU' factor) — no elementwise autograd graph over the feature maps.
    Chained gradients (a chain of layers whose feature maps nothing else consumes — the discriminator update): fold_slope = the slope of
    the layer that produced h: the data-gradient leaves as THAT layer's dL/d(pre-activation) (fp32 + planes, sat_disc_conv's lk_src);
    dpre_in = the gradient this layer receives is already w.r.t. its pre-activation.  One pass over each gradient less per layer."""

    @staticmethod
    def forward(ctx, h, w4, bias, frames, wd, dil_t, slope, last, fm_ref, fold_slope, dpre_in):
        ops = _fn._ops(None)
        h = h.contiguous()
        w4 = w4.contiguous()
        b, cin, _ = h.shape
        cout, _, kh, kw = w4.shape
        pl, slot = ops.disc_take(h, cin, frames, wd)
        y, em = ops.disc_conv(pl, ops.disc_pack(w4, 0), bias, b, cin, cout, frames, wd, kh, kw, dil_t, slope,
                              emit_slot=None if last else 1 - slot)
        if em is not None:
            ops.disc_register(y, em, cout, frames, wd, 1 - slot)
        if fm_ref is not None and (dpre_in or slope == 1.0):
            raise ValueError("_DiscConvFn: the feature-matching term needs an activated layer with the ordinary gradient contract")
        ctx.meta = (ops, frames, wd, dil_t, slope, bias is not None, fold_slope, dpre_in)
        fm_sum = fm_sign = None
        if fm_ref is not None:
            # the distance now, and sign(y - ref) as int8 for the backward: the other signal's feature map need not stay alive
            fm_sum, fm_sign = ops.disc_l1_sum(y, fm_ref.contiguous(), want_sign=True)
        ctx.save_for_backward(h, w4, y if (slope != 1.0 and not dpre_in) else None, fm_sign)
        if fm_ref is None:
            return y
        return y, fm_sum

    @staticmethod
    def backward(ctx, g, g_fm=None):
        ops, frames, wd, dil_t, slope, has_bias, fold_slope, dpre_in = ctx.meta
        h, w4, y, fm_sign = ctx.saved_tensors
        cout, cin, kh, kw = w4.shape
        b = h.shape[0]
        need_h, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if g is None:
            g = torch.zeros(b, cout, h.shape[2], dtype=torch.float32, device=h.device)
        g = g.contiguous()
        if dpre_in:
            dpre = g
            dpl, slot = ops.disc_take(dpre, cout, frames, wd) if need_h else (None, 0)
        else:
            fm = fm_sign is not None and g_fm is not None
            dpre, dpl = ops.disc_planes(g, frames, wd, out=y, slope=slope, want_dst=True, want_planes=need_h, slot=0,
                                        fm_sign=fm_sign if fm else None, fm_coef=g_fm.contiguous() if fm else None)
            slot = 0
        dw = ops.disc_wgrad(dpre, h, frames, wd, kh, kw, dil_t) if need_w else None
        dbias = ops.rowsum(dpre) if (has_bias and ctx.needs_input_grad[2]) else None
        dh = None
        if need_h:
            fold = fold_slope is not None
            dh, em = ops.disc_conv(dpl, ops.disc_pack(w4, 1), None, b, cout, cin, frames, wd, kh, kw, dil_t, 1.0,
                                   emit_slot=(1 - slot) if fold else None, lk_src=h if fold else None, lk_slope=fold_slope if fold else 1.0)
            if fold:
                ops.disc_register(dh, em, cin, frames, wd, 1 - slot)
        return (dh, dw, dbias) + (None,) * 8


def _pitched_ok(ops, kernel_size, dilation):
    """May this Conv2d run on csrc/disc_conv.hip?  (bf16x3 mode; kw 9 or 3 — the weight-gradient kernel's tap counts; frame taps within
    the planes' lead rows)"""
    kh, kw = kernel_size
    return bool(ops.use_bf16x3 and getattr(ops, "disc_pitched", True) and kw in (3, 9) and kh % 2 == 1 and dilation[1] == 1
                and dilation[0] * (kh - 1) // 2 <= 4)


def conv2d_virtual(x, w, bias, dil_t=1, pad_t=0, split_wide=True, slope=1.0):
    """leaky_relu(F.conv2d(x, w, bias, stride=1, dilation=(dil_t, 1), padding=(pad_t, (kw-1)//2)), slope) on the 1-D conv kernels
    ("same" along the frequency axis, as get_2d_padding gives; slope 1 = no activation).  x (B, Cin, T, W); w (Cout, Cin, kh, kw), kw odd.

    Layout: the kh frame taps become channels (time-shifted copies: C' = Cin*kh), and the (T x W) plane becomes one sequence of rows
    [pad zeros | W samples | pad zeros] of pitch W + kw - 1 — a "same" 1-D conv of that sequence never mixes two rows' samples, and
    its outputs at the sample positions are the 2-D conv's.  A 9-tap kernel runs as taps 0..7 (one launch of the k7 kernels, which
    process 8 tap groups per chunk) + tap 8 (a 1-tap conv on the sequence shifted by +4 samples — an offset view of the same buffer —
    added through the residual input of its launch); its weight gradient as taps 1..7 (the pipelined 7-tap kernel) + tap 0 + tap 8
    (_Conv9Fn).
    The sequence tensor is built by ONE kernel (sat_rows_pack; adjoint sat_rows_pack_bwd) and the output rows are un-pitched and
    activated by one (sat_rows_unpack / _bwd): no torch pad / stack / slice / LeakyReLU passes and none of their autograd adds and fills."""
    b, cin, t, wd = x.shape
    cout, _, kh, kw = w.shape
    if kw % 2 != 1:
        raise NotImplementedError("conv2d_virtual: odd frequency kernel")
    if pad_t * 2 != dil_t * (kh - 1):
        raise NotImplementedError("conv2d_virtual: 'same' padding along frames (get_2d_padding)")
    pad_w = (kw - 1) // 2
    pitch = (wd + 2 * pad_w + 3) // 4 * 4          # multiple of 4: 16-byte epilogues and the pipelined weight-gradient kernel
    cp = cin * kh
    L = t * pitch
    w1 = w.reshape(cout, cp, kw)
    if kw == 9 and split_wide:
        # the sequence sits in a buffer with 4 elements of slack on both sides, so that "the sequence shifted by +-4 samples" is a
        # VIEW (same strides, storage offset -+4): the outer taps become 1-tap convs on those views (the k1 kernels), chained
        # through the residual input — no dilated conv, no shifted copies.  Positions where a shifted view reads across a channel
        # boundary are row-padding positions, whose outputs are discarded (and carry zero gradient).
        buf = _PackRowsFn.apply(x, kh, dil_t, pad_t, pad_w, pitch, 4)
        y = _Conv9Fn.apply(buf, w1, bias, b, cp, L)
    else:
        seq = _PackRowsFn.apply(x, kh, dil_t, pad_t, pad_w, pitch, 0).view(b, cp, L)
        if kw < 7 and split_wide and cp >= 64 and torch.is_grad_enabled():
            # training: zero-pad the 3-tap kernel to 7 taps — the k7 kernels process 8 tap groups per chunk either way, and the 7-tap
            # weight-gradient kernel (bf16x3, pipelined) is ~4x faster than the generic one a 3-tap conv would fall back to
            ext = (7 - kw) // 2
            y = SnakeConv1dFn.apply(seq, None, None, F.pad(w1, (ext, ext)).contiguous(), bias, None, 1, 1, 3, False)
        else:
            y = SnakeConv1dFn.apply(seq, None, None, w1.contiguous(), bias, None, 1, 1, pad_w, False)
    return _UnpackActFn.apply(y, t, wd, pad_w, pitch, float(slope))


class _WNConv2d(nn.Module):
    """weight_norm(nn.Conv2d) parameters (old-style names weight_g / weight_v / bias, models/encodec.py:25) + the virtual-channel run."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=(1, 1), dilation=(1, 1), padding=(0, 0)):
        super().__init__()
        ref = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, dilation=dilation, padding=padding)
        self.weight_v = nn.Parameter(ref.weight.detach().clone())
        self.weight_g = nn.Parameter(ref.weight.detach().flatten(1).norm(dim=1).view(-1, 1, 1, 1).clone())
        self.bias = nn.Parameter(ref.bias.detach().clone())
        self.kernel_size, self.stride, self.dilation, self.padding = tuple(kernel_size), tuple(stride), tuple(dilation), tuple(padding)
        if self.stride != (1, 1) or self.dilation[1] != 1 or self.padding != get_2d_padding(self.kernel_size, self.dilation):
            raise NotImplementedError("only stride (1, 1), frequency dilation 1 and 'same' padding (the MS-STFT discriminator as "
                                      "stable-audio-tools configures it: DiscriminatorSTFT(stride=(1, 1)), models/encodec.py:58)")

    def forward(self, x, slope=1.0):
        """slope != 1: LeakyReLU(slope) of the conv output, fused into the pass that un-pitches it."""
        w = WeightNormFn.apply(self.weight_v, self.weight_g)
        return conv2d_virtual(x, w, self.bias, dil_t=self.dilation[0], pad_t=self.padding[0], slope=slope)

    def pitched_ok(self):
        return _pitched_ok(_fn._ops(None), self.kernel_size, self.dilation)

    def forward_pitched(self, h, frames, wd, slope=1.0, last=False, fm_ref=None, fold_slope=None, dpre_in=False):
        """The same layer on the pitched sequence (B, Cin, frames * P) -> (B, Cout, frames * P) (_DiscConvFn; with fm_ref: (y, sum |y - fm_ref|))."""
        w = WeightNormFn.apply(self.weight_v, self.weight_g)
        return _DiscConvFn.apply(h, w, self.bias, frames, wd, self.dilation[0], float(slope), last, fm_ref, fold_slope, dpre_in)


class NormConv2d(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.conv = _WNConv2d(*args, **kwargs)

    def forward(self, x, slope=1.0):
        return self.conv(x, slope)


class DiscriminatorSTFT(nn.Module):
    def __init__(self, filters, in_channels=1, out_channels=1, n_fft=1024, hop_length=256, win_length=1024, max_filters=1024,
                 filters_scale=1, kernel_size=(3, 9), dilations=[1, 2, 4], stride=(1, 1), normalized=True, activation="LeakyReLU",
                 activation_params={"negative_slope": 0.2}, spec_scale_pow=0.0, **kwargs):
        super().__init__()
        if win_length != n_fft or not normalized or spec_scale_pow != 0.0 or in_channels not in (1, 2):
            raise NotImplementedError("DiscriminatorSTFT: win_length == n_fft, normalized, spec_scale_pow == 0, mono/stereo only")
        self.filters, self.in_channels, self.out_channels = filters, in_channels, out_channels
        self.n_fft, self.hop_length, self.win_length = n_fft, hop_length, win_length
        self.activation = getattr(torch.nn, activation)(**activation_params)
        spec_channels = 2 * in_channels
        self.convs = nn.ModuleList()
        self.convs.append(NormConv2d(spec_channels, filters, kernel_size=kernel_size, padding=get_2d_padding(kernel_size)))
        in_chs = min(filters_scale * filters, max_filters)
        for i, dilation in enumerate(dilations):
            out_chs = min((filters_scale ** (i + 1)) * filters, max_filters)
            self.convs.append(NormConv2d(in_chs, out_chs, kernel_size=kernel_size, stride=stride, dilation=(dilation, 1),
                                         padding=get_2d_padding(kernel_size, (dilation, 1))))
            in_chs = out_chs
        out_chs = min((filters_scale ** (len(dilations) + 1)) * filters, max_filters)
        k0 = (kernel_size[0], kernel_size[0])
        self.convs.append(NormConv2d(in_chs, out_chs, kernel_size=k0, padding=get_2d_padding(k0)))
        self.conv_post = NormConv2d(out_chs, out_channels, kernel_size=k0, padding=get_2d_padding(k0))

    def pitched_ok(self):
        return isinstance(self.activation, torch.nn.LeakyReLU) and all(l.conv.pitched_ok() for l in list(self.convs) + [self.conv_post])

    def forward_pitched(self, x, fm_refs=None, exclusive=False, release_refs=False):
        """(logits, feature maps, (frames, freq bins), fm sums) with every tensor in the pitched layout (B, C, frames * P), pad positions
        zero.  fm_refs: the other signal's feature maps (no gradient) -> fm sums[i] = sum |fmap[i] - fm_refs[i]|, differentiable w.r.t.
        this signal's path (else None); release_refs: set fm_refs[i] = None as layer i has consumed it (the layers keep sign(y - ref) as
        int8 for the backward, not the reference map).  exclusive: the caller consumes ONLY the logits — the layers chain their gradients
        (_DiscConvFn fold_slope / dpre_in); gradients w.r.t. the returned feature maps would be wrong, so they are detached."""
        if fm_refs is not None and exclusive:
            raise ValueError("forward_pitched: feature-matching terms need the feature maps' ordinary gradients")
        z = _SpecFn.apply(x, self.n_fft, self.hop_length)
        frames, wd = z.shape[2], z.shape[3]
        h = _PitchFn.apply(z)
        slope = self.activation.negative_slope
        fmap, sums = [], []
        for i, layer in enumerate(self.convs):
            out = layer.conv.forward_pitched(h, frames, wd, slope, fm_ref=None if fm_refs is None else fm_refs[i],
                                             fold_slope=slope if (exclusive and i > 0) else None, dpre_in=exclusive)
            if fm_refs is not None and release_refs:
                fm_refs[i] = None                 # consumed (the layer kept sign(y - ref) only): the caller's list drops its reference
            if fm_refs is not None:
                h, s_i = out
                sums.append(s_i)
            else:
                h = out
            fmap.append(h.detach() if exclusive else h)
        logit = self.conv_post.conv.forward_pitched(h, frames, wd, 1.0, last=True, fold_slope=slope if exclusive else None)
        return logit, fmap, (frames, wd), (sums if fm_refs is not None else None)

    def forward(self, x):
        if self.pitched_ok():
            logit, fmap, (frames, wd), _ = self.forward_pitched(x)
            unp = lambda t: _unpitch_view(t, frames, wd)           # noqa: E731
            return unp(logit), [unp(f) for f in fmap]
        fmap = []
        z = _SpecFn.apply(x, self.n_fft, self.hop_length)          # (B, 2C, frames, freq) = cat(real, imag) + 'b c w t -> b c t w'
        leaky = isinstance(self.activation, torch.nn.LeakyReLU)
        for layer in self.convs:
            # z = activation(layer(z))  (models/encodec.py:101-103); LeakyReLU rides in the conv's output pass
            z = layer(z, self.activation.negative_slope) if leaky else self.activation(layer(z))
            fmap.append(z)
        return self.conv_post(z), fmap


def _unpitch_view(t, frames, wd):
    """pitched (B, C, frames * P) -> the (B, C, frames, wd) strided view of its sample positions (no copy)."""
    b, c, L = t.shape
    return t.view(b, c, frames, L // frames)[..., 4:4 + wd]


class MultiScaleSTFTDiscriminator(nn.Module):
    def __init__(self, filters, in_channels=1, out_channels=1, n_ffts=[1024, 2048, 512], hop_lengths=[256, 512, 128],
                 win_lengths=[1024, 2048, 512], **kwargs):
        super().__init__()
        assert len(n_ffts) == len(hop_lengths) == len(win_lengths)
        self.discriminators = nn.ModuleList([
            DiscriminatorSTFT(filters, in_channels=in_channels, out_channels=out_channels, n_fft=n_ffts[i], win_length=win_lengths[i],
                              hop_length=hop_lengths[i], **kwargs) for i in range(len(n_ffts))])
        self.num_discriminators = len(self.discriminators)

    def forward(self, x):
        logits, fmaps = [], []
        for disc in self.discriminators:
            logit, fmap = disc(x)
            logits.append(logit)
            fmaps.append(fmap)
        return logits, fmaps


def get_hinge_losses(score_real, score_fake):
    """models/discriminators.py:13-16.  The means over the (millions of) logits of a 47 s item run on functional.mean_all
    (ops.sum_all: why not torch's .mean())."""
    gen_loss = -_fn.mean_all(score_fake)
    dis_loss = _fn.mean_all(torch.relu(1 - score_real)) + _fn.mean_all(torch.relu(1 + score_fake))
    return dis_loss, gen_loss


class EncodecDiscriminator(nn.Module):
    def __init__(self, normalize_losses=False, loss_type: tp.Literal["hinge", "rpgan"] = "hinge", *args, **kwargs):
        super().__init__()
        if loss_type != "hinge":
            raise NotImplementedError("only the hinge loss (the configured default) is restated")
        self.discriminators = MultiScaleSTFTDiscriminator(*args, **kwargs)
        self.normalize_losses = normalize_losses
        self.fm_reduction = ((lambda x, y: _fn.mean_all(abs(x - y)) / (_fn.mean_all(abs(x)) + 1e-3)) if normalize_losses
                             else (lambda x, y: _fn.mean_all(abs(x - y))))
        self.loss_type = loss_type

    def forward(self, x):
        return self.discriminators(x)

    def scale_losses(self, i, reals, fakes, need_fm=True):
        """The terms scale i contributes to loss(): (dis_i, adv_i, fm_i), each already divided by the number of scales (need_fm=False:
        fm_i = 0 without computing it — the discriminator update only uses dis_i).  The training
        step back-propagates scale by scale (one scale's activations alive at a time: a 47 s stereo item makes ~23 GB of them per
        scale and branch) instead of holding all five graphs."""
        d = self.discriminators.discriminators[i]
        n = self.discriminators.num_discriminators
        if d.pitched_ok():
            # feature matching on the pitched buffers themselves (pad positions are zero in both: sums over the whole buffer / the
            # number of samples = the reference's means); only the one-channel logits are viewed un-pitched
            #   need_fm=False: only the logits are consumed -> the layers chain their gradients (exclusive);
            #   real feature maps without gradient (the generator update: frozen discriminator, real signal): the distances and their
            #   gradients ride in the fake path's layers (fm_refs); otherwise the distances are torch ops on the pitched buffers.
            logit_t, feat_t, (frames, wd), _ = d.forward_pitched(reals, exclusive=not need_fm)
            fused = need_fm and not any(f.requires_grad for f in feat_t)
            if fused:
                # what the distances need of the real maps besides the maps themselves: element counts (and mean |x| when normalising);
                # the maps are released one by one as the fake path's layers consume them (they keep sign(y - x) as int8)
                counts = [f.shape[0] * f.shape[1] * frames * wd for f in feat_t]
                norms = [_fn.sum_all(f.abs()) / c + 1e-3 for f, c in zip(feat_t, counts)] if self.normalize_losses else None
            logit_f, feat_f, _, sums = d.forward_pitched(fakes, fm_refs=feat_t if fused else None, exclusive=not need_fm, release_refs=fused)
            if not need_fm:
                fm = 0.0
            elif fused:
                fm = sum((s_i / c) / (norms[j] if norms is not None else 1.0) for j, (s_i, c) in enumerate(zip(sums, counts))) / len(counts)
            else:
                fm = sum(self._fm_pitched(a, b, frames, wd) for a, b in zip(feat_t, feat_f)) / len(feat_t)
            dis, adv = get_hinge_losses(_unpitch_view(logit_t, frames, wd), _unpitch_view(logit_f, frames, wd))
            return dis / n, adv / n, fm / n
        logit_t, feat_t = d(reals)
        logit_f, feat_f = d(fakes)
        fm = sum(map(self.fm_reduction, feat_t, feat_f)) / len(feat_t) if need_fm else 0.0
        dis, adv = get_hinge_losses(logit_t, logit_f)
        return dis / n, adv / n, fm / n

    def _fm_pitched(self, x, y, frames, wd):
        count = x.shape[0] * x.shape[1] * frames * wd
        d = _fn.sum_all((x - y).abs()) / count
        return d / (_fn.sum_all(x.abs()) / count + 1e-3) if self.normalize_losses else d

    def loss(self, reals, fakes):
        """(dis_loss, adv_loss, feature_matching_distance) / num_scales — models/discriminators.py:31-63."""
        if all(d.pitched_ok() for d in self.discriminators.discriminators):
            terms = [self.scale_losses(i, reals, fakes) for i in range(self.discriminators.num_discriminators)]
            return tuple(sum(t[k] for t in terms) for k in range(3))
        feature_matching_distance = torch.tensor(0., device=reals.device)
        dis_loss = torch.tensor(0., device=reals.device)
        adv_loss = torch.tensor(0., device=reals.device)
        logits_true, feature_true = self.forward(reals)
        logits_fake, feature_fake = self.forward(fakes)
        for i, (scale_true, scale_fake) in enumerate(zip(feature_true, feature_fake)):
            feature_matching_distance = feature_matching_distance + sum(map(self.fm_reduction, scale_true, scale_fake)) / len(scale_true)
            _dis, _adv = get_hinge_losses(logits_true[i], logits_fake[i])
            dis_loss = dis_loss + _dis
            adv_loss = adv_loss + _adv
        n = len(logits_true)
        return dis_loss / n, adv_loss / n, feature_matching_distance / n
