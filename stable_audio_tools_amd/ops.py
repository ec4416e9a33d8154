"""Python mirror of the C-ABI (include/sat_amd.h): torch tensors in, torch tensors out.

PyTorch is used here only as the device allocator / stream provider.  Every function hands raw
device pointers + shapes to the HIP library on ``torch.cuda.current_stream()``; the caller (the
PyTorch caching allocator) owns all buffers including workspaces.
"""
import ctypes

import torch

from . import _caches, _lib

PACK_CONV_FWD = 0      # w[d0][d1][K] -> [d1][k][d0]
PACK_CONV_DGRAD = 1    # w[d0][d1][K] -> [d0][K-1-k][d1]
PACK_POLYPHASE = 2     # w[d0][d1][2S] -> [r][j][d0][d1]


def _ptr(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def _keep_zeros(shape, dtype, device):
    """A workspace that outlives the call (plane pools, slabs, zero pages): allocated as an ORDINARY tensor even when the first use
    happens under torch.inference_mode — an inference tensor could not be sliced-and-zeroed by a later call outside that mode."""
    with torch.inference_mode(False):
        return torch.zeros(shape, dtype=dtype, device=device)


def _keep_empty(shape, dtype, device):
    with torch.inference_mode(False):
        return torch.empty(shape, dtype=dtype, device=device)


class WgradSlabs:
    """A weight gradient as its kernel left it: `nsplit` slabs of `count` floats, element (m, n, k) of slab z at
    partial[z, m * so_m + n * so_n + k * so_k] (tap-major [K][M][N] for the k = 7 kernels, torch order for the others).  reduce() sums
    them into the torch-layout tensor (M, N, K); a weight-normed conv hands the slabs to SatOps.wn_grad_splits instead."""
    __slots__ = ("partial", "nsplit", "dims", "strides")

    def __init__(self, partial, nsplit, dims, strides):
        self.partial, self.nsplit, self.dims, self.strides = partial, int(nsplit), tuple(int(d) for d in dims), tuple(int(x) for x in strides)

    @property
    def count(self):
        return self.partial.shape[1]

    def reduce(self, ops):
        m, n, k = self.dims
        part = self.partial if self.count == m * n * k else self.partial[:, :m * n * k].contiguous()      # (slabs longer than the gradient)
        flat = ops._reduce_rows(part, self.nsplit, m * n * k)
        if self.strides == (n * k, k, 1):
            return flat.view(m, n, k)
        if self.strides == (n, 1, m * n):
            return flat.view(k, m, n).permute(1, 2, 0).contiguous()
        raise ValueError("WgradSlabs.reduce: unknown slab layout %s" % (self.strides,))


class SatOps:
    def __init__(self, cdll):
        self.lib = cdll
        self.simulator = bool(cdll.sat_is_simulator())

    # ------------------------------------------------------------------ plumbing
    def _stream(self, t):
        if self.simulator:
            return None
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)

    def _chk(self, status):
        if status != 0:
            raise RuntimeError(self.lib.sat_last_error().decode())

    def _f32(self, *tensors):
        for t in tensors:
            if t is None:
                continue
            if t.dtype != torch.float32:
                raise TypeError(f"expected float32 tensor, got {t.dtype}")
            if not t.is_contiguous():
                raise ValueError("expected contiguous tensor")
            if not self.simulator and not t.is_cuda:
                raise RuntimeError("stable_audio_tools_amd kernels need CUDA(HIP) tensors; there is no CPU path")

    # ------------------------------------------------------------------ weight norm / packing
    def wn_fold(self, v, g):
        """w = g * v / ||v||  (norm over all dims but 0) -> (w like v, norm (D0,))"""
        self._f32(v, g)
        d0 = v.shape[0]
        r = v.numel() // d0
        w = torch.empty_like(v)
        norm = torch.empty(d0, dtype=torch.float32, device=v.device)
        self._chk(self.lib.sat_wn_fold(_ptr(v), _ptr(g), _ptr(w), _ptr(norm), d0, r, self._stream(v)))
        return w, norm

    def wn_grad(self, v, g, norm, dw):
        self._f32(v, g, norm, dw)
        d0 = v.shape[0]
        r = v.numel() // d0
        dv = torch.empty_like(v)
        dg = torch.empty_like(g)
        self._chk(self.lib.sat_wn_grad(_ptr(v), _ptr(g), _ptr(norm), _ptr(dw), _ptr(dv), _ptr(dg), d0, r, self._stream(v)))
        return dv, dg

    wn_fused = True     # weight norm inside the conv autograd units (functional._wn_forward); False: separate WeightNormFn nodes (A/B)

    def wn_grad_splits(self, slabs, v, g, norm, bias_partial=None):
        """(dv, dg) of w = g * v / ||v|| from a weight-gradient kernel's split slabs (WgradSlabs) in ONE launch: the sum over the slabs,
        the layout change and the weight-norm gradient (csrc/elementwise.hip sat_wn_grad_splits) — dW never exists in HBM.
        bias_partial (D0, R): per-split sums of dy (rowsum(..., partial=True) or a weight-gradient kernel's fused row sums); their sum,
        the conv's bias gradient, is then a third result of the same launch: (dv, dg, dbias)."""
        self._f32(v, g, norm, slabs.partial, bias_partial)
        m, n, k = slabs.dims
        if tuple(v.shape) != (m, n, k):
            raise ValueError("wn_grad_splits: slabs of a %s weight gradient for a %s weight" % ((m, n, k), tuple(v.shape)))
        dv = torch.empty_like(v)
        dg = torch.empty_like(g)
        dbias, cols = None, 0
        if bias_partial is not None:
            if bias_partial.dim() != 2 or bias_partial.shape[0] != m:
                raise ValueError("wn_grad_splits: bias_partial must be (D0, R)")
            cols = bias_partial.shape[1]
            dbias = torch.empty(m, dtype=torch.float32, device=v.device)
        so_m, so_n, so_k = slabs.strides
        self._chk(self.lib.sat_wn_grad_splits(_ptr(slabs.partial), slabs.nsplit, slabs.count, so_m, so_n, so_k, _ptr(v), _ptr(g), _ptr(norm),
                                              _ptr(dv), _ptr(dg), m, n, k, _ptr(bias_partial), cols, _ptr(dbias), self._stream(v)))
        return (dv, dg) if bias_partial is None else (dv, dg, dbias)

    def pack(self, w, mode, stride=1):
        self._f32(w)
        d0, d1, k = w.shape
        out = torch.empty(w.numel(), dtype=torch.float32, device=w.device)
        self._chk(self.lib.sat_pack_weights(_ptr(w), _ptr(out), d0, d1, k, stride, mode, self._stream(w)))
        return out

    # ------------------------------------------------------------------ convs
    def _reduce_rows(self, partial, rows, count, scale=1.0):
        out = torch.empty(count, dtype=torch.float32, device=partial.device)
        self._chk(self.lib.sat_reduce_splits(_ptr(partial), _ptr(out), count, rows, scale, 0, self._stream(partial)))
        return out

    def _sum_pair(self, pda, pdb):
        """The two (C, R) snake-gradient partial planes (halves of one (2, C, R) buffer) -> (C,), (C,) in one reduction."""
        c, r = pda.shape
        both = self.rowsum(pda._base.view(1, 2 * c, r) if pda._base is not None else torch.stack([pda, pdb]).view(1, 2 * c, r))
        return both[:c], both[c:]

    def _sum_last(self, partial):
        """(C, R) -> (C,): bandwidth-efficient reduction of per-tile partial sums (sat_rowsum)."""
        c, r = partial.shape
        return self.rowsum(partial.view(1, c, r))

    def conv1d(self, x, w_packed, cout, k, stride=1, dil=1, pad=0, tout=None, bias=None, snake=None, res=None,
               tanh_out=False, dsnake=None, out=None):
        """y = conv(snake(x)) [+bias] [+res] ; or, with dsnake=(x2, alpha2, beta2):
        y = conv(x) * dsnake(x2) + res and returns (y, dlog_alpha2, dlog_beta2)."""
        b, cin, tin = x.shape
        if tout is None:
            tout = (tin + 2 * pad - dil * (k - 1) - 1) // stride + 1
        alpha, beta = snake if snake is not None else (None, None)
        self._f32(x, w_packed, bias, alpha, beta, res)
        y = self._conv_out(out, b, cout, tout, x.device)
        x2 = a2 = b2 = pda = pdb = None
        rows = 0
        if dsnake is not None:
            x2, a2, b2 = dsnake
            self._f32(x2, a2, b2)
            rows = self.lib.sat_conv1d_partial_rows(b, tout)
            pda, pdb = torch.empty(2, cout, rows, dtype=torch.float32, device=x.device).unbind(0)
        self._chk(self.lib.sat_conv1d(_ptr(x), _ptr(w_packed), _ptr(bias), _ptr(alpha), _ptr(beta), _ptr(res), _ptr(y),
                                      _ptr(x2), _ptr(a2), _ptr(b2), _ptr(pda), _ptr(pdb),
                                      b, cin, cout, tin, tout, k, stride, dil, pad, int(tanh_out), self._stream(x)))
        if dsnake is not None:
            return (y, *self._sum_pair(pda, pdb))
        return y

    # -- bf16x3 split-MFMA path (csrc/conv1d_bf16x3.hip): stride-1 convs with K <= 8, and the K == 2*stride
    #    down / up (transposed) convs with a power-of-two stride --
    use_bf16x3 = True   # fp32-accurate (hi/lo split, 3 MFMAs per product); set False to force the fp32-MFMA kernels

    # the two-channel ends of the stack (csrc/edge_conv.hip): fp32 FMA streams instead of padded MFMA tiles; False (A/B, second
    # implementation in the tests) sends them to the matrix kernels like every other conv
    edge_convs = True

    def edge_ok(self, cin, cout, k, stride, dil, pad):
        return bool(self.edge_convs) and bool(self.lib.sat_edge_conv_ok(cin, cout, k, stride, dil, pad))

    def edge_conv(self, x, w, pad, mode=0, bias=None, snake=None, tanh_out=False, dsnake=None, emit=None):
        """conv1d(snake(x), W) + bias for a conv with <= 2 channels on one side (edge_ok).  w: the FOLDED torch weight — (Cout, Cin, K),
        or with mode=1 the (Cin', Cout', K) weight whose conv's data-gradient this is.  snake = (log-alpha, log-beta) of the input
        (Cout <= 2 form); dsnake = (x2, log-alpha2, log-beta2): returns (y * dsnake(x2), d log-alpha, d log-beta) (Cin <= 2 form)."""
        b, cin, t = x.shape
        k = w.shape[2]
        cout = w.shape[0] if mode == 0 else w.shape[1]
        alpha, beta = snake if snake is not None else (None, None)
        self._f32(x, w, bias, alpha, beta)
        y = torch.empty(b, cout, t, dtype=torch.float32, device=x.device)
        x2 = a2 = b2 = pda = pdb = None
        if dsnake is not None:
            x2, a2, b2 = dsnake
            self._f32(x2, a2, b2)
            rows = self.lib.sat_edge_conv_partial_rows(b, t)
            pda, pdb = torch.empty(2, cout, rows, dtype=torch.float32, device=x.device).unbind(0)
        ehi = elo = ela = elb = None
        erows = 0
        if emit is not None:      # emit = {"snake": (la, lb) | None}: act_next(y) as the planes of the k7 conv that reads y next (narrow-input form)
            esnake = emit.get("snake")
            if esnake is not None:
                ela, elb = esnake
                self._f32(ela, elb)
            ehi, elo, erows = self._emit_planes(b, cout, t, x.device, self._stream(x))
        self._chk(self.lib.sat_edge_conv(_ptr(x), _ptr(w), _ptr(bias), _ptr(alpha), _ptr(beta), _ptr(y), _ptr(x2), _ptr(a2), _ptr(b2),
                                         _ptr(pda), _ptr(pdb), _ptr(ehi), _ptr(elo), _ptr(ela), _ptr(elb), erows, b, cin, cout, t, k, pad, mode,
                                         int(tanh_out), self._stream(x)))
        if emit is not None:
            self._note_emitted(y, emit.get("snake"), ehi, elo, erows)
        if dsnake is not None:
            return (y, *self._sum_pair(pda, pdb))
        return y

    def edge_conv_wgrad(self, dy, x, k, pad, snake=None, dy_rowsum=False, raw=False):
        """dW (M, N, K) of conv1d(snake(x), W) with <= 2 channels on one side; contract of conv_wgrad7_bf16x3 (raw: WgradSlabs + the bias
        gradient's per-slab partial sums)."""
        b, m, t = dy.shape
        n = x.shape[1]
        alpha, beta = snake if snake is not None else (None, None)
        self._f32(dy, x, alpha, beta)
        nsplit = self.lib.sat_edge_conv_wgrad_nsplit(b, m, n, t)
        partial = torch.empty(nsplit, m * n * k, dtype=torch.float32, device=dy.device)
        fused = dy_rowsum and n <= 2                      # the kernel streams dy in that form and sums its rows on the way
        rs = torch.empty(m, nsplit, dtype=torch.float32, device=dy.device) if fused else None
        self._chk(self.lib.sat_edge_conv_wgrad(_ptr(dy), _ptr(x), _ptr(alpha), _ptr(beta), _ptr(partial), _ptr(rs), b, m, n, t, k, pad,
                                               self._stream(dy)))
        slabs = WgradSlabs(partial, nsplit, (m, n, k), (n * k, k, 1))
        dw = slabs if raw else slabs.reduce(self)
        if not dy_rowsum:
            return dw
        if raw:
            return dw, (rs if fused else self.rowsum(dy, partial=True))
        return dw, (self._sum_last(rs) if fused else self.rowsum(dy))

    def bf16x3_ok(self, k, stride, dil, transposed=False):
        if not self.use_bf16x3:
            return False
        if transposed or stride > 1:
            return k == 2 * stride and stride & (stride - 1) == 0 and dil == 1
        return k == 1 or (2 <= k <= 4 and dil == 1) or (5 <= k <= 8 and (k - 1) * dil <= 62)

    # the k7 kernel (csrc/conv1d_bf16x3_k7q.h): activation planes + 16-channel chunks + two wave rows one barrier apart.  Plain class
    # attributes (an A/B script sets them on the ops object; no environment switches: round 5)
    k7q = True
    k7q_persist = True      # one workgroup per CU walks the tiles (round 6: launches of >= 2 tiles per CU); False: one workgroup per tile
                            # (A/B: bench.py --ops-set k7q_persist=0); "force": persistent at any size (the tests' small shapes)
    k7q_dma_in_mfma = True  # VARIANT 3 of the k7q kernel (round 6): the next chunk's LDS-DMA issued between the MFMAs instead of beside the fragment
                            # reads (k7q 46.2 -> 44.9 ms per generator step; A/B: bench.py --ops-set k7q_dma_in_mfma=0)
    k7q_min_cin = 64
    k7q_min_cout = 128
    k7q_wide_cin = 512      # from this many input channels on, fewer than k7q_min_cout output channels still take the planes kernel

    def k7q_applicable(self, cin, k, stride, dil, pad, cout=None):
        """The ResidualUnit convs (C -> C, C >= 128) and their data-gradients.  Convs with fewer than k7q_min_cout output channels
        stay on the direct kernel: the MS-STFT discriminator's (64 filters: half of the 128-row channel tile would be empty, and the
        planes pre-pass is not paid back — measured, profiles/EXPERIMENTS.md) and the decoder's last conv (128 -> 2) — unless the
        INPUT is wide (the data-gradient of the decoder's first conv: 2048 -> 64 channels on 1024 steps is four workgroups whose
        time is the K loop over 2048 channels; round 2's k7p kernel served it until round 5)."""
        return (self.k7q and self.use_bf16x3 and stride == 1 and 5 <= k <= 7 and 0 <= pad <= 32
                and (k - 1) * dil <= 62 and cin >= self.k7q_min_cin
                and (cout is None or cout >= self.k7q_min_cout or cin >= self.k7q_wide_cin))

    def pack_bf16x3(self, w, mode=0, stride=1, q=False):
        """w: (D0, D1, K) fp32 -> (hi, lo) int16 planes.  mode 0: conv weight [out][in][K]; mode 1: data-gradient of a
        stride-1 conv; mode 2: transposed-conv weight [in][out][K].  q=True (stride 1, 5 <= K <= 7, modes 0 / 1): the layout of
        the k7q kernel (sat_pack_weights_k7q) — returned as (hi, lo, "q") so that conv1d_bf16x3 routes to it."""
        self._f32(w)
        d0, d1, k = w.shape
        if q:
            n = self.lib.sat_pack_weights_k7q_size(d0, d1, k, mode)
            if n <= 0 or stride != 1:
                raise RuntimeError("sat_pack_weights_k7q: unsupported shape")
            hi = torch.empty(n, dtype=torch.int16, device=w.device)
            lo = torch.empty(n, dtype=torch.int16, device=w.device)
            self._chk(self.lib.sat_pack_weights_k7q(_ptr(w), _ptr(hi), _ptr(lo), d0, d1, k, mode, self._stream(w)))
            return hi, lo, "q"
        n = self.lib.sat_pack_weights_bf16x3_size(d0, d1, k, stride, mode)
        if n <= 0:
            raise RuntimeError("sat_pack_weights_bf16x3: unsupported shape")
        hi = torch.empty(n, dtype=torch.int16, device=w.device)
        lo = torch.empty(n, dtype=torch.int16, device=w.device)
        self._chk(self.lib.sat_pack_weights_bf16x3(_ptr(w), _ptr(hi), _ptr(lo), d0, d1, k, stride, mode, self._stream(w)))
        return hi, lo

    def snake_consts(self, alpha, beta):
        self._f32(alpha, beta)
        a = torch.empty_like(alpha)
        ib = torch.empty_like(beta)
        self._chk(self.lib.sat_snake_consts(_ptr(alpha), _ptr(beta), _ptr(a), _ptr(ib), alpha.numel(), self._stream(alpha)))
        return a, ib

    def _conv_out(self, out, b, cout, tout, device):
        """The conv output tensor: a fresh one, or the caller's `out` (B, Cout, Tout) fp32 in dense layout — it may be a view at a
        storage offset and may alias `res` element for element (in-place accumulation through the residual input)."""
        if out is None:
            return torch.empty(b, cout, tout, dtype=torch.float32, device=device)
        if tuple(out.shape) != (b, cout, tout) or not out.is_contiguous():
            raise ValueError("conv out= must be a dense (B, Cout, Tout) tensor")
        self._f32(out)
        return out

    def _bf16x3_call(self, fn, rows, x, w_planes, cout, tout, dims, bias, snake, res, tanh_out, dsnake, out=None, sconsts=None, emit=None):
        b, cin, tin = x.shape
        self._f32(x, bias, res)
        sa = sib = None
        if snake is not None:
            sa, sib = sconsts if sconsts is not None else self.snake_consts(snake[0], snake[1])
        y = self._conv_out(out, b, cout, tout, x.device)
        x2 = a2 = b2 = pda = pdb = None
        if dsnake is not None:
            x2, a2, b2 = dsnake
            self._f32(x2, a2, b2)
            pda, pdb = torch.empty(2, cout, rows, dtype=torch.float32, device=x.device).unbind(0)
        if emit is not None:
            # plane emission (sat_conv1d_bf16x3_emit): the planes the k7 conv that consumes y next would otherwise build in a pre-pass
            esnake = emit.get("snake")
            ea = eib = None
            if esnake is not None:
                ea, eib = self.snake_consts(esnake[0], esnake[1])
            ehi, elo, erows = self._emit_planes(b, cout, tout, x.device, self._stream(x))
            self._chk(self.lib.sat_conv1d_bf16x3_emit(_ptr(x), _ptr(w_planes[0]), _ptr(w_planes[1]), _ptr(bias), _ptr(sa), _ptr(sib),
                                                      _ptr(res), _ptr(y), _ptr(x2), _ptr(a2), _ptr(b2), _ptr(pda), _ptr(pdb),
                                                      b, cin, cout, tin, tout, *dims, int(tanh_out), _ptr(ehi), _ptr(elo), _ptr(ea), _ptr(eib),
                                                      erows, self._stream(x)))
            self._note_emitted(y, esnake, ehi, elo, erows)
            if dsnake is not None:
                return (y, *self._sum_pair(pda, pdb))
            return y
        self._chk(fn(_ptr(x), _ptr(w_planes[0]), _ptr(w_planes[1]), _ptr(bias), _ptr(sa), _ptr(sib),
                     _ptr(res), _ptr(y), _ptr(x2), _ptr(a2), _ptr(b2), _ptr(pda), _ptr(pdb),
                     b, cin, cout, tin, tout, *dims, int(tanh_out), self._stream(x)))
        if dsnake is not None:
            return (y, *self._sum_pair(pda, pdb))
        return y

    # ---- fused ResidualUnit forward (csrc/conv1d_bf16x3_k7q.h, FUSED): one launch for snake -> conv7 -> snake -> conv1 -> + x ----
    ru_fused = True

    def ru_fused_ok(self, c, k, dil, t):
        return (self.ru_fused and self.use_bf16x3 and self.k7q and self.k7q_min_cin <= c <= 128
                and 5 <= k <= 7 and (k - 1) * dil <= 62 and (k - 1) * dil % 2 == 0 and (k - 1) * dil // 2 <= 32 and t % 4 == 0)

    def pack_k7q(self, w, mode=0):
        """(D0, D1, K) fp32 -> (hi, lo) planes in sat_pack_weights_k7q layout (K in 5..7, or K = 1 for the fused unit's 1x1 conv)."""
        self._f32(w)
        d0, d1, k = w.shape
        n = self.lib.sat_pack_weights_k7q_size(d0, d1, k, mode)
        if n <= 0:
            raise RuntimeError("sat_pack_weights_k7q: unsupported shape")
        hi = torch.empty(n, dtype=torch.int16, device=w.device)
        lo = torch.empty(n, dtype=torch.int16, device=w.device)
        self._chk(self.lib.sat_pack_weights_k7q(_ptr(w), _ptr(hi), _ptr(lo), d0, d1, k, mode, self._stream(w)))
        return hi, lo

    def residual_unit_fwd(self, x, snake1, w7q, bias1, snake2, w1q, bias2, k, dil, keep_h=True, emit=None, sconsts=None):
        """y = x + conv1(snake2(conv7_dil(snake1(x)) + bias1)) + bias2 in one launch; returns (h or None, y).  w7q / w1q: pack_k7q of the
        (C, C, K) and (C, C, 1) weights; snake1 / snake2: (log-alpha, log-beta); emit: {"snake": (la, lb) | None} -> also write the
        next unit's activation planes (as conv1d_bf16x3(emit=...))."""
        b, c, t = x.shape
        self._f32(x, bias1, bias2)
        pad = (k - 1) * dil // 2
        st = self._stream(x)
        sa1, sib1 = sconsts[0] if sconsts is not None else self.snake_consts(snake1[0], snake1[1])
        sa2, sib2 = sconsts[1] if sconsts is not None else self.snake_consts(snake2[0], snake2[1])
        em = self._take_emitted(x, snake1)
        if em is not None:
            hi, lo, rows = em["hi"], em["lo"], em["rows"]
        else:
            rows = self.lib.sat_conv1d_k7_plane_rows(t, t, pad)
            c8 = (c + 7) // 8
            need = 2 * b * c8 * rows * 8
            wkey = ("k7p", x.device, st.value if st is not None else 0)
            ws = self.__dict__.setdefault("_planes", {}).get(wkey)
            if ws is None or ws.numel() < need:
                ws = _keep_empty(need, torch.int16, x.device)
                self._planes[wkey] = ws
            hi, lo = ws[:need // 2], ws[need // 2:need]
            self._chk(self.lib.sat_conv1d_k7_planes(_ptr(x), _ptr(sa1), _ptr(sib1), _ptr(hi), _ptr(lo), b, c, t, rows, st))
        h = torch.empty_like(x) if keep_h else None
        y = torch.empty_like(x)
        ehi = elo = ea = eib = None
        erows = 0
        if emit is not None:
            esnake = emit.get("snake")
            if esnake is not None:
                ea, eib = self.snake_consts(esnake[0], esnake[1])
            ehi, elo, erows = self._emit_planes(b, c, t, x.device, st)
            if ehi.data_ptr() == hi.data_ptr():
                # the input planes ARE this shape's emission target (written by the previous unit): a workgroup reads halo rows its
                # neighbours' tiles would overwrite -> emit into a second buffer and alternate
                ehi, elo, erows = self._emit_planes(b, c, t, x.device, st, alt=True)
        self._chk(self.lib.sat_residual_unit_fwd(_ptr(hi), _ptr(lo), rows, _ptr(w7q[0]), _ptr(w7q[1]), _ptr(bias1), _ptr(sa2), _ptr(sib2),
                                                 _ptr(w1q[0]), _ptr(w1q[1]), _ptr(bias2), _ptr(x), _ptr(h), _ptr(y), b, c, t, k, dil, pad,
                                                 _ptr(ehi), _ptr(elo), _ptr(ea), _ptr(eib), erows, st))
        if emit is not None:
            self._note_emitted(y, emit.get("snake"), ehi, elo, erows)
        return h, y

    # ---- plane emission bookkeeping: producer -> the ONE k7 conv that consumes its output next ----
    k7_emit = True

    @staticmethod
    def _snake_key(snake):
        return None if snake is None else (snake[0].data_ptr(), snake[1].data_ptr(), _caches.version_of(snake[0]), _caches.version_of(snake[1]))

    def _note_emitted(self, y, snake, hi, lo, rows):
        """Record that (hi, lo) hold act(y)'s planes — valid for exactly this storage, shape, activation and VERSION of y: an
        in-place edit of y between producer and consumer (a forward hook) makes the consumer rebuild the planes.  Inference tensors
        have no version counter, so their emission is never trusted (the consumer runs its planes pre-pass)."""
        if not _caches.trackable(y, *(snake or ())):
            self._emitted = None
            return
        self._emitted = {"ptr": y.data_ptr(), "shape": tuple(y.shape), "ver": _caches.version_of(y), "snake": self._snake_key(snake),
                         "hi": hi, "lo": lo, "rows": rows}

    def emit_ok(self, cout, k, stride, tout, consumer_dil):
        """May the conv (k, stride) producing (B, cout, tout) emit planes for a k7 conv of dilation consumer_dil that reads it next?"""
        if not (self.k7_emit and self.use_bf16x3 and self.k7q_applicable(cout, 7, 1, consumer_dil, 3 * consumer_dil, cout)):
            return False
        generic = (stride == 1 and k <= 4) or stride > 1          # the plans of csrc/conv1d_bf16x3.hip's generic kernel
        return generic and tout % 4 == 0

    def edge_emit_ok(self, cin, cout, k, stride, dil, pad, consumer_dil):
        """May the narrow-input edge conv (the encoder's first conv) emit the planes of the k7 conv (dilation consumer_dil) that reads it next?"""
        return (self.k7_emit and self.use_bf16x3 and cin <= 2 and self.edge_ok(cin, cout, k, stride, dil, pad)
                and self.k7q_applicable(cout, 7, 1, consumer_dil, 3 * consumer_dil, cout))

    def _emit_planes(self, b, c, t, device, st, alt=False):
        """Emission target for a (b, c, t) tensor: planes [b][ceil(c/8)][rows][8] with the rows around the sequence zero.  One pair
        per (shape, device, stream) (+ an alternate for the fused unit, which reads one while writing the other), zero-filled ONCE:
        producers only ever write rows 32 .. 32 + t - 1 of existing channels."""
        rows = self.lib.sat_conv1d_k7_plane_rows(t, t, 0)          # pad 0 needs the most rows: valid for every consumer padding
        key = ("emit", b, c, t, device, st.value if st is not None else 0, alt)
        cache = self.__dict__.setdefault("_planes", {})
        pl = cache.get(key)
        if pl is None:
            n = b * ((c + 7) // 8) * rows * 8
            pl = (_keep_zeros(n, torch.int16, device), _keep_zeros(n, torch.int16, device), rows)
            cache[key] = pl
        return pl

    def _take_emitted(self, x, snake):
        """Planes a producer emitted for exactly this tensor and activation (consumed once), or None."""
        e = self.__dict__.get("_emitted")
        if e is None:
            return None
        self._emitted = None
        if (e["ptr"] == x.data_ptr() and e["shape"] == tuple(x.shape) and _caches.trackable(x) and e["ver"] == _caches.version_of(x)
                and e["snake"] == self._snake_key(snake)):
            return e
        return None

    def conv1d_bf16x3(self, x, w_planes, cout, k, stride=1, dil=1, pad=0, tout=None, bias=None, snake=None, res=None,
                      tanh_out=False, dsnake=None, out=None, sconsts=None, emit=None):
        """Same contract as conv1d; `snake` = (log-alpha, log-beta) as everywhere else; sconsts = snake_consts(*snake) if the
        caller keeps them (frozen layers)."""
        b, cin, tin = x.shape
        if tout is None:
            tout = (tin + 2 * pad - dil * (k - 1) - 1) // stride + 1
        rows = self.lib.sat_conv1d_bf16x3_partial_rows(b, tout, k, stride)
        if len(w_planes) == 3:          # sat_pack_weights_k7q layout (pack_bf16x3(q=True) after k7q_applicable)
            if not (stride == 1 and 5 <= k <= 7 and 0 <= pad <= 32 and (k - 1) * dil <= 62):
                raise ValueError("conv1d_bf16x3: q-packed weights need stride 1, 5 <= K <= 7, pad <= 32, (K-1)*dil <= 62")
            return self._k7_planes_call(rows, x, w_planes, cout, tout, k, dil, pad, bias, snake, res, tanh_out, dsnake, out, sconsts)
        return self._bf16x3_call(self.lib.sat_conv1d_bf16x3, rows, x, w_planes, cout, tout, (k, stride, dil, pad),
                                 bias, snake, res, tanh_out, dsnake, out, sconsts, emit)

    # the k = 7 convs of the ResidualUnits read their (activated) input as pre-split bf16 planes: written by the producer's epilogue
    # (plane emission) or by one conversion pass per conv (sat_conv1d_k7_planes) instead of one per workgroup.  The two planes live in
    # a cached workspace of the largest size seen, one per (device, stream): the pre-pass and its conv are enqueued back to back on
    # the caller's current stream.
    def _k7_planes_call(self, prows, x, w_planes, cout, tout, k, dil, pad, bias, snake, res, tanh_out, dsnake, out=None, sconsts=None):
        b, cin, tin = x.shape
        self._f32(x, bias, res)
        sa = sib = None
        if snake is not None:
            sa, sib = sconsts if sconsts is not None else self.snake_consts(snake[0], snake[1])
        st = self._stream(x)
        em = self._take_emitted(x, snake)
        if em is not None:
            hi, lo, rows = em["hi"], em["lo"], em["rows"]          # the producer's epilogue already wrote act(x) as planes
        else:
            rows = self.lib.sat_conv1d_k7_plane_rows(tin, tout, pad)
            c8 = (cin + 7) // 8
            need = 2 * b * c8 * rows * 8
            wkey = ("k7p", x.device, st.value if st is not None else 0)       # one workspace per (device, stream)
            ws = self.__dict__.setdefault("_planes", {}).get(wkey)
            if ws is None or ws.numel() < need:
                ws = _keep_empty(need, torch.int16, x.device)
                self._planes[wkey] = ws
            hi, lo = ws[:need // 2], ws[need // 2:need]
            self._chk(self.lib.sat_conv1d_k7_planes(_ptr(x), _ptr(sa), _ptr(sib), _ptr(hi), _ptr(lo), b, cin, tin, rows, st))
        y = self._conv_out(out, b, cout, tout, x.device)
        x2 = a2 = b2 = pda = pdb = None
        if dsnake is not None:
            x2, a2, b2 = dsnake
            self._f32(x2, a2, b2)
            pda, pdb = torch.empty(2, cout, prows, dtype=torch.float32, device=x.device).unbind(0)
        self._chk(self.lib.sat_conv1d_bf16x3_planesq(_ptr(hi), _ptr(lo), rows, _ptr(w_planes[0]), _ptr(w_planes[1]), _ptr(bias), _ptr(res),
                                                     _ptr(y), _ptr(x2), _ptr(a2), _ptr(b2), _ptr(pda), _ptr(pdb), b, cin, cout, tin, tout,
                                                     k, dil, pad, int(tanh_out), {True: 0, False: 1, "force": 2}[self.k7q_persist] | (4 if self.k7q_dma_in_mfma else 0), st))
        if dsnake is not None:
            return (y, *self._sum_pair(pda, pdb))
        return y

    def convtr1d_bf16x3(self, x, w_planes, cout, k, stride, pad, tout=None, bias=None, snake=None, res=None,
                        tanh_out=False, dsnake=None, sconsts=None):
        """Same contract as convtr1d."""
        b, cin, tin = x.shape
        if tout is None:
            tout = (tin - 1) * stride - 2 * pad + k
        rows = self.lib.sat_convtr1d_bf16x3_partial_rows(b, tout, stride, pad)
        return self._bf16x3_call(self.lib.sat_convtr1d_bf16x3, rows, x, w_planes, cout, tout, (k, stride, pad),
                                 bias, snake, res, tanh_out, dsnake, None, sconsts)

    def convtr1d(self, x, w_packed, cout, k, stride, pad, tout=None, bias=None, snake=None, res=None,
                 tanh_out=False, dsnake=None):
        b, cin, tin = x.shape
        if tout is None:
            tout = (tin - 1) * stride - 2 * pad + k
        alpha, beta = snake if snake is not None else (None, None)
        self._f32(x, w_packed, bias, alpha, beta, res)
        y = torch.empty(b, cout, tout, dtype=torch.float32, device=x.device)
        x2 = a2 = b2 = pda = pdb = None
        rows = 0
        if dsnake is not None:
            x2, a2, b2 = dsnake
            self._f32(x2, a2, b2)
            rows = self.lib.sat_convtr1d_partial_rows(b, tout, stride, pad)
            if rows < 0:
                raise RuntimeError("sat_convtr1d: unsupported stride")
            pda, pdb = torch.empty(2, cout, rows, dtype=torch.float32, device=x.device).unbind(0)
        self._chk(self.lib.sat_convtr1d(_ptr(x), _ptr(w_packed), _ptr(bias), _ptr(alpha), _ptr(beta), _ptr(res), _ptr(y),
                                        _ptr(x2), _ptr(a2), _ptr(b2), _ptr(pda), _ptr(pdb),
                                        b, cin, cout, tin, tout, k, stride, pad, int(tanh_out), self._stream(x)))
        if dsnake is not None:
            return (y, *self._sum_pair(pda, pdb))
        return y

    def conv_wgrad(self, lo, hi, k, stride=1, dil=1, pad=0, snake=None, snake_on=0, lo_rowsum=False, raw=False):
        """dW[m][n][k] = sum_{b,t} actA(lo[b,m,t]) * actB(hi[b,n,t*s + k*d - pad]), (M, N, K).
        lo_rowsum=True also returns sum_{b,t} lo[b,m,t] (the bias gradient when lo = dy): (dW, (M,)).
        raw=True: dW as the kernel's un-summed slabs (WgradSlabs) for wn_grad_splits."""
        bsz, m, tlo = lo.shape
        _, n, thi = hi.shape
        alpha, beta = snake if snake is not None else (None, None)
        self._f32(lo, hi, alpha, beta)
        x3 = self.use_bf16x3 and dil == 1 and self.lib.sat_conv_wgrad_bf16x3_nsplit(bsz, m, n, tlo, k, stride) > 0
        if x3:
            nsplit = self.lib.sat_conv_wgrad_bf16x3_nsplit(bsz, m, n, tlo, k, stride)
        else:
            nsplit = self.lib.sat_conv_wgrad_nsplit(bsz, m, n, tlo, k, stride, dil)
        if nsplit < 0:
            raise RuntimeError("sat_conv_wgrad: receptive field too large")
        partial = torch.empty(nsplit, m * n * k, dtype=torch.float32, device=lo.device)
        so_m, so_n, so_k = n * k, k, 1
        rs = None
        if x3:
            fused = lo_rowsum and not (snake is not None and snake_on == 1)     # the kernel sums the rows it stages anyway
            rs = torch.empty(m, nsplit, dtype=torch.float32, device=lo.device) if fused else None
            self._chk(self.lib.sat_conv_wgrad_bf16x3(_ptr(lo), _ptr(hi), _ptr(alpha), _ptr(beta),
                                                     snake_on if snake is not None else 0, _ptr(partial), so_m, so_n, so_k,
                                                     bsz, m, n, tlo, thi, k, stride, pad, _ptr(rs), self._stream(lo)))
        else:
            self._chk(self.lib.sat_conv_wgrad(_ptr(lo), _ptr(hi), _ptr(alpha), _ptr(beta), snake_on if snake is not None else 0,
                                              _ptr(partial), so_m, so_n, so_k, bsz, m, n, tlo, thi, k, stride, dil, pad,
                                              self._stream(lo)))
        slabs = WgradSlabs(partial, nsplit, (m, n, k), (so_m, so_n, so_k))
        dw = slabs if raw else slabs.reduce(self)
        if not lo_rowsum:
            return dw
        if raw:
            return dw, (rs if rs is not None else self.rowsum(lo, partial=True))
        return dw, (self._sum_last(rs) if rs is not None else self.rowsum(lo))

    # ---- fused backward of a ResidualUnit's 1x1 conv (csrc/ru_k1_bwd.hip): one pass over dy and h ----
    ru_k1_fused = True

    def ru_k1_bwd_ok(self, b, c, t):
        return self.ru_k1_fused and self.use_bf16x3 and self.lib.sat_ru_k1_bwd_nsplit(b, c, t) > 0

    def ru_k1_pack(self, w2):
        """(C, C, 1) fp32 -> W2^T as bf16 hi / lo planes [ci][co] (sat_ru_k1_pack)."""
        self._f32(w2)
        c = w2.shape[0]
        planes = torch.empty(2, c * c, dtype=torch.int16, device=w2.device)
        self._chk(self.lib.sat_ru_k1_pack(_ptr(w2), _ptr(planes[0]), _ptr(planes[1]), c, self._stream(w2)))
        return planes[0], planes[1]

    def ru_k1_bwd(self, dy, h, w2, snake2, emit=False, wt=None, raw=False):
        """Backward of y = x + conv1x1(snake2(h)) w.r.t. everything but x, in one launch: returns (dh, dlog_alpha2, dlog_beta2, dW2 (C, C, 1),
        dbias2 (C,), dbias1 (C,) = sum dh).  emit=True also writes dh as the activation planes of the k7 data-gradient that consumes it
        next (as conv1d_bf16x3(emit={"snake": None})).  raw=True: dW2 as WgradSlabs (for wn_grad_splits)."""
        b, c, t = dy.shape
        a2, b2 = snake2
        self._f32(dy, h, w2, a2, b2)
        ns = self.lib.sat_ru_k1_bwd_nsplit(b, c, t)
        if ns <= 0:
            raise RuntimeError("ru_k1_bwd: shape not served by the fused kernel (ru_k1_bwd_ok)")
        st = self._stream(dy)
        wt_hi, wt_lo = wt if wt is not None else self.ru_k1_pack(w2)
        dh = torch.empty_like(dy)
        slabs = torch.empty(ns, c * c, dtype=torch.float32, device=dy.device)
        part = torch.empty(4 * c, ns, dtype=torch.float32, device=dy.device)
        ehi = elo = None
        erows = 0
        if emit:
            ehi, elo, erows = self._emit_planes(b, c, t, dy.device, st)
        self._chk(self.lib.sat_ru_k1_bwd(_ptr(dy), _ptr(h), _ptr(wt_hi), _ptr(wt_lo), _ptr(a2), _ptr(b2), _ptr(dh), _ptr(ehi), _ptr(elo), erows,
                                         _ptr(slabs), _ptr(part), b, c, t, st))
        if emit:
            self._note_emitted(dh, None, ehi, elo, erows)
        dw2 = WgradSlabs(slabs, ns, (c, c, 1), (c, 1, 1))
        if not raw:
            dw2 = dw2.reduce(self)
        sums = self._sum_last(part)
        return dh, sums[:c], sums[c:2 * c], dw2, sums[3 * c:], sums[2 * c:3 * c]

    def wgrad7_bf16x3_ok(self, n_in, k, stride, dil):
        return self.use_bf16x3 and stride == 1 and k == 7 and dil in (1, 3, 9)

    def conv_wgrad7_bf16x3(self, dy, x, dil, pad, snake=None, dy_rowsum=False, raw=False):
        """dW (Cout, Cin, 7) of a k7 stride-1 conv: dy (B, Cout, T), x (B, Cin, T) pre-activation, snake = (log-alpha, log-beta).
        dy_rowsum=True also returns the bias gradient sum_{b,t} dy (fused into the kernel): (dW, (Cout,)).
        raw=True: dW as the kernel's un-summed slabs (WgradSlabs) for wn_grad_splits."""
        b, m, t = dy.shape
        n = x.shape[1]
        alpha, beta = snake if snake is not None else (None, None)
        self._f32(dy, x, alpha, beta)
        nsplit = self.lib.sat_conv_wgrad7_bf16x3_nsplit(b, m, n, t)
        partial = torch.empty(nsplit, m * n * 7, dtype=torch.float32, device=dy.device)
        # slabs are written tap-major ([7][M][N]: the 32 lanes of an accumulator row store 128 contiguous bytes; the
        # reference (M, N, 7) order would scatter 4-byte stores 28 bytes apart — 6.7x the write traffic, measured)
        fused = dy_rowsum and self.lib.sat_conv_wgrad7_bf16x3_fuses_rowsum(b, m, n, t) == 1
        rs = torch.empty(m, nsplit, dtype=torch.float32, device=dy.device) if fused else None
        self._chk(self.lib.sat_conv_wgrad7_bf16x3(_ptr(dy), _ptr(x), _ptr(alpha), _ptr(beta), _ptr(partial), n, 1, m * n,
                                                  b, m, n, t, dil, pad, _ptr(rs), self._stream(dy)))
        slabs = WgradSlabs(partial, nsplit, (m, n, 7), (n, 1, m * n))
        dw = slabs if raw else slabs.reduce(self)
        if not dy_rowsum:
            return dw
        if raw:         # the bias gradient one reduction short of done: wn_grad_splits finishes it
            return dw, (rs if fused else self.rowsum(dy, partial=True))
        return dw, (self._sum_last(rs) if fused else self.rowsum(dy))

    def rowsum(self, x, partial=False):
        """(B, C, T) -> (C,) sum over batch and time: per-(channel, time split) partial sums laid out [C][nsplit], summed
        by a second pass of the same kernel (deterministic, no atomics).  partial=True: the first pass only, (C, nsplit) — for a
        consumer that finishes the sum itself (wn_grad_splits)."""
        self._f32(x)
        b, c, t = x.shape
        while True:
            ns = self.lib.sat_rowsum_nsplit(t)
            part = torch.empty(c, ns, dtype=torch.float32, device=x.device)
            self._chk(self.lib.sat_rowsum(_ptr(x), _ptr(part), b, c, t, self._stream(x)))
            if partial:
                return part
            if ns == 1:
                return part.view(c)
            x, b, t = part, 1, ns

    def sum_all(self, x):
        """Sum of EVERY element of a tensor as a 0-d fp32 tensor: passes of sat_rowsum over (1, 1, N) until one partial is left —
        deterministic, no atomics and, unlike torch's `x.sum()` / `.mean()` / `.norm()` over millions of elements, no semaphore
        buffer zeroed by a memset in front of the kernel.  That matters under HIP-graph replay: on this stack (torch 2.10 / ROCm 7)
        a replayed multi-block torch reduction returns stale or foreign values after a few replays — reproduced with torch ops alone
        (tools/diag_graph_reduce.py, profiles/r04_experiments/graph_reductions/) — so no reduction to a scalar inside the training
        step goes through one (functional.sum_all / mean_all are the autograd forms)."""
        x = x.contiguous()
        if x.dtype != torch.float32:
            x = x.float()
        n = x.numel()
        if n == 0:
            return torch.zeros((), dtype=torch.float32, device=x.device)
        if n > 1 << 29:           # sat_rowsum splits a row into 16384-element pieces along grid.y (<= 65535): sum 2^29-element pieces
            flat = x.view(-1)
            return torch.stack([self.sum_all(piece) for piece in flat.split(1 << 29)]).sum()
        if x.data_ptr() % 16:
            x = x.clone()
        return self.rowsum(x.view(1, 1, n)).view(())

    # ------------------------------------------------------------------ VAE bottleneck
    def vae_sample_fwd(self, pre, noise):
        self._f32(pre, noise)
        b, c2, t = pre.shape
        c = c2 // 2
        z = torch.empty(b, c, t, dtype=torch.float32, device=pre.device)
        nb = self.lib.sat_vae_nblocks(b * c * t)
        klp = torch.empty(nb, dtype=torch.float32, device=pre.device)
        self._chk(self.lib.sat_vae_sample_fwd(_ptr(pre), _ptr(noise), _ptr(z), _ptr(klp), b, c, t, self._stream(pre)))
        kl = (self._sum_last(klp.view(1, nb)) * (1.0 / (b * t))).view(())
        return z, kl

    def vae_sample_bwd(self, pre, noise, dz, dkl):
        """dz: (B, C, T) or None; dkl: 0-dim device tensor (dL/dkl) or None."""
        self._f32(pre, noise, dz, dkl)
        b, c2, t = pre.shape
        c = c2 // 2
        dpre = torch.empty_like(pre)
        self._chk(self.lib.sat_vae_sample_bwd(_ptr(pre), _ptr(noise), _ptr(dz), _ptr(dkl), _ptr(dpre), b, c, t,
                                              self._stream(pre)))
        return dpre

    # ------------------------------------------------------------------ MR-STFT loss
    def fir(self, x, taps, adjoint=False):
        """x: (N, T) -> cross-correlation with `taps` (zero pad ntaps//2); adjoint=True applies the transpose."""
        self._f32(x, taps)
        n, t = x.shape
        y = torch.empty_like(x)
        self._chk(self.lib.sat_fir(_ptr(x), _ptr(taps), _ptr(y), n, t, taps.numel(), int(adjoint), self._stream(x)))
        return y

    def stft_sums(self, x, y, views, n_fft, hop):
        """x, y: (NI, C, T); views: (NV, 2).  Returns (NI, NV, 3) = [sum(|Y|-|X|)^2, sum|Y|^2, sum|log|X|-log|Y||]."""
        self._f32(x, y, views)
        ni, c, t = x.shape
        nv = views.shape[0]
        tiles = self.lib.sat_stft_tiles(n_fft, hop, t)
        if tiles < 0:
            raise RuntimeError(f"sat_stft: unsupported n_fft={n_fft} hop={hop} T={t}")
        partial = torch.empty(ni * nv * 3, tiles, dtype=torch.float32, device=x.device)
        self._chk(self.lib.sat_stft_fwd(_ptr(x), _ptr(y), _ptr(views), _ptr(partial), ni, c, t, nv, n_fft, hop, self._stream(x)))
        return self._sum_last(partial).view(ni, nv, 3)

    def stft_backward(self, x, y, views, coef, planes, n_fft, hop, wrt_x=False):
        """dL/dy (or dL/dx with wrt_x) of one resolution, written WITHOUT atomics into `planes` (4, NI, C, T) — zero-filled by the
        caller; [direct-even | direct-odd | mirror-even | mirror-odd] workgroup / reflection classes, plain stores — whose sum over
        dim 0 is the gradient (bit-reproducible).  coef: (NI, NV, 3) = (c1, c2, c3) — see csrc/stft.hip."""
        self._f32(x, y, views, coef, planes)
        ni, c, t = x.shape
        nv = views.shape[0]
        if tuple(planes.shape) != (4, ni, c, t):
            raise ValueError("stft_backward: planes must be (4, NI, C, T)")
        self._chk(self.lib.sat_stft_bwd(_ptr(x), _ptr(y), _ptr(views), _ptr(coef), _ptr(planes), ni, c, t, nv, n_fft, hop,
                                        int(wrt_x), self._stream(x)))

    # ------------------------------------------------------------------ discriminator spectrogram
    # ---- row packing for the discriminator's Conv2d layers as virtual-channel 1-D convs (discriminators.conv2d_virtual) ----
    def rows_pack(self, x, kh, dil_t, pad_t, pad_w, pitch, lead):
        """x (B, C, T, W) fp32 -> flat buffer of lead + B*C*kh*T*pitch + lead floats (see sat_rows_pack in sat_amd.h)."""
        self._f32(x)
        b, c, t, w = x.shape
        buf = torch.empty(2 * lead + b * c * kh * t * pitch, dtype=torch.float32, device=x.device)
        self._chk(self.lib.sat_rows_pack(_ptr(x), _ptr(buf), b, c, t, w, kh, dil_t, pad_t, pad_w, pitch, lead, self._stream(x)))
        return buf

    def rows_pack_bwd(self, dbuf, shape, kh, dil_t, pad_t, pad_w, pitch, lead):
        self._f32(dbuf)
        b, c, t, w = shape
        dx = torch.empty(shape, dtype=torch.float32, device=dbuf.device)
        self._chk(self.lib.sat_rows_pack_bwd(_ptr(dbuf), _ptr(dx), b, c, t, w, kh, dil_t, pad_t, pad_w, pitch, lead, self._stream(dbuf)))
        return dx

    def release_workspaces(self):
        """Drop the cached plane / emission buffers (they are per activation shape and re-created, zero-filled, on demand): call after a
        run at a batch size or length that will not come back, before torch.cuda.empty_cache()."""
        for name in ("_planes", "_disc_pool", "_disc_gen", "_disc_geoms"):
            d = self.__dict__.get(name)
            if d is not None:
                d.clear()
        self._emitted = None
        self._disc_emitted = None

    # ---- the discriminator's Conv2d layers on the pitched-rows layout (csrc/disc_conv.hip; discriminators._DiscConvFn) ----
    def disc_geom(self, frames, w):
        """(P, L, lead, rows) of the pitched sequence / planes of a (.., frames, w) activation (sat_disc_geom)."""
        key = ("disc_geom", frames, w)
        cache = self.__dict__.setdefault("_disc_geoms", {})
        g = cache.get(key)
        if g is None:
            out = [ctypes.c_int() for _ in range(4)]
            self._chk(self.lib.sat_disc_geom(frames, w, *[ctypes.byref(o) for o in out]))
            g = cache[key] = tuple(o.value for o in out)
        return g

    def _disc_plane_buf(self, b, c, frames, w, device, slot):
        """bf16 hi / lo plane pair [b][ceil(c/8)][rows][8] for a (b, c, frames, w) activation; `slot` 0 / 1: the two alternating
        targets of a layer chain (a layer reads one and emits into the other).  ONE pool entry per (device, narrow | wide, slot) serves
        every scale of the discriminator (the scales run one after the other): writers only touch rows lead .. lead + L - 1 of the
        existing channel groups, so when an entry changes geometry the rows around them — the frame padding — are zeroed again."""
        P, L, lead, rows = self.disc_geom(frames, w)
        c8 = (c + 7) // 8
        n = b * c8 * rows * 8
        key = (device, 0 if c8 == 1 else 1, slot)
        gen = self.__dict__.setdefault("_disc_gen", {})
        gen[key] = gen.get(key, 0) + 1                              # every request is a write: invalidates earlier registrations
        pool = self.__dict__.setdefault("_disc_pool", {})
        e = pool.get(key)
        if e is None or e["cap"] < n:
            e = pool[key] = {"hi": _keep_zeros(n, torch.int16, device), "lo": _keep_zeros(n, torch.int16, device),
                             "cap": n, "geom": (b, c8, frames, w)}
        elif e["geom"] != (b, c8, frames, w):
            for t in (e["hi"], e["lo"]):
                v = t[:n].view(b * c8, rows, 8)
                v[:, :lead].zero_()
                v[:, lead + L:].zero_()
            e["geom"] = (b, c8, frames, w)
        return e["hi"][:n], e["lo"][:n]

    def disc_register(self, t, planes, c, frames, w, slot):
        """Remember that `planes` (slot `slot`) hold tensor t's operand planes (consumed by disc_take if nothing wrote the slot since)."""
        key = (t.device, 0 if (c + 7) // 8 == 1 else 1, slot)
        if not _caches.trackable(t):          # inference tensor: no version counter, the emission cannot be validated later
            self._disc_emitted = None
            return
        self._disc_emitted = {"ptr": t.data_ptr(), "ver": _caches.version_of(t), "key": key, "gen": self.__dict__.get("_disc_gen", {}).get(key, 0),
                              "geom": (t.shape[0], (c + 7) // 8, frames, w), "planes": planes, "slot": slot}

    def disc_take(self, h, c, frames, w):
        """((hi, lo) planes of the pitched tensor h, their slot): the producer's emission if it is still intact, else a planes pass."""
        e = self.__dict__.get("_disc_emitted")
        self._disc_emitted = None
        if (e is not None and e["ptr"] == h.data_ptr() and _caches.trackable(h) and e["ver"] == _caches.version_of(h) and e["geom"] == (h.shape[0], (c + 7) // 8, frames, w)
                and self.__dict__.get("_disc_gen", {}).get(e["key"], 0) == e["gen"]):
            return e["planes"], e["slot"]
        return self.disc_planes(h, frames, w, slot=0)[1], 0

    def disc_planes(self, src, frames, w, out=None, slope=1.0, want_dst=False, want_planes=True, slot=0, fm_sign=None, fm_coef=None):
        """src: (B, C, frames, w) or pitched (B, C, L) -> (dst pitched fp32 or None, (hi, lo) planes or None); with `out` (pitched):
        (src + fm_coef * fm_sign) * LeakyReLU'(out) — fm_sign (int8, disc_l1_sum) / fm_coef (a device scalar) optional."""
        self._f32(src, out, fm_coef)
        if fm_sign is not None and (fm_sign.dtype != torch.int8 or not fm_sign.is_contiguous() or fm_sign.numel() != src.numel()):
            raise ValueError("disc_planes: fm_sign must be a contiguous int8 tensor of src's size")
        b, c = src.shape[0], src.shape[1]
        P, L, lead, rows = self.disc_geom(frames, w)
        pitched = src.dim() == 3
        if pitched and src.shape[2] != L or not pitched and tuple(src.shape[2:]) != (frames, w):
            raise ValueError("disc_planes: shape does not match (frames, w)")
        dst = torch.empty(b, c, L, dtype=torch.float32, device=src.device) if want_dst else None
        pl = self._disc_plane_buf(b, c, frames, w, src.device, slot) if want_planes else None
        self._chk(self.lib.sat_disc_planes(_ptr(src), _ptr(out), _ptr(fm_sign), _ptr(fm_coef), _ptr(dst), _ptr(pl[0]) if pl else None,
                                           _ptr(pl[1]) if pl else None, b, c, frames, w, 1 if pitched else 0, float(slope), self._stream(src)))
        return dst, pl

    def disc_l1_sum(self, a, b, want_sign=False):
        """sum |a - b| of two equal-shape contiguous tensors (numel % 4 == 0) as a 0-d tensor (+ sign(a - b) as int8 if asked)."""
        self._f32(a, b)
        partial = torch.empty(self.lib.sat_disc_l1_blocks(), dtype=torch.float32, device=a.device)
        sign = torch.empty(a.shape, dtype=torch.int8, device=a.device) if want_sign else None
        self._chk(self.lib.sat_disc_l1_sum(_ptr(a), _ptr(b), _ptr(partial), _ptr(sign), a.numel(), self._stream(a)))
        return (partial.sum(), sign) if want_sign else partial.sum()

    def disc_pack(self, w4, mode):
        """w (Cout, Cin, kh, kw) -> the packed bf16 hi + lo weights for sat_disc_conv (mode 0) / its data-gradient (mode 1)."""
        self._f32(w4)
        cout, cin, kh, kw = w4.shape
        n = self.lib.sat_disc_pack_size(cout, cin, kh, kw, mode)
        if n < 0:
            raise ValueError("disc_pack: unsupported kernel shape")
        wq = torch.empty(n, dtype=torch.int16, device=w4.device)
        self._chk(self.lib.sat_disc_pack_weights(_ptr(w4), _ptr(wq), cout, cin, kh, kw, mode, self._stream(w4)))
        return wq

    def disc_conv(self, planes, wq, bias, b, cin, cout, frames, w, kh, kw, dil_t, slope, emit_slot=None, lk_src=None, lk_slope=1.0):
        """LeakyReLU_slope(conv2d + bias) on the pitched layout: planes (hi, lo) of the (b, cin) input, wq = disc_pack(...).
        Returns (y (b, cout, L), emitted planes of y or None).  lk_src (b, cout, L): y *= LeakyReLU'(lk_src) (slope lk_slope)."""
        self._f32(bias, lk_src)
        L = self.disc_geom(frames, w)[1]
        y = torch.empty(b, cout, L, dtype=torch.float32, device=planes[0].device)
        em = self._disc_plane_buf(b, cout, frames, w, y.device, emit_slot) if emit_slot is not None else None
        self._chk(self.lib.sat_disc_conv(_ptr(planes[0]), _ptr(planes[1]), _ptr(wq), _ptr(bias), _ptr(y),
                                         _ptr(em[0]) if em else None, _ptr(em[1]) if em else None, b, cin, cout, frames, w, kh, kw, dil_t,
                                         float(slope), _ptr(lk_src), float(lk_slope), self._stream(y)))
        return y, em

    def disc_wgrad(self, dy, x, frames, w, kh, kw, dil_t):
        """dW (M, Cin, kh, kw) from dy = dL/d(pre-activation) (B, M, L) and the layer input x (B, Cin, L), both pitched."""
        self._f32(dy, x)
        b, m, _ = dy.shape
        cin = x.shape[1]
        nsplit = self.lib.sat_disc_wgrad_nsplit(b, m, cin, kh, frames, w)
        mp, npad = (m + 63) // 64 * 64, (kh * cin + 63) // 64 * 64
        partial = torch.empty(nsplit, kw * mp * npad, dtype=torch.float32, device=dy.device)
        self._chk(self.lib.sat_disc_wgrad(_ptr(dy), _ptr(x), _ptr(partial), b, m, cin, frames, w, kh, kw, dil_t, self._stream(dy)))
        dw = self._reduce_rows(partial, nsplit, kw * mp * npad).view(kw, mp, npad)[:, :m, :kh * cin]
        return dw.reshape(kw, m, kh, cin).permute(1, 3, 2, 0).contiguous()

    def rows_unpack(self, y, t, w, pad_w, pitch, slope):
        """y (B, C, T*pitch) -> leaky_relu(y[..., t*pitch + pad_w + w], slope) as (B, C, T, W)."""
        self._f32(y)
        b, c = y.shape[0], y.shape[1]
        out = torch.empty(b, c, t, w, dtype=torch.float32, device=y.device)
        self._chk(self.lib.sat_rows_unpack(_ptr(y), _ptr(out), b, c, t, w, pad_w, pitch, float(slope), self._stream(y)))
        return out

    def rows_unpack_bwd(self, dout, out, pad_w, pitch, slope):
        self._f32(dout, out)
        b, c, t, w = dout.shape
        dy = torch.empty(b, c, t * pitch, dtype=torch.float32, device=dout.device)
        self._chk(self.lib.sat_rows_unpack_bwd(_ptr(dout), _ptr(out), _ptr(dy), b, c, t, w, pad_w, pitch, float(slope), self._stream(dout)))
        return dy

    def spec_fwd(self, x, n_fft, hop):
        """x (NI, C, T), C in {1, 2} -> (NI, 2C, frames, n_fft/2+1): [Re X_c | Im X_c] of the normalised, un-centred STFT."""
        self._f32(x)
        ni, c, t = x.shape
        frames = self.lib.sat_spec_frames(n_fft, hop, t)
        if frames < 0:
            raise RuntimeError(f"sat_spec: unsupported n_fft={n_fft} hop={hop} T={t}")
        z = torch.empty(ni, 2 * c, frames, n_fft // 2 + 1, dtype=torch.float32, device=x.device)
        self._chk(self.lib.sat_spec_fwd(_ptr(x), _ptr(z), ni, c, t, n_fft, hop, self._stream(x)))
        return z

    def spec_bwd(self, dz, c, t, n_fft, hop):
        """Adjoint of spec_fwd: dz (NI, 2C, frames, bins) -> dx (NI, C, T) (two write-once planes summed here)."""
        self._f32(dz)
        ni = dz.shape[0]
        planes = torch.zeros(2, ni, c, t, dtype=torch.float32, device=dz.device)
        self._chk(self.lib.sat_spec_bwd(_ptr(dz), _ptr(planes), ni, c, t, n_fft, hop, self._stream(dz)))
        return planes[0] + planes[1]

    # ------------------------------------------------------------------ DiT operators
    def _dt(self, *tensors):
        """dtype code for the DiT kernels: 0 = fp32, 1 = bf16 (all given tensors must agree)."""
        dt = None
        for t in tensors:
            if t is None:
                continue
            if t.dtype not in (torch.float32, torch.bfloat16):
                raise TypeError(f"DiT kernels take float32 or bfloat16 tensors, got {t.dtype}")
            if dt is not None and t.dtype != dt:
                raise TypeError("mixed dtypes")
            dt = t.dtype
            if not self.simulator and not t.is_cuda:
                raise RuntimeError("stable_audio_tools_amd kernels need CUDA(HIP) tensors; there is no CPU path")
        return 0 if dt == torch.float32 else 1

    def attn_planes(self, t, row_major=True, transposed=False):
        """(B, H, N, 64) strided fp32|bf16 (head dim contiguous) -> bf16 planes, zero padded to Np = ceil64(N):
        dict with 'rm' = (hi, lo) of shape (B, H, Np, 64) and/or 'tr' = (hi, lo) of shape (B, H, 64, Np);
        lo is None for bf16 sources."""
        dt = self._dt(t)
        if t.stride(3) != 1:
            raise ValueError("head dim must be contiguous")
        b, h, n, d = t.shape
        npad = (n + 63) // 64 * 64
        split = dt == 0

        def alloc(shape, want):
            return torch.empty(shape, dtype=torch.int16, device=t.device) if want else None
        rm_hi, rm_lo = alloc((b, h, npad, d), row_major), alloc((b, h, npad, d), row_major and split)
        tr_hi, tr_lo = alloc((b, h, d, npad), transposed), alloc((b, h, d, npad), transposed and split)
        self._chk(self.lib.sat_attn_prepare(_ptr(t), t.stride(0), t.stride(1), t.stride(2), _ptr(rm_hi), _ptr(rm_lo),
                                            _ptr(tr_hi), _ptr(tr_lo), b, h, n, npad, dt, self._stream(t)))
        return {"rm": (rm_hi, rm_lo), "tr": (tr_hi, tr_lo), "n": n, "np": npad, "dt": dt}

    def attention(self, q, k, v, scale, need_lse=False, return_planes=False):
        """q: (B, H, Nq, 64) k, v: (B, Hkv, Nk, 64) — any strides with the head dim contiguous.
        Returns o: (B, Nq, H*64) [, lse (B, H, Nq) fp32] [, planes dict for the backward]."""
        dt = self._dt(q, k, v)
        b, h, nq, d = q.shape
        _, hk, nk, _ = k.shape
        qp = self.attn_planes(q, row_major=True, transposed=return_planes)
        kp = self.attn_planes(k, row_major=True, transposed=return_planes)
        vp = self.attn_planes(v, row_major=return_planes, transposed=True)
        o = torch.empty(b, nq, h * d, dtype=q.dtype, device=q.device)
        lse = torch.empty(b, h, nq, dtype=torch.float32, device=q.device) if (need_lse or return_planes) else None
        if self.cross_ok(h, hk, nk, d, dt):      # short key sequence (the DiT's cross-attention): all keys resident, one-pass softmax
            self._chk(self.lib.sat_attention_cross_fwd(_ptr(qp["rm"][0]), _ptr(kp["rm"][0]), _ptr(vp["tr"][0]), _ptr(o), _ptr(lse), b, h, hk,
                                                       nq, nk, qp["np"], kp["np"], d, float(scale), self._stream(q)))
        else:
            self._chk(self.lib.sat_attention_fwd(_ptr(qp["rm"][0]), _ptr(qp["rm"][1]), _ptr(kp["rm"][0]), _ptr(kp["rm"][1]),
                                                 _ptr(vp["tr"][0]), _ptr(vp["tr"][1]), _ptr(o), _ptr(lse), b, h, hk, nq, nk,
                                                 qp["np"], kp["np"], d, float(scale), self._fwd_dtype(dt), self._stream(q)))
        out = (o,)
        if need_lse or return_planes:
            out += (lse,)
        if return_planes:
            out += ({"q": qp, "k": kp, "v": vp},)
        return out[0] if len(out) == 1 else out

    def attention_bwd(self, planes, o, do, lse, scale, hkv, nk, out=None):
        """Gradients (dq (B,H,Nq,64), dk, dv (B,Hkv,Nk,64)) in the dtype of `o`; `planes` from attention(return_planes=True).
        out: optional (dq, dk, dv) contiguous tensors of those shapes to write into (e.g. slices of one buffer)."""
        dt = self._dt(o, do)
        b, nq, hd = o.shape
        h = hd // 64
        do = do.contiguous()
        dsum = torch.empty(b, h, nq, dtype=torch.float32, device=o.device)
        self._chk(self.lib.sat_attention_rowdot(_ptr(do), _ptr(o), _ptr(dsum), b, h, nq, dt, self._stream(o)))
        gp = self.attn_planes(do.view(b, nq, h, 64).permute(0, 2, 1, 3), row_major=True, transposed=True)
        qp, kp, vp = planes["q"], planes["k"], planes["v"]
        ptrs = [qp["rm"], kp["rm"], vp["rm"], kp["tr"], qp["tr"], gp["rm"], gp["tr"], (None, None)]
        arr = (ctypes.c_void_p * 16)(*[(x.data_ptr() if x is not None else None) for pair in ptrs for x in pair])
        if out is not None:
            dq, dk, dv = out
            if (tuple(dq.shape), tuple(dk.shape), tuple(dv.shape)) != ((b, h, nq, 64), (b, hkv, nk, 64), (b, hkv, nk, 64)) \
                    or not (dq.is_contiguous() and dk.is_contiguous() and dv.is_contiguous()) or {dq.dtype, dk.dtype, dv.dtype} != {o.dtype}:
                raise ValueError("attention_bwd: out must be contiguous (B,H,Nq,64) / (B,Hkv,Nk,64) x 2 tensors in the dtype of o")
        else:
            dq = torch.empty(b, h, nq, 64, dtype=o.dtype, device=o.device)
            dk = torch.empty(b, hkv, nk, 64, dtype=o.dtype, device=o.device)
            dv = torch.empty(b, hkv, nk, 64, dtype=o.dtype, device=o.device)
        if self.cross_ok(h, hkv, nk, 64, dt):
            nbytes = int(self.lib.sat_attention_cross_bwd_ws(b, h, hkv, nq, nk))
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=o.device)      # fp32 dK / dV slabs of the query ranges
            self._chk(self.lib.sat_attention_cross_bwd(arr, _ptr(lse), _ptr(dsum), _ptr(dq), _ptr(dk), _ptr(dv), _ptr(ws), nbytes, b, h, hkv,
                                                       nq, nk, qp["np"], kp["np"], 64, float(scale), self._stream(o)))
        else:
            self._chk(self.lib.sat_attention_bwd(arr, _ptr(lse), _ptr(dsum), _ptr(dq), _ptr(dk), _ptr(dv), b, h, hkv, nq, nk,
                                                 qp["np"], kp["np"], 64, float(scale), dt, self._stream(o)))
        return dq, dk, dv

    # the short-key attention kernels (csrc/attention_cross.h) serve bf16 planes with Nk <= 256; False sends every shape to the general
    # flash-style kernels (A/B runs: bench.py --ops-set cross_kernels=0, and the kernel tests' second implementation)
    cross_kernels = True
    # DiT training glue (round 6): rotary + head split + attention core as ONE autograd node (transformer._SelfAttnFn / _CrossAttnFn), both
    # bf16 copies of a weight from one cast launch (sat_cast_bf16_dual), the bias row of the weight-gradient GEMM's operand kept in a cached
    # buffer instead of two fills per call.  False: the separate nodes / launches of round 5 (A/B: bench.py --ops-set train_fused_nodes=0)
    train_fused_nodes = True
    ln_residual = True      # training: the residual path's gradient is added inside the LayerNorm backward kernel (transformer.LayerNormResFn); False: autograd's add
    cast_pair = True        # both transposed operands of a weight-gradient GEMM from one launch (sat_cast_bf16_tpair); False: two sat_cast_bf16 launches
    # bf16 self-attention forward: None = the library picks 32 or 64 queries per wave by grid size (sat_attention_fwd), False / True force
    # the 32- / 64-query kernel (A/B runs: bench.py --ops-set attn_q64=1; the kernel tests run both)
    attn_q64 = None

    def _fwd_dtype(self, dt):
        if dt != 1 or self.attn_q64 is None:
            return dt
        return 3 if self.attn_q64 else 2

    def cross_ok(self, h, hkv, nk, d, dt):
        return bool(self.cross_kernels) and bool(self.lib.sat_attention_cross_ok(h, hkv, nk, d, dt))

    def layernorm(self, x, gamma, beta=None, scale=None, shift=None, eps=1e-5, save_stats=False):
        """x: (B, N, D) contiguous; gamma/beta fp32 (D,); scale/shift: (B, D) views (last dim contiguous) for adaLN."""
        dt = self._dt(x, scale, shift)
        self._f32(gamma, beta)
        if not x.is_contiguous():
            raise ValueError("expected contiguous x")
        b, n, d = x.shape
        y = torch.empty_like(x)
        mean = rstd = None
        if save_stats:
            mean = torch.empty(b * n, dtype=torch.float32, device=x.device)
            rstd = torch.empty(b * n, dtype=torch.float32, device=x.device)
        ms = scale.stride(0) if scale is not None else 0
        self._chk(self.lib.sat_layernorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(scale), _ptr(shift), ms, _ptr(y),
                                             _ptr(mean), _ptr(rstd), b * n, d, n, eps, dt, self._stream(x)))
        return (y, mean, rstd) if save_stats else y

    def layernorm_fp8(self, x, gamma, beta=None, scale=None, shift=None, eps=1e-5):
        """LayerNorm (+ adaLN modulate) with the output quantised per row to fp8 e4m3 (sat_layernorm_fwd_fp8): returns (q (B*N, D) uint8,
        row_scale (B*N,) fp32), or None when the shape is outside the kernel's vector path (the caller then normalises and quantises
        in two steps)."""
        dt = self._dt(x, scale, shift)
        self._f32(gamma, beta)
        if not x.is_contiguous():
            raise ValueError("expected contiguous x")
        b, n, d = x.shape
        q = torch.empty(b * n, d, dtype=torch.uint8, device=x.device)
        rs = torch.empty(b * n, dtype=torch.float32, device=x.device)
        ms = scale.stride(0) if scale is not None else 0
        rc = self.lib.sat_layernorm_fwd_fp8(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(scale), _ptr(shift), ms, _ptr(q), _ptr(rs),
                                            b * n, d, n, eps, dt, self._stream(x))
        if rc == 2:
            return None
        self._chk(rc)
        return q, rs

    def layernorm_bwd(self, dy, x, gamma, beta, scale, mean, rstd, dres=None):
        """Returns dx, dgamma (D,), dscale (B, D) or None, dshift (B, D) or None.  dres: optional (B, N, D) contiguous addend of dx in the
        activation dtype (the gradient that reached x along the residual path: sat_layernorm_bwd_res)."""
        dt = self._dt(dy, x, scale, dres)
        if dres is not None and (dres.shape != x.shape or not dres.is_contiguous()):
            raise ValueError("layernorm_bwd: dres must be contiguous with the shape of x")
        b, n, d = x.shape
        dx = torch.empty_like(x)
        nby = self.lib.sat_layernorm_bwd_nblocks(b * n, n)
        part = torch.empty(3, nby, d, dtype=torch.float32, device=x.device)
        ms = scale.stride(0) if scale is not None else 0
        self._chk(self.lib.sat_layernorm_bwd_res(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(scale), ms, _ptr(mean),
                                                 _ptr(rstd), _ptr(dres), _ptr(dx), _ptr(part), b * n, d, n, dt, self._stream(x)))
        per_b = nby // b
        dgamma = self._reduce_rows(part[0].contiguous(), nby, d)
        if scale is None:
            return dx, dgamma, None, None
        dscale = torch.stack([self._reduce_rows(part[1, i * per_b:(i + 1) * per_b].contiguous(), per_b, d) for i in range(b)])
        dshift = torch.stack([self._reduce_rows(part[2, i * per_b:(i + 1) * per_b].contiguous(), per_b, d) for i in range(b)])
        return dx, dgamma, dscale, dshift

    def rope_tables(self, inv_freq, n, pos_scale=1.0):
        self._f32(inv_freq)
        half = inv_freq.numel()
        cs = torch.empty(n, half, 2, dtype=torch.float32, device=inv_freq.device)
        self._chk(self.lib.sat_rope_tables(_ptr(inv_freq), _ptr(cs), n, half, float(pos_scale), self._stream(inv_freq)))
        return cs

    def rope_apply_(self, t, cs, transpose=False):
        """In place on t: (B, N, H, dh) view (dh contiguous); rotates the first 2*half dims of every head."""
        dt = self._dt(t)
        b, n, h, dh = t.shape
        half = cs.shape[1]
        if t.stride(3) != 1 or 2 * half > dh:
            raise ValueError("bad rotary layout")
        self._chk(self.lib.sat_rope_apply(_ptr(t), _ptr(cs), t.stride(0), t.stride(1), t.stride(2), b, n, h, half,
                                          cs.shape[0] - n, int(transpose), dt, self._stream(t)))
        return t

    def swiglu(self, xin):
        dt = self._dt(xin)
        f = xin.shape[-1] // 2
        rows = xin.numel() // (2 * f)
        out = torch.empty(*xin.shape[:-1], f, dtype=xin.dtype, device=xin.device)
        self._chk(self.lib.sat_swiglu(_ptr(xin), None, _ptr(out), rows, f, 0, dt, self._stream(xin)))
        return out

    def swiglu_bwd(self, xin, dout):
        dt = self._dt(xin, dout)
        f = xin.shape[-1] // 2
        rows = xin.numel() // (2 * f)
        dxin = torch.empty_like(xin)
        self._chk(self.lib.sat_swiglu(_ptr(xin), _ptr(dout), _ptr(dxin), rows, f, 1, dt, self._stream(xin)))
        return dxin

    def gate_residual(self, x, gate, res):
        """x * sigmoid(1 - gate[b]) + res;  x, res: (B, N, D) contiguous; gate: (B, D) view."""
        dt = self._dt(x, gate, res)
        b, n, d = x.shape
        y = torch.empty_like(x)
        self._chk(self.lib.sat_gate_residual(_ptr(x), _ptr(gate), gate.stride(0), _ptr(res), _ptr(y), b, n, d, dt,
                                             self._stream(x)))
        return y

    def gate_residual_bwd(self, dy, x, gate):
        """Returns dx (B, N, D) and dgate (B, D) fp32 for y = x * sigmoid(1 - gate[b]) + res (d_res = dy)."""
        dt = self._dt(dy, x, gate)
        b, n, d = x.shape
        dx = torch.empty_like(x)
        nch = self.lib.sat_gate_residual_bwd_nchunks(n)
        part = torch.empty(b, nch, d, dtype=torch.float32, device=x.device)
        self._chk(self.lib.sat_gate_residual_bwd(_ptr(dy), _ptr(x), _ptr(gate), gate.stride(0), _ptr(dx), _ptr(part), b, n, d,
                                                 dt, self._stream(x)))
        dgate = torch.stack([self._reduce_rows(part[i], nch, d) for i in range(b)])
        return dx, dgate

    def cfg_step(self, out2, ncond, cfg_scale=1.0, scale_phi=0.0, x=None, coef=None, want_second=False, prev=None):
        """Guidance combine (+rescale) of the batched model output and, with x/coef, the sampler update in the same pass:
        out2 (ncond*B, C, T); returns v, or (y0 [, y1]) with
            y0 = c0x*x + c0v*v + c0p*prev + c0u*u,   y1 = c1x*x + c1v*v + c1p*prev + c1u*u      (u = the unconditioned output)
        coef: 4 values (c0x, c0v, c1x, c1v) or 8 values (c0x c0v c0p c0u c1x c1v c1p c1u) — a host sequence, or a DEVICE fp32
        tensor of 8 (HIP-graph replay: the kernel reads it)."""
        dt = self._dt(out2, x)
        if not out2.is_contiguous() or (x is not None and not x.is_contiguous()) or (prev is not None and not prev.is_contiguous()):
            raise ValueError("cfg_step: contiguous tensors expected")
        nb2, c, t = out2.shape
        b = nb2 // ncond
        if prev is not None and (x is None or prev.shape != x.shape or prev.dtype != out2.dtype):
            raise ValueError("cfg_step: prev must match x")
        y0 = torch.empty(b, c, t, dtype=out2.dtype, device=out2.device)
        y1 = torch.empty_like(y0) if (want_second and x is not None) else None
        if isinstance(coef, torch.Tensor):        # device coefficients (HIP-graph replay): 8 fp32 values read by the kernel
            self._f32(coef)
            if coef.numel() != 8 or x is None:
                raise ValueError("cfg_step: device coef must hold 8 fp32 values (c0x c0v c0p c0u c1x c1v c1p c1u) and needs x")
            self._chk(self.lib.sat_sampler_step_dev(_ptr(out2), _ptr(x), _ptr(prev), _ptr(y0), _ptr(y1), b, c, t, ncond, float(cfg_scale),
                                                    float(scale_phi), _ptr(coef), dt, self._stream(out2)))
            return (y0, y1) if y1 is not None else y0
        coef = tuple(float(v) for v in coef) if coef is not None else (0.0, 1.0, 0.0, 0.0)
        if len(coef) == 4:
            coef = (coef[0], coef[1], 0.0, 0.0, coef[2], coef[3], 0.0, 0.0)
        if len(coef) != 8:
            raise ValueError("cfg_step: coef takes 4 or 8 values")
        if (coef[2] != 0.0 or coef[6] != 0.0) and prev is None:
            raise ValueError("cfg_step: a coefficient on `prev` without the tensor")
        host = (ctypes.c_float * 8)(*coef)
        self._chk(self.lib.sat_sampler_step(_ptr(out2), _ptr(x), _ptr(prev), _ptr(y0), _ptr(y1), b, c, t, ncond, float(cfg_scale),
                                            float(scale_phi), host, dt, self._stream(out2)))
        return (y0, y1) if y1 is not None else y0

    # ------------------------------------------------------------------ dense projections (csrc/gemm.hip)
    EPI_STORE, EPI_RES, EPI_GATE_RES, EPI_SWIGLU = 0, 1, 2, 3
    gemm_tile = None     # None: pick per shape (_pick_tile); 0 = 128x128 (4 waves, 2 workgroups per CU), 4 = 256x256, 7 = 160x256, 8 = 128x128 (8 waves)

    def _zeros_page(self, device):
        z = getattr(self, "_zpage", None)
        if z is None or z.device != device:
            z = _keep_zeros(64, torch.int16, device)
            self._zpage = z
        return z

    # Workgroup-tile model of csrc/gemm.hip, one entry per shipped tile: (rows, columns, workgroups per CU, us per 64-deep K-step of one
    # workgroup when <= 128 workgroups are on the chip, the same with the chip full, fixed us per workgroup: launch gap + first-tile
    # latency + epilogue).  A projection's time is rounds-of-the-chip x (K-steps x us + fixed).  Round 5: tiles 4, 7 and 8 re-fitted
    # (tools/fit_tile_model.py) to the specialised K loops that replaced round 4's (profiles/r05_experiments/lean_ab/gemm_lean.jsonl,
    # M = 260 .. 4100: median error 3.6 / 5.2 / 2.4 %, worst 20 %); tile 0 keeps round 4's fit (profiles/r04_gemm_bench.jsonl).
    _TILE_MODEL = {0: (128, 128, 2, 0.65, 1.00, 6.2), 4: (256, 256, 1, 1.18, 1.36, 8.0),
                   7: (160, 256, 1, 0.79, 0.94, 10.0), 8: (128, 128, 1, 0.53, 0.485, 4.75)}

    def _tile_cost(self, tile, m, n, k, splits=1):
        bm, bn, per_cu, u_light, u_full, f = self._TILE_MODEL[tile]
        tiles = -(-m // bm) * -(-n // bn) * splits
        rounds = -(-tiles // (256 * per_cu))
        # (tile 0: two workgroups of a CU share its matrix pipes once more than 256 are in flight)
        u = u_full if tiles > (256 if per_cu > 1 else 128) else u_light
        ksteps = -(-(-(-k // 64)) // splits)
        t = rounds * (u * ksteps + f)
        if splits > 1:
            t += 4.0 + splits * m * n * 4 / 4e6      # sat_splitk_epilogue: launch + the slabs read back at ~4 TB/s
        return t

    def _pick_tile(self, m, n, splits=1, k=None):
        """Workgroup tile of a projection.  Round 4: the cheapest of the shipped tiles under _TILE_MODEL — 256 x 256 (sat_gemm256_kernel)
        for the many-tile shapes, 160 x 256 where 256-row tiles leave CUs idle (QKV at M = 2050: 144 full tiles -> 234 workgroups),
        128 x 128 on eight waves for the 1536 -> 1536 projections (sat_gemm8_kernel), the four-wave 128 x 128 kernel for the rest."""
        if self.gemm_tile is not None:
            return self.gemm_tile
        if k is None:
            return 4 if splits == 1 and ((m + 255) // 256) * ((n + 255) // 256) >= 150 else 0
        return min(self._TILE_MODEL, key=lambda t: (self._tile_cost(t, m, n, k, splits), t))

    def gemm_bf16(self, a, b, bias=None, res=None, gate=None, rows_per_gate=0, epilogue=0, out_dtype=torch.bfloat16, want_pre=False,
                  splits=1, out=None):
        """C = epilogue(A · B^T): a (M, K), b (N, K) bf16 (row strides free, K contiguous), fp32 accumulation.
        epilogue 0: [+bias]; 1: + res; 2: * sigmoid(1 - gate[m // rows_per_gate]) + res; 3: SwiGLU over b = [value rows | gate rows]
        (returns (M, N/2), and the (M, N) pre-activation too with want_pre).  bias fp32 (N,); res / gate in out_dtype.
        splits > 1: split-K slabs (splits, M, N) fp32, summed here with sat_reduce_splits."""
        if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16:
            raise TypeError("gemm_bf16 takes bf16 operands (fp32 models go through split_bf16x3)")
        if a.stride(1) != 1 or b.stride(1) != 1 or a.shape[1] != b.shape[1]:
            raise ValueError("gemm_bf16: operands must be (rows, K) with K contiguous and equal")
        if not self.simulator and not a.is_cuda:
            raise RuntimeError("stable_audio_tools_amd kernels need CUDA(HIP) tensors; there is no CPU path")
        m, k = a.shape
        n = b.shape[0]
        f32 = out_dtype == torch.float32
        if out_dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("gemm_bf16: output is bf16 or fp32")
        if bias is not None:
            self._f32(bias)
        for t in (res, gate):
            if t is not None and (t.dtype != out_dtype or t.stride(-1) != 1):
                raise TypeError("gemm_bf16: res / gate must have the output dtype and a contiguous last dim")
        nout = n // 2 if epilogue == self.EPI_SWIGLU else n
        if out is not None and splits == 1:
            if tuple(out.shape) != (m, nout) or out.dtype != out_dtype or not out.is_contiguous():
                raise ValueError("gemm_bf16: bad `out`")
            c = out
        else:
            c = torch.empty((splits, m, nout) if splits > 1 else (m, nout), dtype=out_dtype, device=a.device)
        pre = torch.empty(m, n, dtype=out_dtype, device=a.device) if (want_pre and epilogue == self.EPI_SWIGLU) else None
        self._chk(self.lib.sat_gemm_bf16(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(c), nout, _ptr(bias),
                                         _ptr(res), res.stride(0) if res is not None else 0,
                                         _ptr(gate), gate.stride(0) if gate is not None else 0, rows_per_gate,
                                         _ptr(pre), n, _ptr(self._zeros_page(a.device)), m, n, k, epilogue, int(f32), splits,
                                         self._pick_tile(m, n, splits, k), self._stream(a)))
        if splits > 1:
            c = self._reduce_rows(c.view(splits, m * nout), splits, m * nout).view(m, nout)
        return (c, pre) if want_pre and epilogue == self.EPI_SWIGLU else c

    def _plane_cache(self, key, shape, device):
        """Persistent zero-initialised bf16 planes for the no-grad path: rows / columns past the sequence length are never
        written by the projection epilogue, so one memset at first use keeps them zero for every later call."""
        cache = self.__dict__.setdefault("_planes", {})
        t = cache.get(key)
        if t is None or t.shape != shape or t.device != device:
            t = _keep_zeros(shape, torch.int16, device)
            cache[key] = t
        return t

    def gemm_bf16_splitk(self, a, b, splits, bias=None, res=None, out_dtype=torch.bfloat16, out=None):
        """gemm_bf16 (epilogue: [+bias] [+res]) with K cut into `splits` ranges: fp32 slabs from the GEMM kernel, summed with the
        bias / residual by sat_splitk_epilogue.  For few-tile / long-K projections (more workgroups than tiles)."""
        m, n = a.shape[0], b.shape[0]
        st = self._stream(a)
        key = ("splitk", splits, m, n, a.device, st.value if st is not None else 0)      # one slab set per (device, stream)
        slabs = self.__dict__.setdefault("_planes", {}).get(key)
        if slabs is None:
            slabs = _keep_empty((splits, m, n), torch.float32, a.device)
            self._planes[key] = slabs
        self._chk(self.lib.sat_gemm_bf16(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(slabs), n, None, None, 0, None, 0, 0, None, 0,
                                         _ptr(self._zeros_page(a.device)), m, n, a.shape[1], 0, 1, splits,
                                         self._pick_tile(m, n, splits, a.shape[1]), self._stream(a)))
        c = out if out is not None else torch.empty(m, n, dtype=out_dtype, device=a.device)
        if bias is not None:
            self._f32(bias)
        self._chk(self.lib.sat_splitk_epilogue(_ptr(slabs), splits, _ptr(bias), _ptr(res), res.stride(0) if res is not None else 0, _ptr(c), n,
                                               m, n, int(out_dtype == torch.float32), self._stream(a)))
        return c

    def splitk_for(self, m, n, k):
        """Split count for a projection: > 1 only for few-tile / long-K shapes (FF2: 6144 -> 1536), where cutting K puts more workgroups
        on the chip and the slab round trip (fp32, M x N x 4 B per slice) is cheap next to the saving — the cheapest (tile, splits) pair
        under _TILE_MODEL."""
        if self.gemm_splitk is not None:
            return self.gemm_splitk
        if k < 4096 or n % 4:
            return 1
        if self.gemm_tile is not None:
            return 1 if ((m + 127) // 128) * ((n + 127) // 128) > 256 else 2
        return min((1, 2, 3, 4), key=lambda sp: (min(self._tile_cost(t, m, n, k, sp) for t in self._TILE_MODEL), sp))

    gemm_splitk = None

    def gemm_heads_bf16(self, x, w, cs, heads, nb, ntok, sec0, nsec, reuse=None):
        """Attention input projection with head split / rotary / plane layout fused (sat_gemm_qkv_bf16): x (nb*ntok, K) bf16,
        w (nsec*heads*64, K) bf16, cs (>= ntok, 16, 2) fp32 rotary table or None.  Sections sec0 .. sec0+nsec-1 of (q, k, v).
        Returns dict(q=, k= (nb,H,Np,64), v_tr= (nb,H,64,Np)) with the produced planes (bf16 bits as int16, zero padded).
        reuse: a hashable tag -> the planes live in a per-ops cache (only safe when nothing keeps them for a backward)."""
        if x.dtype != torch.bfloat16 or w.dtype != torch.bfloat16:
            raise TypeError("gemm_heads_bf16 takes bf16 operands")
        if x.stride(1) != 1 or w.stride(1) != 1 or w.shape[0] != nsec * heads * 64 or x.shape[0] != nb * ntok:
            raise ValueError("gemm_heads_bf16: bad operand layout")
        npad = (ntok + 63) // 64 * 64
        out = {"n": ntok, "np": npad}
        for i, (name, shape) in enumerate((("q", (nb, heads, npad, 64)), ("k", (nb, heads, npad, 64)), ("v_tr", (nb, heads, 64, npad)))):
            if sec0 <= i < sec0 + nsec:
                out[name] = (self._plane_cache((reuse, name, shape), shape, x.device) if reuse is not None
                             else torch.zeros(shape, dtype=torch.int16, device=x.device))
        if cs is not None:
            self._f32(cs)
        self._chk(self.lib.sat_gemm_qkv_bf16(_ptr(x), x.stride(0), _ptr(w), w.stride(0), _ptr(cs), (cs.shape[0] - ntok) if cs is not None else 0,
                                             _ptr(out.get("q")), _ptr(out.get("k")), _ptr(out.get("v_tr")), _ptr(self._zeros_page(x.device)),
                                             nb, ntok, npad, heads, x.shape[1], sec0, nsec, self._pick_tile(nb * ntok, nsec * heads * 64, 1, x.shape[1]),
                                             self._stream(x)))
        return out

    def attention_planes(self, q_rm, k_rm, v_tr, nq, nk, scale, out_dtype=torch.bfloat16):
        """Attention forward straight from bf16 operand planes (gemm_heads_bf16): q_rm (B,H,Npq,64), k_rm (B,Hkv,Npk,64),
        v_tr (B,Hkv,64,Npk) -> o (B, Nq, H*64)."""
        b, h, npq, d = q_rm.shape
        hk, npk = k_rm.shape[1], k_rm.shape[2]
        o = torch.empty(b, nq, h * d, dtype=out_dtype, device=q_rm.device)
        if out_dtype != torch.bfloat16:
            raise TypeError("attention_planes produces bf16")
        if self.cross_ok(h, hk, nk, d, 1):
            self._chk(self.lib.sat_attention_cross_fwd(_ptr(q_rm), _ptr(k_rm), _ptr(v_tr), _ptr(o), None, b, h, hk, nq, nk, npq, npk, d,
                                                       float(scale), self._stream(q_rm)))
        else:
            self._chk(self.lib.sat_attention_fwd(_ptr(q_rm), None, _ptr(k_rm), None, _ptr(v_tr), None, _ptr(o), None, b, h, hk, nq, nk,
                                                 npq, npk, d, float(scale), self._fwd_dtype(1), self._stream(q_rm)))
        return o

    # ---- fp8 (e4m3) forward projections: per-tensor dynamic scaling, MX MFMA with unit block scales ----
    def quant_fp8(self, src):
        """src (R, C) fp32|bf16 -> (q (R, C) uint8 e4m3 bits, dequant scale = amax / 448 as a 0-dim device tensor)."""
        dt = self._dt(src)
        if src.dim() != 2 or src.stride(1) != 1 or src.shape[1] % 4:
            raise ValueError("quant_fp8 takes a 2-D tensor with a contiguous last dim, C % 4 == 0")
        # two launches: max|src| -> (448 / amax, amax / 448) on the device (sat_absmax_scale), then the conversion itself
        st = self._stream(src)
        wkey = ("absmax", src.device, st.value if st is not None else 0)
        work = self.__dict__.setdefault("_planes", {}).get(wkey)
        if work is None:
            work = _keep_zeros(1 + 1024, torch.float32, src.device)
            self._planes[wkey] = work
        scales = torch.empty(2, dtype=torch.float32, device=src.device)
        self._chk(self.lib.sat_absmax_scale(_ptr(src), src.stride(0), _ptr(work), _ptr(scales), src.shape[0], src.shape[1], int(dt == 0), st))
        q = torch.empty(src.shape, dtype=torch.uint8, device=src.device)
        self._chk(self.lib.sat_quant_fp8(_ptr(src), src.stride(0), _ptr(q), q.stride(0), _ptr(scales), src.shape[0], src.shape[1], int(dt == 0), st))
        return q, scales[1]

    def quant_fp8_rows(self, src):
        """src (R, C) fp32|bf16 -> (q (R, C) uint8 e4m3 bits, scale (R,) fp32 = row max / 448): per-ROW dynamic quantisation in one
        pass over the activation (sat_quant_fp8_rows); the scales go to gemm_fp8 / gemm_heads_fp8 as `row_alpha`."""
        dt = self._dt(src)
        if src.dim() != 2 or src.stride(1) != 1 or src.shape[1] % 8 or src.stride(0) % 8 or src.shape[1] > 8192:
            raise ValueError("quant_fp8_rows takes a 2-D tensor with a contiguous last dim, C % 8 == 0, C <= 8192")
        q = torch.empty(src.shape, dtype=torch.uint8, device=src.device)
        scale = torch.empty(src.shape[0], dtype=torch.float32, device=src.device)
        self._chk(self.lib.sat_quant_fp8_rows(_ptr(src), src.stride(0), _ptr(q), q.stride(0), _ptr(scale), src.shape[0], src.shape[1],
                                              int(dt == 0), self._stream(src)))
        return q, scale

    gemm_fp8_tile = None     # None: pick per shape (the fp8 instances: 0, 4, 7, 8)

    def _pick_tile_fp8(self, m, n, k):
        """As _pick_tile for the fp8 kernels (K-steps of 128 fp8 values: half as many per projection)."""
        if self.gemm_fp8_tile is not None:
            return self.gemm_fp8_tile
        return min(self._TILE_MODEL, key=lambda t: (self._tile_cost(t, m, n, (k + 1) // 2), t))

    def gemm_fp8(self, a, b, alpha, bias=None, res=None, gate=None, rows_per_gate=0, epilogue=0, out_dtype=torch.bfloat16, want_pre=False,
                 out=None, row_alpha=None, col_alpha=None):
        """gemm_bf16 on fp8 operands: a (M, K), b (N, K) uint8 (quant_fp8), alpha = 0-dim fp32 device tensor (scale_a * scale_b);
        row_alpha (M,) fp32: a was quantised row by row (quant_fp8_rows) — alpha is then b's scale alone; col_alpha (N,) fp32: b was
        quantised row by row (one scale per output channel) — alpha is then a's per-tensor scale, or None with row_alpha."""
        if a.dtype != torch.uint8 or b.dtype != torch.uint8 or a.stride(1) != 1 or b.stride(1) != 1 or a.shape[1] != b.shape[1]:
            raise TypeError("gemm_fp8 takes (rows, K) uint8 operands")
        m, k = a.shape
        n = b.shape[0]
        f32 = out_dtype == torch.float32
        if bias is not None:
            self._f32(bias)
        nout = n // 2 if epilogue == self.EPI_SWIGLU else n
        c = out if out is not None else torch.empty(m, nout, dtype=out_dtype, device=a.device)
        pre = torch.empty(m, n, dtype=out_dtype, device=a.device) if (want_pre and epilogue == self.EPI_SWIGLU) else None
        alpha, col_alpha = self._fp8_alphas(alpha, col_alpha, n)
        if row_alpha is not None and (row_alpha.dtype != torch.float32 or row_alpha.numel() != m or not row_alpha.is_contiguous()):
            raise TypeError("gemm_fp8: row_alpha is (M,) fp32")
        self._chk(self.lib.sat_gemm_fp8(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(c), nout, _ptr(bias),
                                        _ptr(res), res.stride(0) if res is not None else 0, _ptr(gate), gate.stride(0) if gate is not None else 0,
                                        rows_per_gate, _ptr(pre), n, _ptr(self._zeros_page(a.device)), _ptr(alpha), _ptr(row_alpha), _ptr(col_alpha),
                                        m, n, k, epilogue, int(f32), self._pick_tile_fp8(m, n, k), self._stream(a)))
        return (c, pre) if want_pre and epilogue == self.EPI_SWIGLU else c

    @staticmethod
    def _fp8_alphas(alpha, col_alpha, n):
        if alpha is None and col_alpha is None:
            raise TypeError("fp8 GEMM: alpha (per-tensor de-quantisation scale) or col_alpha (per output channel) is required")
        if alpha is not None:
            alpha = alpha.float().reshape(1).contiguous()
        if col_alpha is not None and (col_alpha.dtype != torch.float32 or col_alpha.numel() != n or not col_alpha.is_contiguous()):
            raise TypeError("fp8 GEMM: col_alpha is (N,) fp32, one factor per row of b")
        return alpha, col_alpha

    def gemm_heads_fp8(self, x, w, alpha, cs, heads, nb, ntok, sec0, nsec, reuse=None, row_alpha=None, col_alpha=None):
        """gemm_heads_bf16 on fp8 operands (planes come out in bf16); row_alpha / col_alpha as gemm_fp8."""
        npad = (ntok + 63) // 64 * 64
        out = {"n": ntok, "np": npad}
        for i, (name, shape) in enumerate((("q", (nb, heads, npad, 64)), ("k", (nb, heads, npad, 64)), ("v_tr", (nb, heads, 64, npad)))):
            if sec0 <= i < sec0 + nsec:
                out[name] = (self._plane_cache((reuse, name, shape), shape, x.device) if reuse is not None
                             else torch.zeros(shape, dtype=torch.int16, device=x.device))
        alpha, col_alpha = self._fp8_alphas(alpha, col_alpha, w.shape[0])
        self._chk(self.lib.sat_gemm_qkv_fp8(_ptr(x), x.stride(0), _ptr(w), w.stride(0), _ptr(cs), (cs.shape[0] - ntok) if cs is not None else 0,
                                            _ptr(out.get("q")), _ptr(out.get("k")), _ptr(out.get("v_tr")), _ptr(self._zeros_page(x.device)),
                                            _ptr(alpha), _ptr(row_alpha), _ptr(col_alpha), nb, ntok, npad, heads, x.shape[1], sec0, nsec,
                                            self._pick_tile_fp8(nb * ntok, nsec * heads * 64, x.shape[1]), self._stream(x)))
        return out

    def cast_bf16(self, src, transpose=False, row_pad=1, out=None):
        """src (R, C) fp32|bf16 (last dim contiguous) -> bf16 (R, C), or transposed (C, Rp) with Rp = R rounded up to row_pad
        (extra columns zero).  out: optional destination of that shape (row stride free)."""
        dt = self._dt(src)
        if src.dim() != 2 or src.stride(1) != 1:
            raise ValueError("cast_bf16 takes a 2-D tensor with a contiguous last dim")
        r, c = src.shape
        rp = (r + row_pad - 1) // row_pad * row_pad
        shape = (c, rp) if transpose else (r, c)
        dst = out if out is not None else torch.empty(shape, dtype=torch.bfloat16, device=src.device)
        if tuple(dst.shape) != shape or dst.dtype != torch.bfloat16 or dst.stride(1) != 1:
            raise ValueError("cast_bf16: bad destination")
        self._chk(self.lib.sat_cast_bf16(_ptr(src), src.stride(0), _ptr(dst), dst.stride(0), r, c, rp, int(dt == 0), int(transpose),
                                         self._stream(src)))
        return dst

    def cast_bf16_tpair(self, a, b, row_pad=8, out_b=None):
        """(a^T, b^T) as cast_bf16(transpose=True, row_pad=...) of two 2-D tensors, in ONE launch (sat_cast_bf16_tpair).  out_b: optional
        destination for b^T (row stride free)."""
        da, db = self._dt(a), self._dt(b)
        if a.dim() != 2 or b.dim() != 2 or a.stride(1) != 1 or b.stride(1) != 1:
            raise ValueError("cast_bf16_tpair takes 2-D tensors with a contiguous last dim")
        (ra, ca), (rb, cb) = a.shape, b.shape
        rpa, rpb = (ra + row_pad - 1) // row_pad * row_pad, (rb + row_pad - 1) // row_pad * row_pad
        ta = torch.empty(ca, rpa, dtype=torch.bfloat16, device=a.device)
        tb = out_b if out_b is not None else torch.empty(cb, rpb, dtype=torch.bfloat16, device=b.device)
        if tuple(tb.shape) != (cb, rpb) or tb.dtype != torch.bfloat16 or tb.stride(1) != 1:
            raise ValueError("cast_bf16_tpair: bad destination")
        self._chk(self.lib.sat_cast_bf16_tpair(_ptr(a), a.stride(0), _ptr(ta), ta.stride(0), ra, ca, rpa, int(da == 0),
                                               _ptr(b), b.stride(0), _ptr(tb), tb.stride(0), rb, cb, rpb, int(db == 0), self._stream(a)))
        return ta, tb

    def cast_bf16_dual(self, src, row_pad=8):
        """src (R, C) fp32|bf16 -> (bf16 (R, C), bf16 transposed (C, Rp)) in one pass (sat_cast_bf16_dual), or None when the shape is outside
        its 16-byte path (C % 8, alignment): the caller then casts twice."""
        dt = self._dt(src)
        if src.dim() != 2 or src.stride(1) != 1:
            raise ValueError("cast_bf16_dual takes a 2-D tensor with a contiguous last dim")
        r, c = src.shape
        rp = (r + row_pad - 1) // row_pad * row_pad
        if c % 8 or rp % 8 or src.stride(0) % (4 if dt == 0 else 8) or src.data_ptr() % 16:
            return None
        dst = torch.empty(r, c, dtype=torch.bfloat16, device=src.device)
        dst_t = torch.empty(c, rp, dtype=torch.bfloat16, device=src.device)
        self._chk(self.lib.sat_cast_bf16_dual(_ptr(src), src.stride(0), _ptr(dst), dst.stride(0), _ptr(dst_t), dst_t.stride(0), r, c, rp,
                                              int(dt == 0), self._stream(src)))
        return dst, dst_t

    def split_bf16x3(self, src, side):
        """fp32 (R, C) -> bf16 (R, 3C): side 0 (activations) [hi|hi|lo], side 1 (weights) [hi|lo|hi]."""
        self._f32(src)
        r, c = src.shape
        dst = torch.empty(r, 3 * c, dtype=torch.bfloat16, device=src.device)
        self._chk(self.lib.sat_split_bf16x3(_ptr(src), src.stride(0), _ptr(dst), 3 * c, r, c, side, self._stream(src)))
        return dst

    # ------------------------------------------------------------------ optimizer
    def adamw_step_dev(self, p, g, m, v, hyper, beta1, beta2, eps, weight_decay, ema=None):
        """adamw_step with (lr, 1 - beta1^t, sqrt(1 - beta2^t), grad_scale, ema_decay) read from the device tensor `hyper` (5 fp32)."""
        self._f32(p, g, m, v, ema, hyper)
        if hyper.numel() < 5:
            raise ValueError("adamw_step_dev: hyper holds 5 fp32 values")
        self._chk(self.lib.sat_adamw_step_dev(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(hyper), beta1, beta2, eps, weight_decay,
                                              _ptr(ema), self._stream(p)))

    # ---- native gradient exchange (csrc/comm.hip: RCCL behind the C-ABI; training.GradAllReduce(native=True)) ----
    def allreduce_available(self):
        return bool(self.lib.sat_allreduce_available())

    def allreduce_unique_id(self):
        """128 bytes from rank 0's RCCL (ncclGetUniqueId) for every rank's allreduce_init."""
        buf = ctypes.create_string_buffer(128)
        self._chk(self.lib.sat_allreduce_unique_id(buf))
        return buf.raw

    def allreduce_init(self, uid, world, rank):
        """Join the communicator (the current device is this rank's GPU); returns the opaque handle."""
        if len(uid) != 128:
            raise ValueError("allreduce_init: the unique id is 128 bytes")
        comm = ctypes.c_void_p()
        self._chk(self.lib.sat_allreduce_init(ctypes.create_string_buffer(bytes(uid), 128), int(world), int(rank), ctypes.byref(comm)))
        return comm

    def allreduce_bucket(self, comm, buf, mode=0):
        """In-place SUM over the ranks of a contiguous fp32 / bf16 device tensor on its current stream (mode 1: reduce-scatter + all-gather)."""
        if buf.dtype not in (torch.float32, torch.bfloat16) or not buf.is_contiguous():
            raise TypeError("allreduce_bucket takes a contiguous fp32 or bf16 tensor")
        if not self.simulator and not buf.is_cuda:
            raise RuntimeError("stable_audio_tools_amd kernels need CUDA(HIP) tensors; there is no CPU path")
        self._chk(self.lib.sat_allreduce_bucket(comm, _ptr(buf), buf.numel(), 0 if buf.dtype == torch.float32 else 1, int(mode), self._stream(buf)))

    def allreduce_finalize(self, comm):
        self._chk(self.lib.sat_allreduce_finalize(comm))

    def multi_copy(self, srcs, dsts):
        """dsts[i] <- srcs[i] (contiguous fp32 tensors of equal sizes, pairwise) in ceil(n / 160) launches (sat_multi_copy: the table
        rides in the kernel arguments — nothing to stage, capturable in a HIP graph)."""
        import numpy as np
        n = len(srcs)
        if n == 0:
            return
        tab = np.empty((n, 3), dtype=np.int64)
        for i, (src, dst) in enumerate(zip(srcs, dsts)):
            if src.dtype != torch.float32 or dst.dtype != torch.float32 or not src.is_contiguous() or not dst.is_contiguous() \
                    or src.numel() != dst.numel() or src.device != dst.device:
                raise ValueError("multi_copy: sources and destinations must be contiguous fp32 tensors of equal sizes on one device")
            tab[i] = (src.data_ptr(), dst.data_ptr(), src.numel())
        self._chk(self.lib.sat_multi_copy(tab.ctypes.data, n, self._stream(srcs[0])))

    def adamw_step(self, p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, ema=None, ema_decay=0.0):
        self._f32(p, g, m, v, ema)
        self._chk(self.lib.sat_adamw_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), lr, beta1, beta2, eps,
                                          weight_decay, step, grad_scale, _ptr(ema), ema_decay, self._stream(p)))


_ops = None


def get_ops():
    """The product singleton: bound to the gfx950 library, or raises."""
    global _ops
    if _ops is None:
        _ops = SatOps(_lib.load())
    return _ops
