/* sat_amd.h — C-ABI of libsat_amd.so, the MI355X (gfx950) implementation of stable-audio-tools'
 * denoising hot path.
 *
 * The reference (Stability-AI/stable-audio-tools) is pure Python: it has no FFI, and its boundary
 * for this path is the nn.Module API.  This header is therefore the boundary a
 * maintainer binds ONCE (ctypes; stable_audio_tools_amd/_lib.py is that binding)
 * underneath module shims that keep the reference's class names and state_dict keys.  Each entry
 * point cites the reference code whose device work it replaces (paths relative to
 * stable_audio_tools/ in the reference tree).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) owned by the caller (the PyTorch caching allocator),
 *     including workspaces and partial-sum buffers; the library allocates nothing and keeps no state
 *     besides a thread-local last-error string;
 *   - tensors are fp32, contiguous, (B, C, T) with T fastest unless stated otherwise;
 *   - `stream` is a hipStream_t; every call is asynchronous and stream-ordered, no hidden syncs;
 *   - return value: 0 = ok, non-zero = error (sat_last_error() describes it); nothing throws;
 *   - re-entrant; one process per GPU.
 */
#ifndef SAT_AMD_H
#define SAT_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

int sat_abi_version(void);
int sat_is_simulator(void);          /* 0 for the gfx950 library; 1 only for the CPU test-suite's simulator build */
const char* sat_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Oobleck conv stack — models/autoencoders.py:23-27 (WNConv1d / WNConvTranspose1d), :58-83
 * (ResidualUnit), :233-283 (Encoder/DecoderBlock), :285-362 (OobleckEncoder/Decoder);
 * SnakeBeta models/blocks.py:291-329.  Replaces F.conv1d / F.conv_transpose1d / snake_beta and
 * their autograd.
 * ---------------------------------------------------------------------------------------------- */

/* y = [tanh]( conv1d(snake(x; alpha, beta), W, stride, dil, pad) + bias + res )
 * or, when x2 != NULL (backward epilogue of the conv that consumed snake(x2)):
 *   y = conv1d(x, W) * dsnake(x2)/dx2 + res, and per-tile partial sums of dL/dlog-alpha2,
 *   dL/dlog-beta2 in part_da / part_db ([Cout][sat_conv1d_partial_rows(B,Tout)], reduce with
 *   sat_rowsum).
 * w_packed: [Cin][K][Cout]  (sat_pack_weights mode 0; mode 1 of the forward weight for the
 * stride-1 data-gradient).  alpha/beta, bias, res may be NULL.  stride > 1 requires dil == 1. */
int sat_conv1d(const float* x, const float* w_packed, const float* bias, const float* alpha, const float* beta,
               const float* res, float* y, const float* x2, const float* alpha2, const float* beta2,
               float* part_da, float* part_db, int B, int Cin, int Cout, int Tin, int Tout, int K, int stride,
               int dil, int pad, int tanh_out, void* stream);
int sat_conv1d_partial_rows(int B, int Tout);

/* The convolutions of the stack on the bf16 matrix cores at fp32 accuracy (operands split hi+lo, three MFMAs per product).
 * Every conv is run as a stride-1 implicit GEMM over virtual channels (csrc/conv1d_bf16x3.hip):
 *   sat_conv1d_bf16x3:   stride 1 with K <= 8 (the k7 / k1 convs of every ResidualUnit, autoencoders.py:58-83, and their
 *                        data-gradients), or stride S = power of two with K == 2S (the down-convs, :245-247, and the
 *                        data-gradient of the up-convs) — space-to-depth on the input.
 *   sat_convtr1d_bf16x3: conv_transpose1d, K == 2S, S a power of two (the up-convs, :266-268, and the data-gradient of
 *                        the down-convs) — depth-to-space on the output.
 * Weights come from sat_pack_weights_bf16x3(w[D0][D1][K], stride, mode): mode 0 = conv weight [out][in][K];
 * mode 1 = data-gradient of a stride-1 conv (flipped/transposed); mode 2 = transposed-conv weight [in][out][K].
 * sat_pack_weights_bf16x3_size gives the plane length in elements (-1: unsupported).  snake_a / snake_ib are the
 * pre-exponentiated SnakeBeta constants from sat_snake_consts (a = e^alpha, ib = 1/(e^beta + 1e-9)) or NULL.
 * Everything else (epilogue options, partial-sum layout) as sat_conv1d / sat_convtr1d; partial rows are
 * sat_conv1d_bf16x3_partial_rows(B, Tout, K, stride) resp. sat_convtr1d_bf16x3_partial_rows(B, Tout, stride, pad). */
int sat_conv1d_bf16x3(const float* x, const short* w_hi, const short* w_lo, const float* bias, const float* snake_a,
                      const float* snake_ib, const float* res, float* y, const float* x2, const float* alpha2,
                      const float* beta2, float* part_da, float* part_db, int B, int Cin, int Cout, int Tin, int Tout,
                      int K, int stride, int dil, int pad, int tanh_out, void* stream);
/* The same conv that ALSO writes act(y) as the activation planes its consumer reads (sat_conv1d_bf16x3_planesq): em_hi /
 * em_lo [B][ceil(Cout/8)][em_rows][8] bf16, row 32 + t (the caller keeps the rows around the sequence zero); em_a / em_ib = the
 * consumer's pre-exponentiated SnakeBeta constants (sat_snake_consts) or NULL for planes of y itself.  Replaces the consumer's
 * sat_conv1d_k7_planes pre-pass (ResidualUnit chains: autoencoders.py:58-83, :233-283).  K <= 4 or strided plans, Tout % 4 == 0. */
int sat_conv1d_bf16x3_emit(const float* x, const short* w_hi, const short* w_lo, const float* bias, const float* snake_a,
                      const float* snake_ib, const float* res, float* y, const float* x2, const float* alpha2,
                      const float* beta2, float* part_da, float* part_db, int B, int Cin, int Cout, int Tin, int Tout,
                      int K, int stride, int dil, int pad, int tanh_out, void* em_hi, void* em_lo,
                           const float* em_a, const float* em_ib, int em_rows, void* stream);
/* The whole ResidualUnit forward in ONE launch (models/autoencoders.py:58-83), 1 <= C <= 128 (csrc/conv1d_bf16x3_k7q.h, FUSED):
 *   h = conv7_dil(snake1(x)) + bias1   -> `h` (fp32 (B, C, T); NULL: not kept — the backward needs it)
 *   y = x + conv1(snake2(h)) + bias2   -> `y`, and (em_hi != NULL) snake_next(y) as the next unit's activation planes.
 * xp_hi / xp_lo [B][ceil(C/8)][rows][8]: planes of snake1(x) (sat_conv1d_k7_planes or a producer's emission); w7_*:
 * sat_pack_weights_k7q(K, mode 0); w1_*: sat_pack_weights_k7q of the (C, C, 1) weight (K = 1); a2 / ib2: sat_snake_consts of snake2.
 * 'same' padding (2 * pad == (K - 1) * dil), T % 4 == 0. */
int sat_residual_unit_fwd(const short* xp_hi, const short* xp_lo, int rows, const short* w7_hi, const short* w7_lo,
                          const float* bias1, const float* a2, const float* ib2, const short* w1_hi, const short* w1_lo,
                          const float* bias2, const float* x, float* h, float* y, int B, int C, int T, int K, int dil, int pad,
                          void* em_hi, void* em_lo, const float* em_a, const float* em_ib, int em_rows, void* stream);
int sat_convtr1d_bf16x3(const float* x, const short* w_hi, const short* w_lo, const float* bias, const float* snake_a,
                        const float* snake_ib, const float* res, float* y, const float* x2, const float* alpha2,
                        const float* beta2, float* part_da, float* part_db, int B, int Cin, int Cout, int Tin, int Tout,
                        int K, int stride, int pad, int tanh_out, void* stream);
int sat_conv1d_bf16x3_partial_rows(int B, int Tout, int K, int stride);

/* Row packing for the Conv2d layers of the MS-STFT discriminator (models/encodec.py:37-106) run as 1-D convs over virtual channels:
 * sat_rows_pack builds buf[lead + ((b*C*kh + c*kh + kt)*T + t)*pitch + pad_w + w] = x[b][c][t + kt*dil_t - pad_t][w] (zeros elsewhere,
 * incl. `lead` floats before and after; pitch % 4 == 0, lead % 4 == 0), sat_rows_pack_bwd is its adjoint; sat_rows_unpack takes a conv
 * output (B, C, T*pitch) back to (B, C, T, W) with LeakyReLU(slope) (slope 1: copy), sat_rows_unpack_bwd is its adjoint (`out` = the
 * activated output, or NULL for slope 1). */
int sat_rows_pack(const float* x, float* buf, int B, int C, int T, int W, int kh, int dil_t, int pad_t, int pad_w, int pitch, int lead,
                  void* stream);
int sat_rows_pack_bwd(const float* dbuf, float* dx, int B, int C, int T, int W, int kh, int dil_t, int pad_t, int pad_w, int pitch,
                      int lead, void* stream);
int sat_rows_unpack(const float* y, float* out, int B, int C, int T, int W, int pad_w, int pitch, float slope, void* stream);
int sat_rows_unpack_bwd(const float* dout, const float* out, float* dy, int B, int C, int T, int W, int pad_w, int pitch, float slope,
                        void* stream);

/* The Conv2d layers of the MS-STFT discriminator (models/encodec.py:19-28 NormConv2d, :37-106 DiscriminatorSTFT: kernel (3, 9) /
 * dilated (3, 9) / (3, 3), stride (1, 1), 'same' padding, LeakyReLU 0.2) on the "pitched rows" layout — csrc/disc_conv.hip:
 * a (B, C, frames, W) activation is kept as (B, C, L), frame r = [4 zeros | W samples | >= 4 zeros] of pitch P; sat_disc_geom gives
 * P, L = frames * P and the geometry of the bf16 hi / lo planes [B][ceil(C/8)][rows][8] (`lead` zero rows before position 0) that the
 * matrix kernels read.  The frame taps are virtual channels read from the same buffer dil_t * P positions away: nothing is copied.
 *   sat_disc_planes   src ((B, C, frames, W), or pitched (B, C, L) when pitched != 0), optionally times LeakyReLU'(out) (out pitched,
 *                     slope), pad positions zeroed -> dst (pitched fp32, or NULL) and hi / lo planes (or NULL); fm_sign / fm_coef (or
 *                     NULL): + fm_coef[0] * fm_sign before the LeakyReLU' factor (the L1 feature-matching term of `out`,
 *                     models/discriminators.py:52-56; fm_sign = sign(out - other signal's feature map) as int8, fm_coef a DEVICE
 *                     scalar).  sat_disc_l1_sum: sat_disc_l1_blocks() partial sums of |a - b| (the feature-matching distance on
 *                     pitched buffers) and, optionally, sign(a - b) as int8.
 *   sat_disc_pack_weights  w (Cout, Cin, kh, kw) -> wq, the packed bf16 hi + lo weights (sat_disc_pack_size elements); mode 0: the conv,
 *                     mode 1: its data-gradient (a conv of Cout channels -> Cin channels)
 *   sat_disc_conv     y = LeakyReLU_slope(conv2d + bias) (slope 1: none), pitched fp32 (B, Cout, L), pads zero; em_hi / em_lo (or
 *                     NULL): y's planes for the next layer.  Cin / Cout are the channels of the conv RUN (swapped for mode 1 weights).
 *                     lk_src (or NULL): y *= LeakyReLU'(lk_src) (slope lk_slope): a data-gradient leaving as the previous layer's
 *                     dL/d(pre-activation).
 *   sat_disc_wgrad    dW slabs [nsplit][kw][ceil64(M)][ceil64(kh * Cin)] (virtual channel tap_t * Cin + c) from dy = dL/d(pre-
 *                     activation) (B, M, L) and the layer input x (B, Cin, L), both pitched; sum with sat_reduce_splits.  kw 9 or 3. */
int sat_disc_geom(int frames, int W, int* P, int* L, int* lead, int* rows);
int sat_disc_planes(const float* src, const float* out, const signed char* fm_sign, const float* fm_coef, float* dst, void* hi, void* lo, int B,
                    int C, int frames, int W, int pitched, float slope, void* stream);
int sat_disc_l1_blocks(void);
int sat_disc_l1_sum(const float* a, const float* b, float* partial, signed char* sign, long long n, void* stream);
long long sat_disc_pack_size(int Cout, int Cin, int kh, int kw, int mode);
int sat_disc_pack_weights(const float* w, short* wq, int Cout, int Cin, int kh, int kw, int mode, void* stream);
int sat_disc_conv(const void* xp_hi, const void* xp_lo, const void* wq, const float* bias, float* y, void* em_hi,
                  void* em_lo, int B, int Cin, int Cout, int frames, int W, int kh, int kw, int dil_t, float slope, const float* lk_src,
                  float lk_slope, void* stream);
int sat_disc_wgrad_nsplit(int B, int M, int Cin, int kh, int frames, int W);
int sat_disc_wgrad(const float* dy, const float* x, float* partial, int B, int M, int Cin, int frames, int W, int kh, int kw, int dil_t,
                   void* stream);

/* The stride-1, 5 <= K <= 7 convolutions (the k = 7 convs of the ResidualUnits, autoencoders.py:58-83, and their data-gradients) with
 * the activated input converted ONCE into bf16 hi / lo planes [B][ceil(Cin/8)][rows][8 channels] (row = 32 + t, zero rows around the
 * sequence) instead of per workgroup while staging: sat_conv1d_k7_planes writes the planes (SnakeBeta with pre-exponentiated constants
 * or no activation) unless the producing conv's epilogue already did (sat_conv1d_bf16x3_emit); sat_conv1d_bf16x3_planesq reads them by
 * LDS-DMA.  rows = sat_conv1d_k7_plane_rows(...) (-1: pad > 32 is not supported). */
int sat_conv1d_k7_plane_rows(int Tin, int Tout, int pad);
int sat_conv1d_k7_planes(const float* x, const float* snake_a, const float* snake_ib, short* xp_hi, short* xp_lo, int B, int Cin, int Tin,
                         int rows, void* stream);
/* The planes kernel (csrc/conv1d_bf16x3_k7q.h: 16-channel K-chunks with one tap per MFMA k-step, two wave rows
 * one barrier apart), 5 <= K <= 7: same arguments, the weight planes packed by sat_pack_weights_k7q
 * ([chunk of 16 in-channels][tap][8-channel group][out channel padded to 128][8]; mode 0 = conv weight [out][in][K], mode 1 = the
 * data-gradient of a stride-1 conv).  sat_pack_weights_k7q_size = elements per plane (-1: unsupported). */
long long sat_pack_weights_k7q_size(int D0, int D1, int K, int mode);
int sat_pack_weights_k7q(const float* w, short* hi, short* lo, int D0, int D1, int K, int mode, void* stream);
int sat_conv1d_bf16x3_planesq(const short* xp_hi, const short* xp_lo, int rows, const short* w_hi, const short* w_lo, const float* bias,
                             const float* res, float* y, const float* x2, const float* alpha2, const float* beta2, float* part_da,
                             float* part_db, int B, int Cin, int Cout, int Tin, int Tout, int K, int dil, int pad, int tanh_out,
                             int flags, void* stream);      /* flags bit 0: one workgroup per tile instead of the persistent launch (A/B); bit 1: persistent at any size (tests) */
int sat_convtr1d_bf16x3_partial_rows(int B, int Tout, int stride, int pad);
int sat_pack_weights_bf16x3(const float* w, short* hi, short* lo, int D0, int D1, int K, int stride, int mode, void* stream);
long long sat_pack_weights_bf16x3_size(int D0, int D1, int K, int stride, int mode);
int sat_snake_consts(const float* alpha, const float* beta, float* a, float* ib, int C, void* stream);

/* Transposed conv, K == 2*stride (the Oobleck resampler, autoencoders.py:266-268), polyphase form.
 * Also the data-gradient of the strided down-conv (:245-247).  w_packed: [r][j][Cin][Cout]
 * (sat_pack_weights mode 2).  Same prologue/epilogue options as sat_conv1d. stride in [2, 8]. */
int sat_convtr1d(const float* x, const float* w_packed, const float* bias, const float* alpha, const float* beta,
                 const float* res, float* y, const float* x2, const float* alpha2, const float* beta2,
                 float* part_da, float* part_db, int B, int Cin, int Cout, int Tin, int Tout, int K, int stride,
                 int pad, int tanh_out, void* stream);
int sat_convtr1d_partial_rows(int B, int Tout, int stride, int pad);

/* Weight gradient of any of the above as a split-K GEMM over (batch, time):
 *   dW[m][n][k] = sum_{b,t} actA(lo[b][m][t]) * actB(hi[b][n][t*stride + k*dil - pad])
 * snake_on: 0 none, 1 snake on `lo` rows (ConvTranspose input), 2 on `hi` rows (Conv1d input).
 * Writes sat_conv_wgrad_nsplit(...) slabs of M*N*K floats into `partial`, element (m,n,k) at
 * m*so_m + n*so_n + k*so_k; sum the slabs with sat_reduce_splits. */
int sat_conv_wgrad(const float* lo, const float* hi, const float* alpha, const float* beta, int snake_on,
                   float* partial, long long so_m, long long so_n, long long so_k, int B, int M, int N, int Tlo,
                   int Thi, int K, int stride, int dil, int pad, void* stream);
int sat_conv_wgrad_nsplit(int B, int M, int N, int Tlo, int K, int stride, int dil);

/* The K = 7, stride-1, dilation 1|3|9 case of the above (every ResidualUnit's k7 conv) on the bf16 matrix cores at
 * fp32 accuracy (hi/lo split).  dy: (B, M, T), x: (B, N, T) pre-activation, alpha/beta: SnakeBeta log-params of x or
 * NULL.  Slab stride M*N*7; nsplit from sat_conv_wgrad7_bf16x3_nsplit.  dy_rowsum (or NULL): [M][nsplit] per-split sums over
 * (b, t) of the dy rows — the conv's bias gradient, produced by the workgroups that stream dy anyway (sum the nsplit
 * columns with sat_rowsum) when sat_conv_wgrad7_bf16x3_fuses_rowsum says so: the 4-wave kernel (N < 64 or T % 4 != 0) always, the
 * pipelined kernel only in its A/B variant (SAT_WG_ROWSUM=1: per-thread partial sums while a stage is converted; it saves 13 GB of
 * HBM reads per train step but costs the kernel ~13 % in situ, so the default is a separate sat_rowsum pass). */
int sat_conv_wgrad7_bf16x3(const float* dy, const float* x, const float* alpha, const float* beta, float* partial,
                           long long so_m, long long so_n, long long so_k, int B, int M, int N, int T, int dil, int pad,
                           float* dy_rowsum, void* stream);
int sat_conv_wgrad7_bf16x3_nsplit(int B, int M, int N, int T);
int sat_conv_wgrad7_bf16x3_fuses_rowsum(int B, int M, int N, int T);

/* sat_conv_wgrad for K == 1 (stride 1; the k1 conv of every ResidualUnit) and K == 2*stride with a power-of-two stride
 * (the down / up convs, autoencoders.py:245-247, :266-268) on the bf16 matrix cores at fp32 accuracy.  Same arguments as
 * sat_conv_wgrad (dilation 1); slab stride M*N*K; nsplit from sat_conv_wgrad_bf16x3_nsplit (-1: unsupported shape).
 * lo_rowsum (or NULL): [M][nsplit] per-split row sums of the raw lo tensor (the bias gradient when lo = dy). */
int sat_conv_wgrad_bf16x3(const float* lo, const float* hi, const float* alpha, const float* beta, int snake_on,
                          float* partial, long long so_m, long long so_n, long long so_k, int B, int M, int N, int Tlo,
                          int Thi, int K, int stride, int pad, float* lo_rowsum, void* stream);
int sat_conv_wgrad_bf16x3_nsplit(int B, int M, int N, int Tlo, int K, int stride);

/* The whole backward of a ResidualUnit's 1x1 conv (autoencoders.py:58-83: y = x + conv1(snake2(h))) in ONE pass over dy and h
 * (csrc/ru_k1_bwd.hip; C == 128, T % 32 == 0 — sat_ru_k1_bwd_nsplit returns -1 otherwise and the caller keeps
 * sat_conv_wgrad_bf16x3 + sat_conv1d_bf16x3_emit + sat_rowsum): replaces the autograd of F.conv1d(k = 1) + snake_beta for that conv.
 *   dh (B, C, T)               = (W2^T dy) * dsnake2(h)        [+ em_hi / em_lo: dh as the k7 data-gradient's activation planes
 *                                                                 [B][C/8][em_rows][8], row 32 + t; NULL: not written]
 *   dw_partial [nsplit][C][C]  : slabs of dW2 (torch layout (Cout, Cin, 1)); sum with sat_reduce_splits
 *   part [4][C][nsplit]        : per-split sums of d log-alpha2, d log-beta2, dh (= bias gradient of the k7 conv), dy (= bias
 *                                gradient of the 1x1 conv); sum the last axis (sat_rowsum)
 * wt_hi / wt_lo: W2^T as bf16 hi / lo planes [ci][co] (sat_ru_k1_pack of the (C, C) weight).  All pointers 16-byte aligned. */
int sat_ru_k1_bwd_nsplit(int B, int C, int T);
int sat_ru_k1_pack(const float* w, short* hi, short* lo, int C, void* stream);
int sat_ru_k1_bwd(const float* dy, const float* h, const short* wt_hi, const short* wt_lo, const float* alpha2, const float* beta2,
                  float* dh, short* em_hi, short* em_lo, int em_rows, float* dw_partial, float* part, int B, int C, int T,
                  void* stream);

/* out[i] (+)= scale * sum_z partial[z*count + i]   (deterministic split reduction) */
int sat_reduce_splits(const float* partial, float* out, long long count, int nsplit, float scale, int accumulate,
                      void* stream);
/* bias gradient / row sums: partial[c][z] = sum over the z-th time slice and all b of x[b][c][t]; z < nsplit = sat_rowsum_nsplit(T).
 * nsplit > 1: sum the (1, C, nsplit) partials with a second call. */
int sat_rowsum(const float* x, float* partial, int B, int C, int T, void* stream);
int sat_rowsum_nsplit(int T);

/* torch.nn.utils.weight_norm (dim 0) fold and its gradient — autoencoders.py:23-27.
 * v: (D0, R) g: (D0).  fold: w = g*v/||v||, norm[d] = ||v[d]||.  grad: dv, dg from dw. */
int sat_wn_fold(const float* v, const float* g, float* w, float* norm, int D0, int R, void* stream);
int sat_wn_grad(const float* v, const float* g, const float* norm, const float* dw, float* dv, float* dg, int D0,
                int R, void* stream);
/* The same gradient taken straight from a weight-gradient kernel's split slabs (sat_conv_wgrad*, sat_ru_k1_bwd): element (d, n, k) of
 * dW = sum over z < nsplit of partial[z * count + d * so_m + n * so_n + k * so_k]; v / dv (D0, N, K) in torch layout.  Replaces
 * sat_reduce_splits (+ layout permute) + sat_wn_grad for a weight-normed conv (reference: torch.nn.utils.weight_norm's backward through
 * the convs of models/autoencoders.py:23-27).  bias_partial (D0, bias_cols) | NULL: per-split sums of dy (a weight-gradient kernel's
 * row sums, or sat_rowsum's first pass); dbias[d] = their sum — the conv's bias gradient finishes in the same launch. */
int sat_wn_grad_splits(const float* partial, int nsplit, long long count, long long so_m, long long so_n, long long so_k,
                       const float* v, const float* g, const float* norm, float* dv, float* dg, int D0, int N, int K,
                       const float* bias_partial, int bias_cols, float* dbias, void* stream);
/* The two-channel ends of the Oobleck stack (csrc/edge_conv.hip, round 6): the encoder's first conv (models/autoencoders.py:303), the
 * decoder's last conv (:355-356: Snake -> WNConv1d(channels, out_channels, 7, padding=3, bias=False)), its data-gradient and both
 * weight gradients as fp32 FMA streams bound by the one pass over the 128-channel tensor (a 2-channel operand wastes 7/8 of an MFMA
 * k-chunk / 63/64 of an output tile).  sat_edge_conv_ok: stride 1, dilation 1, odd K <= 7 with 2 * pad == K - 1, one of (cin, cout)
 * <= 2 and the other >= 8. */
int sat_edge_conv_ok(int cin, int cout, int k, int stride, int dil, int pad);
/* rows of the data-gradient epilogue's partial sums: part_da / part_db are [Cout][rows] (sum the rows: sat_rowsum) */
int sat_edge_conv_partial_rows(int B, int T);
/* y (B, Cout, T) = conv1d(act(x), W) + bias.  w: the torch weight (Cout, Cin, K) with mode 0, or (Cin, Cout, K) with mode 1 (the
 * data-gradient of that weight's conv: transposed, taps flipped).  alpha / beta: log parameters of the INPUT's SnakeBeta (Cout <= 2
 * only) | NULL.  x2 / alpha2 / beta2 / part_da / part_db: y *= dsnake(x2) with the per-channel d log-alpha / d log-beta partial sums
 * (Cin <= 2 only) | NULL.  em_hi / em_lo | NULL (Cin <= 2 only): plane emission as sat_conv1d_bf16x3_emit — act_next(y) as the bf16 hi / lo
 * planes [B][ceil(Cout/8)][em_rows][8] (row 32 + t) of the k7 conv that reads y next; em_alpha / em_beta: its SnakeBeta's log parameters | NULL. */
int sat_edge_conv(const float* x, const float* w, const float* bias, const float* alpha, const float* beta, float* y, const float* x2,
                  const float* alpha2, const float* beta2, float* part_da, float* part_db, short* em_hi, short* em_lo,
                  const float* em_alpha, const float* em_beta, int em_rows, int B, int Cin, int Cout, int T, int K, int pad, int mode,
                  int tanh_out, void* stream);
/* slabs sat_edge_conv_wgrad writes for this shape (-1: bad shape) */
int sat_edge_conv_wgrad_nsplit(int B, int M, int N, int T);
/* dW (M, N, K) of y = conv1d(act(x), W) as slabs partial[nsplit][M * N * K] in torch order (sat_wn_grad_splits / sat_reduce_splits);
 * dy (B, M, T), x (B, N, T) pre-activation; alpha / beta: x's SnakeBeta (M <= 2 form only) | NULL; rowsum [M][nsplit] | NULL: per-slab
 * row sums of dy = the bias gradient one reduction short (N <= 2 form only). */
int sat_edge_conv_wgrad(const float* dy, const float* x, const float* alpha, const float* beta, float* partial, float* rowsum, int B,
                        int M, int N, int T, int K, int pad, void* stream);
/* torch weight w[D0][D1][K] -> GEMM-side layout.  mode 0: [D1][k][D0]; 1: [D0][K-1-k][D1]; 2: [r][j][D0][D1], k=r+j*S */
int sat_pack_weights(const float* w, float* out, int D0, int D1, int K, int S, int mode, void* stream);

/* VAEBottleneck — models/bottleneck.py:105-133.  pre: (B, 2C, T) = [mean | scale]; noise: (B, C, T)
 * N(0,1) draw supplied by the caller.  fwd: z = noise*(softplus(scale)+1e-4)+mean and kl partial sums
 * (sat_vae_nblocks(B*C*T) floats; kl = sum / (B*T)).  bwd: dpre from dz (may be NULL) and the device
 * scalar dkl (may be NULL). */
int sat_vae_nblocks(long long n);
int sat_vae_sample_fwd(const float* pre, const float* noise, float* z, float* kl_partial, int B, int C, int T,
                       void* stream);
int sat_vae_sample_bwd(const float* pre, const float* noise, const float* dz, const float* dkl, float* dpre, int B,
                       int C, int T, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Multi-resolution STFT loss — training/losses/auraloss.py: FIRFilter :76-169, STFTLoss :226-449,
 * MultiResolutionSTFTLoss :451-539, SumAndDifferenceSTFTLoss :542-615.  Replaces F.conv1d (FIR),
 * torch.stft, the magnitude/log/norm reductions and their autograd.
 * ---------------------------------------------------------------------------------------------- */
/* y[n][t] = sum_k taps[k] * x[n][t + k - ntaps/2] (zero padded); adjoint != 0 applies the transpose. x: (N, T) */
int sat_fir(const float* x, const float* taps, float* y, int N, int T, int ntaps, int adjoint, void* stream);
/* One resolution. x, y: (NI, C, T), C in {1,2}; views: (NV, 2) channel weights (sum/diff/left/right).
 * fwd writes partial[NI][NV][3][tile] = {sum(|Y|-|X|)^2, sum|Y|^2, sum|log|X|-log|Y||}, tile < sat_stft_tiles()  (sum over tiles: sat_rowsum).
 * bwd writes dL/dy — or dL/dx if wrt_x — given coef[NI][NV][3] = {c1, c2, c3}: dL/d|Y| = c1*((|Y|-|X|) - c2*|Y|) +
 * c3*sign(log|Y|-log|X|)/|Y|, as FOUR planes dy[4][NI][C][T] (even / odd workgroups x direct / reflected samples; the caller
 * zero-fills them and sums them in a fixed order): plain stores, no atomics — the gradient is bit-reproducible.
 * Periodic Hann window of length n_fft, centre/reflect padding, hop, one-sided, unnormalised, power clamped at 1e-8.
 * Round 6: each CHANNEL is transformed once and the views' bins are formed from the channels' (the DFT is linear); the backward needs
 * n_fft <= (frames per workgroup + 1) * hop (every hop >= n_fft / 4; status 1 otherwise — the two-plane write-out's invariant). */
int sat_stft_tiles(int n_fft, int hop, int T);
int sat_stft_fwd(const float* x, const float* y, const float* views, float* partial, int NI, int C, int T, int NV,
                 int n_fft, int hop, void* stream);
int sat_stft_bwd(const float* x, const float* y, const float* views, const float* coef, float* dy, int NI, int C,
                 int T, int NV, int n_fft, int hop, int wrt_x, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer — torch.optim.AdamW as configured by training/utils.py:60-79 and
 * configs/model_configs/autoencoders/stable_audio_2_0_vae.json:41-49, over ONE flat buffer; optional
 * EMA shadow (ema_pytorch.EMA, training/autoencoders.py:262-270) updated from the pre-step parameters.
 * grad_scale folds the 1/world_size of the data-parallel mean into the same pass.
 * ---------------------------------------------------------------------------------------------- */
int sat_adamw_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, float grad_scale, float* ema, float ema_decay,
                   void* stream);
/* The same step with the per-step scalars read from DEVICE memory: hyper[5] (fp32) = {lr, 1 - beta1^t, sqrt(1 - beta2^t), grad_scale,
 * ema_decay}.  A whole optimisation step captured into a HIP graph (training.GraphedTrainStep — the reference's
 * training_step, training/autoencoders.py:367-527, as ONE graph launch) is replayed with the next step's values by rewriting 20 bytes. */
int sat_adamw_step_dev(float* p, const float* g, float* m, float* v, long long n, const float* hyper, float beta1, float beta2,
                       float eps, float weight_decay, float* ema, void* stream);

/* Many small fp32 device-to-device copies in a few launches — the gather of the per-parameter gradients autograd produced into the
 * flat gradient buffer (what torch's AccumulateGrad does with one `add` launch per parameter for the reference's optimizers,
 * training/autoencoders.py:507-515).  entries: HOST array of nent {const float* src; float* dst; long long numel} (24 bytes each; read
 * during the call: the table rides in the kernel arguments, 160 entries per launch, so a HIP-graph capture records it with the launch). */
int sat_multi_copy(const void* entries, int nent, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange — what Lightning's `ddp` strategy does for the reference (train.py:138, :148-164): the SUM of
 * the flat gradient buffer over the ranks, RCCL over xGMI, one communicator per process, enqueued on the caller's stream.
 * The default exchange of training.GradAllReduce goes through torch.distributed (backend "nccl" = RCCL); these entry points
 * are the same collectives behind the C-ABI (`GradAllReduce(native=True)`).  RCCL is resolved at first use with dlopen (the
 * copy already mapped into the process, else the system's) — not a load-time dependency of this library.
 *   sat_allreduce_available   1 when an RCCL library can be loaded
 *   sat_allreduce_unique_id   rank 0: a fresh 128-byte id, handed to every rank by the caller's side channel
 *   sat_allreduce_init        every rank (its GPU current): join `world` ranks; *comm = opaque handle (the library's only state)
 *   sat_allreduce_bucket      in-place SUM of `count` elements (dtype 0 fp32 | 1 bf16); mode 0 all-reduce, mode 1
 *                             reduce-scatter + all-gather in place (count % world == 0, else mode 0)
 *   sat_allreduce_finalize    destroy the communicator
 * ---------------------------------------------------------------------------------------------- */
int sat_allreduce_available(void);
int sat_allreduce_unique_id(void* id128);
int sat_allreduce_init(const void* id128, int world, int rank, void** comm);
int sat_allreduce_bucket(void* comm, void* buf, long long count, int dtype, int mode, void* stream);
int sat_allreduce_finalize(void* comm);

/* ------------------------------------------------------------------------------------------------
 * DiT block operators — models/transformer.py.  dtype: 0 = fp32 tensors, 1 = bf16 tensors
 * (statistics / softmax / accumulation always fp32).
 * ---------------------------------------------------------------------------------------------- */
/* Attention.apply_attn (transformer.py:406-441): o = softmax(q k^T * scale) v, dense, non-causal, no mask;
 * grouped-query when Hkv < H (replaces repeat_interleave :408-411); head_dim must be 64.
 *
 * sat_attn_prepare: src element (b,h,n,d) at b*sb + h*sh + n*sn + d (ELEMENT strides, d contiguous, fp32 or
 * bf16) -> bf16 planes zero-padded to Np = N rounded up to 64: row-major [B][H][Np][64] (rm_*) and/or transposed
 * [B][H][64][Np] (tr_*); NULL outputs are skipped.  fp32 sources (dtype 0) give hi AND lo planes (x ~ hi + lo)
 * and the kernels then form every product as three bf16 MFMAs — the fp32-parity mode; bf16 sources use hi only. */
int sat_attn_prepare(const void* src, long long sb, long long sh, long long sn, short* rm_hi, short* rm_lo,
                     short* tr_hi, short* tr_lo, int B, int H, int N, int Np, int dtype, void* stream);
/* forward: q_* row-major planes (B,H,Nqp,64), k_* row-major (B,Hkv,Nkp,64), vt_* TRANSPOSED (B,Hkv,64,Nkp);
 * o: (B, Nq, H*64) in the model dtype — heads merged; lse: (B, H, Nq) fp32 or NULL.  dtype 1 (bf16): a wave owns 64 queries when
 * Nq >= 2048 and 256-query workgroups still put two on every CU (the long context), else 32; dtype 2 / 3 force 32 / 64 (A/B, tests). */
int sat_attention_fwd(const short* q_hi, const short* q_lo, const short* k_hi, const short* k_lo, const short* vt_hi,
                      const short* vt_lo, void* o, float* lse, int B, int H, int Hkv, int Nq, int Nk, int Nqp, int Nkp,
                      int head_dim, float scale, int dtype, void* stream);
/* dsum[b][h][q] = sum_d dout[b][q][h*64+d] * out[b][q][h*64+d]   (the softmax-gradient row term) */
int sat_attention_rowdot(const void* dout, const void* out, float* dsum, int B, int H, int Nq, int dtype, void* stream);
/* backward (autograd of the above): planes[16] = {q_rm, k_rm, v_rm, k_tr, q_tr, dout_rm, dout_tr, unused} x {hi, lo};
 * two launches: dQ (wave = 32 queries, key tiles) and dK/dV (wave = 32 keys, query tiles, summed over the query
 * heads of a kv group); probabilities are recomputed from lse.  dq: (B,H,Nq,64), dk/dv: (B,Hkv,Nk,64), model dtype. */
int sat_attention_bwd(const short* const* planes, const float* lse, const float* dsum, void* dq, void* dk, void* dv,
                      int B, int H, int Hkv, int Nq, int Nk, int Nqp, int Nkp, int head_dim, float scale, int dtype,
                      void* stream);
/* Attention against a SHORT key sequence — the DiT's cross-attention (transformer.py:351-357, :459-472; kv heads repeated per
 * :408-411; 130 conditioning tokens): bf16 planes, head dim 64, Nk <= 256.  All keys of a (batch item, kv head) are staged into LDS
 * once per workgroup and serve every query head of the GQA group; exact one-pass softmax (csrc/attention_cross.h).
 * sat_attention_cross_ok: 1 when the shape is served (the callers fall back to sat_attention_fwd / _bwd otherwise). */
int sat_attention_cross_ok(int H, int Hkv, int Nk, int head_dim, int dtype);
int sat_attention_cross_fwd(const short* q_rm, const short* k_rm, const short* v_tr, void* o, float* lse, int B, int H, int Hkv,
                            int Nq, int Nk, int Nqp, int Nkp, int head_dim, float scale, void* stream);
/* bytes of caller-owned workspace for sat_attention_cross_bwd (fp32 dK / dV slabs of the query ranges; -1: bad shape) */
long long sat_attention_cross_bwd_ws(int B, int H, int Hkv, int Nq, int Nk);
/* backward: planes[16] as sat_attention_bwd (only the hi planes are read); three launches — dQ, dK / dV partial sums per query
 * range into ws, their sum in index order (no atomics).  dq: (B,H,Nq,64), dk / dv: (B,Hkv,Nk,64) bf16. */
int sat_attention_cross_bwd(const short* const* planes, const float* lse, const float* dsum, void* dq, void* dk, void* dv,
                            void* ws, long long ws_bytes, int B, int H, int Hkv, int Nq, int Nk, int Nqp, int Nkp, int head_dim,
                            float scale, void* stream);

/* LayerNorm.forward (transformer.py:236-241: gamma, beta buffer, eps) fused with the adaLN modulation
 * y = LN(x) * (1 + scale[b]) + shift[b] (TransformerBlock.forward :682, :697).  x, y: (rows, D); gamma/beta fp32;
 * scale/shift: rows of a (B, ...) tensor with element stride mod_stride between batches, or NULL;
 * rows_per_batch = N.  mean/rstd (rows) are saved when non-NULL. */
int sat_layernorm_fwd(const void* x, const float* gamma, const float* beta, const void* scale, const void* shift,
                      long long mod_stride, void* y, float* mean, float* rstd, int rows, int D, int rows_per_batch,
                      float eps, int dtype, void* stream);
/* The same LayerNorm (+ adaLN modulate) with its output quantised per row to fp8 e4m3 for the projection that consumes it
 * (transformer.py:682/:697 feeding :481 / :263 in the fp8 long-context configuration): q (rows, D) bytes, qscale (rows) = row max / 448 —
 * the row_alpha of sat_gemm_fp8 / sat_gemm_qkv_fp8.  Returns 2 (nothing launched) when the shape is outside the vector path
 * (D % (64 lanes x 16 bytes), 16-byte aligned pointers): use sat_layernorm_fwd + sat_quant_fp8_rows then. */
int sat_layernorm_fwd_fp8(const void* x, const float* gamma, const float* beta, const void* scale, const void* shift,
                          long long mod_stride, void* q, float* qscale, int rows, int D, int rows_per_batch, float eps,
                          int dtype, void* stream);
/* dx plus partial column sums part[3][sat_layernorm_bwd_nblocks()][D] = {d_gamma, d_scale, d_shift} (slabs of one
 * batch item are contiguous; reduce with sat_reduce_splits).  Without modulation (scale == NULL) only part[0] is defined. */
int sat_layernorm_bwd_nblocks(int rows, int rows_per_batch);
int sat_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const void* scale,
                      long long mod_stride, const float* mean, const float* rstd, void* dx, float* part, int rows, int D,
                      int rows_per_batch, int dtype, void* stream);
/* The same with an addend: dx = LayerNorm backward + dres, dres (rows, D) in the activation dtype (or NULL) — the gradient that reached x
 * along the residual connection around the normalised branch (x = x + f(LN(x)), transformer.py:703-712). */
int sat_layernorm_bwd_res(const void* dy, const void* x, const float* gamma, const float* beta, const void* scale,
                          long long mod_stride, const float* mean, const float* rstd, const void* dres, void* dx, float* part,
                          int rows, int D, int rows_per_batch, int dtype, void* stream);

/* RotaryEmbedding.forward (transformer.py:125-138): cs[n][j] = {cos, sin}(pos_scale * n * inv_freq[j]), fp32. */
int sat_rope_tables(const float* inv_freq, float* cs, int N, int half, float pos_scale, void* stream);
/* apply_rotary_pos_emb (transformer.py:155-174), in place: rotates dims [0, 2*half) of every head of
 * t: element (b,n,h,d) at b*sb + n*sn + h*sh + d.  Table row = tab_off + n.  transpose != 0 applies the inverse
 * rotation (the backward pass). */
int sat_rope_apply(void* t, const float* cs, long long sb, long long sn, long long sh, int B, int N, int H, int half,
                   int tab_off, int transpose, int dtype, void* stream);

/* GLU.forward (transformer.py:274-275): out = x * silu(gate) for xin = [x | gate] (rows, 2F);
 * backward != 0: out = d_xin (rows, 2F) from dout (rows, F). */
int sat_swiglu(const void* xin, const void* dout, void* out, long long rows, int F, int backward, int dtype,
               void* stream);
/* y = x * sigmoid(1 - gate[b]) + res   (transformer.py:684-686, :699-701); gate rows at stride gstride. */
int sat_gate_residual(const void* x, const void* gate, long long gstride, const void* res, void* y, int B, int N, int D,
                      int dtype, void* stream);
/* its backward: dx = dy * sigmoid(1-gate); part[B][sat_gate_residual_bwd_nchunks(N)][D] partial sums of d_gate
 * (reduce per batch item with sat_reduce_splits); d_res = dy. */
int sat_gate_residual_bwd_nchunks(int N);
int sat_gate_residual_bwd(const void* dy, const void* x, const void* gate, long long gstride, void* dx, float* part,
                          int B, int N, int D, int dtype, void* stream);

/* Classifier-free-guidance combine + CFG rescale + sampler update in one pass — models/dit.py:400-410 (chunk, uncond +
 * (cond - uncond) * scale, channel-std rescale with scale_phi) and inference/sampling.py:254-307 (v-DDIM), :98-135 (Euler).
 * out2 (ncond*B, C, T): conditioned half first (ncond 2) or the plain output (ncond 1); v = guided(+rescaled) output;
 * y0 = c0x*x + c0v*v (x NULL: y0 = v); y1 = c1x*x + c1v*v (optional).  dtype 0 fp32 / 1 bf16. */
int sat_cfg_step(const void* out2, const void* x, void* y0, void* y1, int B, int C, int T, int ncond, float scale,
                 float phi, float c0x, float c0v, float c1x, float c1v, int dtype, void* stream);
/* The general sampler step: every update rule of inference/sampling.py is linear in (x, v, one more tensor `prev`, the
 * unconditioned output u) — v-DDIM with eta > 0 (:296-300: prev = fresh noise) and cfg_pp (:270-284: eps from u), RK4 stages
 * (:160-170: prev = the running k-sum), DPM-Solver++ (:203-217: prev = the previous step's denoised), ping-pong (:240-247: prev =
 * fresh noise).  coef[8] (HOST, fp32) = c0x c0v c0p c0u c1x c1v c1p c1u:
 *   y0 = c0x*x + c0v*v + c0p*prev + c0u*u,  y1 = c1x*x + c1v*v + c1p*prev + c1u*u   (prev NULL: its terms drop; y1 optional;
 *   u = the unconditioned half of out2 when ncond == 2, else v). */
int sat_sampler_step(const void* out2, const void* x, const void* prev, void* y0, void* y1, int B, int C, int T, int ncond,
                     float scale, float phi, const float* coef, int dtype, void* stream);
/* The same with coef[8] read from DEVICE memory: the launch can be frozen in a HIP graph and replayed with new sampler
 * coefficients (inference/sampling.py changes them every step). */
int sat_sampler_step_dev(const void* out2, const void* x, const void* prev, void* y0, void* y1, int B, int C, int T, int ncond,
                         float scale, float phi, const float* coef, int dtype, void* stream);
/* Complex spectrogram of the MS-STFT discriminator — models/encodec.py:73-76, :97-102 (torchaudio Spectrogram: periodic Hann,
 * normalized by ||w||_2, center = False, onesided, power = None; real / imaginary parts concatenated on the channel axis, axes
 * swapped to (frames, freq)).  x (NI, C, T), C in {1, 2} -> z (NI, 2C, frames, n_fft/2+1), frames = sat_spec_frames().
 * sat_spec_bwd: dz -> dx as TWO planes dx[2][NI][C][T] (even / odd workgroups; caller zero-fills and sums): no atomics. */
int sat_spec_frames(int n_fft, int hop, int T);
int sat_spec_fwd(const float* x, float* z, int NI, int C, int T, int n_fft, int hop, void* stream);
int sat_spec_bwd(const float* dz, float* dx, int NI, int C, int T, int n_fft, int hop, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dense projections of the DiT — models/transformer.py: to_qkv :362/:481, to_out :364/:534, to_q / to_kv :356-357,
 * GLU proj + x*silu(gate) :263-275, FeedForward linear_out :308, residual / gate updates :684-712; models/dit.py :49-77.
 * Replaces nn.Linear (forward, data gradient, weight gradient) and the elementwise kernels that followed it.
 * One "NT" kernel on the bf16 matrix cores (csrc/gemm.hip): C[M,N] = epilogue(A[M,K] · B[N,K]^T), fp32 accumulation.
 * ---------------------------------------------------------------------------------------------- */

/* A (M, K), B (N, K): bf16, K contiguous, row strides lda / ldb in elements (multiples of 8; K and N multiples of 8).
 * epilogue 0: C = acc [+ bias];  1: + res;  2: * sigmoid(1 - gate[m / rows_per_gate]) + res  (adaLN gate, :684/:699);
 * 3: SwiGLU — B = [value rows | gate rows] (N = 2F): C (M, F) = (v + bv) * silu(g + bg); `pre` (M, 2F, row stride ldp),
 * if not NULL, receives the pre-activation for the backward.  bias: fp32 (N) or NULL.  out_f32: C / res / gate / pre are
 * fp32 instead of bf16.  splits > 1 (epilogue 0, fp32, no bias): split-K partial slabs, slab z at C + z*M*ldc — sum them
 * with sat_reduce_splits.  zeros: >= 16 bytes of device zeros (source of K-tail chunks).  tile: 0 = 128x128 workgroup
 * tile (4 waves, two workgroups per CU), 4 = 256x256 (8 waves, two wave rows one barrier apart), 7 / 8 = 160x256 /
 * 128x128 on the eight-wave ring kernel (any other value is an error).  fp32 models run this kernel on sat_split_bf16x3
 * operands (K' = 3K). */
int sat_gemm_bf16(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, const float* bias,
                  const void* res, long long ldr, const void* gate, long long ldg, int rows_per_gate, void* pre,
                  long long ldp, const void* zeros, int M, int N, int K, int epilogue, int out_f32, int splits, int tile,
                  void* stream);

/* Attention input projections with head split, partial rotary (first 32 dims of each 64-dim head, rotate_half pairs
 * (d, d+16), :155-174, table from sat_rope_tables; NULL = none) and the attention kernel's operand layout fused into the
 * epilogue (:469-507).  B = nsec blocks of heads*64 rows, block i = section sec0+i of (q, k, v): to_qkv (0,3), to_q (0,1),
 * to_kv (1,2).  q_rm / k_rm: (nb, heads, npad, 64) bf16; v_tr: (nb, heads, 64, npad) bf16.  Entries past ntok are not
 * written (zero-fill the planes once). */
int sat_gemm_qkv_bf16(const void* A, long long lda, const void* B, long long ldb, const float* rope_cs, int rope_off,
                      void* q_rm, void* k_rm, void* v_tr, const void* zeros, int nb, int ntok, int npad, int heads,
                      int K, int sec0, int nsec, int tile, void* stream);

/* Second half of a split-K projection (sat_gemm_bf16 with splits > 1 writes fp32 slabs): out = sum_z slabs[z] (+ bias) (+ res),
 * in bf16 or fp32 — used for the few-tile / long-K projections (FF2), where cutting K doubles the workgroups on the chip. */
int sat_splitk_epilogue(const float* slabs, int S, const float* bias, const void* res, long long ldr, void* out, long long ldo,
                        int M, int N, int out_f32, void* stream);

/* fp8 (OCP e4m3) forward projections for the long-context configuration (BASELINE.json configs[4], stable_audio_2_0.json:3):
 * as sat_gemm_bf16 / sat_gemm_qkv_bf16 with A (M, K), B (N, K) in fp8 bytes (K, lda, ldb multiples of 16) on
 * v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales; alpha = device scalar (dequant scale of A x that of B).
 * row_alpha (M floats, or NULL): per-row factor applied with alpha — A quantised row by row (sat_quant_fp8_rows; alpha = B's scale).
 * col_alpha (N floats, 16-byte aligned, or NULL): per-column factor — B's rows (the weight's output channels) quantised one by one
 * (sat_quant_fp8_rows on the weight); alpha may be NULL when col_alpha is given.
 * tile: 0 = 128x128 (4 waves), 4 = 256x256, 7 = 160x256, 8 = 128x128 (eight-wave kernels, as sat_gemm_bf16). */
int sat_gemm_fp8(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, const float* bias,
                 const void* res, long long ldr, const void* gate, long long ldg, int rows_per_gate, void* pre,
                 long long ldp, const void* zeros, const float* alpha, const float* row_alpha, const float* col_alpha, int M, int N,
                 int K, int epilogue, int out_f32, int tile, void* stream);
int sat_gemm_qkv_fp8(const void* A, long long lda, const void* B, long long ldb, const float* rope_cs, int rope_off,
                     void* q_rm, void* k_rm, void* v_tr, const void* zeros, const float* alpha, const float* row_alpha,
                     const float* col_alpha, int nb, int ntok, int npad, int heads, int K, int sec0, int nsec, int tile, void* stream);
/* dst (R, C) fp8 e4m3 = saturate_448(src * qscale[0]), round to nearest even; src fp32 | bf16; qscale a DEVICE scalar. */
/* Dynamic per-tensor scale of the fp8 quantisation in one launch (last-arriving block reduces the per-block maxima): scales[0] =
 * 448 / max|src| (what sat_quant_fp8 takes as qscale), scales[1] = max|src| / 448 (the GEMM's de-quantisation factor).  work: >= 1 +
 * sat_absmax_scale_blocks(R, C) floats, work[0] == 0 on entry (left 0).  src (R, C) fp32 | bf16, C % 4 == 0. */
int sat_absmax_scale_blocks(int R, int C);
int sat_absmax_scale(const void* src, long long lds, float* work, float* scales, int R, int C, int src_f32, void* stream);
int sat_quant_fp8(const void* src, long long lds, void* dst, long long ldd, const float* qscale, int R, int C, int src_f32,
                  void* stream);
/* Per-ROW dynamic quantisation in one pass over the activation: dst[r][:] = saturate_448(src[r][:] * 448 / max|src[r][:]|) (round to
 * nearest even), scale[r] = max|src[r][:]| / 448 — the GEMM's row_alpha.  src (R, C) fp32 | bf16, C % 8 == 0, C <= 8192. */
int sat_quant_fp8_rows(const void* src, long long lds, void* dst, long long ldd, float* scale, int R, int C, int src_f32,
                       void* stream);

/* src (R, C) fp32 (src_f32 = 1) or bf16, row stride lds -> dst bf16: (R, C) row stride ldd; or, transpose = 1, (C, Rpad)
 * with columns R..Rpad-1 zero (reduction-dim padding of the weight-gradient GEMM). */
int sat_cast_bf16(const void* src, long long lds, void* dst, long long ldd, int R, int C, int Rpad, int src_f32,
                  int transpose, void* stream);
/* Both bf16 copies of a weight in one pass (nn.Linear under bf16 autocast, models/transformer.py:263,308,362,481: the forward GEMM reads
 * W (R, C), the data-gradient GEMM its transpose): dst (R, C) row stride ldd and dst_t (C, Rpad) row stride ldd_t, columns R..Rpad-1
 * zero.  All three tensors with 16-byte aligned rows, C % 8 == 0 (status 1 otherwise: the caller keeps two sat_cast_bf16 calls). */
/* Two transposing casts in one launch — the operands of a weight-gradient GEMM (autograd of nn.Linear, models/transformer.py:263,308,362,481):
 * dst_a (Ca, Rpad_a) = src_a (Ra, Ca)^T and dst_b (Cb, Rpad_b) = src_b (Rb, Cb)^T, each exactly as sat_cast_bf16(transpose = 1). */
int sat_cast_bf16_tpair(const void* src_a, long long lds_a, void* dst_a, long long ldd_a, int Ra, int Ca, int Rpad_a, int a_f32,
                        const void* src_b, long long lds_b, void* dst_b, long long ldd_b, int Rb, int Cb, int Rpad_b, int b_f32,
                        void* stream);
int sat_cast_bf16_dual(const void* src, long long lds, void* dst, long long ldd, void* dst_t, long long ldd_t, int R, int C, int Rpad,
                       int src_f32, void* stream);

/* fp32 (R, C) -> bf16 (R, 3C): side 0 (activations) [hi | hi | lo], side 1 (weights) [hi | lo | hi], so that
 * A' · B'^T = hi·hi + hi·lo + lo·hi  (|x - hi - lo| <= 2^-17 |x|). */
int sat_split_bf16x3(const float* src, long long lds, void* dst, long long ldd, int R, int C, int side, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SAT_AMD_H */
