// conv_wgrad_bf16x3.hip — weight gradient of the k = 7 stride-1 (dilated) convs — 7/8 of the conv stack's
// weight-gradient FLOPs — on the bf16 matrix cores at fp32 accuracy (hi/lo split, 3 MFMAs per product,
// fp32 accumulation; see conv1d_bf16x3.hip for the numerics).
//
//   dW[co][ci][tap] = sum_b sum_t  dy[b][co][t] * snake(x)[b][ci][t + tap*dil - pad]
//
// GEMM view: the MFMA reduction dim is TIME (16 consecutive steps per v_mfma_f32_32x32x16_bf16), rows = co,
// cols = ci; a wave keeps 7 accumulator tiles (one per tap) of 32(co) x 32(ci), a workgroup covers
// 128(co) x 32(ci); the (b, t) range is split over gridDim.z and the slabs summed by sat_reduce_splits.
// The A fragment (8 consecutive t of one dy row) is an aligned 16-byte LDS read shared by all taps.  The B
// fragment of tap k needs the x row shifted by k*dil samples — never 16-byte aligned — so a wave reads the
// ALIGNED chunks covering [t, t + 8 + 6*dil) once per k-step and forms each tap's fragment with compile-time
// register selection + v_alignbit (dil is a template parameter: 1, 3, 9).
#include "conv_common.h"
#include <type_traits>

#define SAT_WB_TT 64                 // time steps per LDS stage (4 MFMA k-steps)
#define SAT_WB_LOROW (SAT_WB_TT + 8) // 144 B rows: conflict-free b128
#define SAT_WB_HIROW 136             // 64 + 54 + 8 (chunk overrun) = 126 -> 136 elements = 272 B rows: conflict-free b128

struct SatWgBfParams {
    const float* dy;     // (B, M, T)
    const float* x;      // (B, N, T)   conv input (pre-activation)
    const float* alpha;  // (N) snake log-params or null
    const float* beta;
    float* out;          // partial slabs [nsplit][M*N*7] addressed by so_*
    long long so_split, so_m, so_n, so_k;
    int B, M, N, T, pad;
    int chunks_per_split, nchunks, nT;
    float* rowsum;       // or null: [M][nsplit] per-split sums over (b, t) of dy rows (the conv's bias gradient), written by
    int nsplit;          //          the workgroups of the first column tile — dy streams through them anyway
};

// sum of a float4 over the 16 consecutive lanes that hold one staged row (64 time steps); valid in lanes with (lane & 15) == 0
SAT_DEVICE unsigned sat_umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
SAT_DEVICE float sat_row16_sum(float4 q) {
    float s = (q.x + q.y) + (q.z + q.w);
#pragma unroll
    for (int m = 1; m <= 8; m <<= 1) s += __shfl_xor(s, m);
    return s;
}

#if defined(SAT_HIPEMU)
static inline unsigned sat_alignbit(unsigned hi, unsigned lo, unsigned s) {
    return (unsigned)((((unsigned long long)hi << 32) | lo) >> s);
}
#else
SAT_DEVICE unsigned sat_alignbit(unsigned hi, unsigned lo, unsigned s) { return __builtin_amdgcn_alignbit(hi, lo, s); }
#endif


template <int DIL>
__global__ void __launch_bounds__(256)
#if !defined(SAT_HIPEMU)
__attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
sat_wgrad7_bf16x3_kernel(SatWgBfParams p) {
    constexpr int NCH = (6 * DIL + 7) / 8 + 1;                       // aligned 8-element chunks covering all 7 taps
    __shared__ __attribute__((aligned(16))) short lo_lds[2][SAT_CO_T][SAT_WB_LOROW];   // dy  [plane][co][t]
    __shared__ __attribute__((aligned(16))) short hi_lds[2][32][SAT_WB_HIROW];         // act [plane][ci][t]
    __shared__ float sn_a[32], sn_ib[32];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    // grid = (co tiles, ci tiles, splits): the tiles of one split read the same dy / x time range -> same XCD
    int mn_tile, split;
    sat_xcd_tile(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y, gridDim.z, &mn_tile, &split);
    const int m0 = (mn_tile % (int)gridDim.x) * SAT_CO_T, n0 = (mn_tile / (int)gridDim.x) * 32;
    const int m_w = wave * 32;
    const bool wave_on = (m0 + m_w) < p.M;

    f32x16 acc[7];
#pragma unroll
    for (int k = 0; k < 7; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;

    const bool want_rs = p.rowsum != nullptr && n0 == 0;     // block-uniform
    float rs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // row (tid >> 4) + 16 i, valid in lanes with (tid & 15) == 0

    if (tid < 32) {
        const int c = n0 + tid;
        float sa = 1.f, sib = 0.f;
        if (p.alpha && c < p.N) {
            sa = expf(p.alpha[c]);
            sib = 1.0f / (expf(p.beta[c]) + 1e-9f);
        }
        sn_a[tid] = sa;
        sn_ib[tid] = sib;
    }
    // hi-tile columns past the staged span are read by the chunk overrun of the last k-step: keep them zero
    for (int i = tid; i < 2 * 32 * SAT_WB_HIROW; i += 256) (&hi_lds[0][0][0])[i] = 0;
    __syncthreads();

    const int c_begin = split * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > p.nchunks) c_end = p.nchunks;
    constexpr int HSPAN = SAT_WB_TT + 6 * DIL;                        // activation samples needed per stage

    for (int ch = c_begin; ch < c_end; ++ch) {
        const int b = ch / p.nT;
        const int tt0 = (ch - b * p.nT) * SAT_WB_TT;
        // ---- stage dy: 128 rows x 64 t (8 float4 per thread, all loads first) ----
        {
            const float* src = p.dy + (size_t)b * p.M * p.T;
            for (int half = 0; half < 2; ++half) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = tid + (half * 4 + u) * 256;
                const int row = idx >> 4, c4 = (idx & 15) * 4;
                const int m = m0 + row, t = tt0 + c4;
                float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < p.M) {
                    const float* s = src + (size_t)m * p.T + t;
                    if (t + 3 < p.T && ((p.T & 3) == 0)) {
                        q = *reinterpret_cast<const float4*>(s);
                    } else {
                        if (t + 0 < p.T) q.x = s[0];
                        if (t + 1 < p.T) q.y = s[1];
                        if (t + 2 < p.T) q.z = s[2];
                        if (t + 3 < p.T) q.w = s[3];
                    }
                }
                v[u] = q;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = tid + (half * 4 + u) * 256;
                const int row = idx >> 4, c4 = (idx & 15) * 4;
                if (want_rs) rs[half * 4 + u] += sat_row16_sum(v[u]);
                uint32_t h0, h1, l0, l1;
                sat_split2_pk(v[u].x, v[u].y, &h0, &l0);
                sat_split2_pk(v[u].z, v[u].w, &h1, &l1);
                typedef uint32_t u2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<u2*>(&lo_lds[0][row][c4]) = u2{h0, h1};
                *reinterpret_cast<u2*>(&lo_lds[1][row][c4]) = u2{l0, l1};
            }
            }
        }
        // ---- stage snake(x): 32 rows x HSPAN samples starting at tt0 - pad (batches of 8 scalar loads) ----
        {
            const float* src = p.x + (size_t)b * p.N * p.T;
            const int th0 = tt0 - p.pad;
            constexpr int HP = HSPAN / 2;                 // HSPAN is even: pairs never straddle rows
            constexpr int TOTAL = 32 * HP;
            for (int base = tid; base < TOTAL; base += 8 * 256) {
                float v[8][2];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 256;
                    const int row = idx / HP, col = (idx - row * HP) * 2;
                    const int n = n0 + row, t = th0 + col;
                    const bool ok = idx < TOTAL && n < p.N;
                    const float* s = src + (size_t)(ok ? n : 0) * p.T;
                    v[u][0] = (ok && t >= 0 && t < p.T) ? s[t] : 0.0f;
                    v[u][1] = (ok && t + 1 >= 0 && t + 1 < p.T) ? s[t + 1] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 256;
                    if (idx < TOTAL) {
                        const int row = idx / HP, col = (idx - row * HP) * 2;
                        float o0 = v[u][0], o1 = v[u][1];
                        if (p.alpha) {
                            const float sa = sn_a[row], sib = sn_ib[row];
                            o0 = sat_snake(o0, sa, sib);
                            o1 = sat_snake(o1, sa, sib);
                        }
                        uint32_t h, l;
                        sat_split2_pk(o0, o1, &h, &l);
                        *reinterpret_cast<uint32_t*>(&hi_lds[0][row][col]) = h;
                        *reinterpret_cast<uint32_t*>(&hi_lds[1][row][col]) = l;
                    }
                }
            }
        }
        __syncthreads();
        if (wave_on) {
#pragma unroll
            for (int ks = 0; ks < SAT_WB_TT / 16; ++ks) {
                const int tb = 16 * ks + 8 * hi;
                bf16x8 af[2];
                af[0] = *reinterpret_cast<const bf16x8*>(&lo_lds[0][m_w + l31][tb]);
                af[1] = *reinterpret_cast<const bf16x8*>(&lo_lds[1][m_w + l31][tb]);
                // one activation plane at a time (halves the live chunk registers): hi plane pairs with dy hi + lo,
                // lo plane with dy hi only
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    u32x4 cw[NCH];
#if !defined(SAT_HIPEMU)
                    asm volatile("" ::: "memory");       // keep the two planes' chunk loads from being hoisted together
#endif
#pragma unroll
                    for (int j = 0; j < NCH; ++j) cw[j] = *reinterpret_cast<const u32x4*>(&hi_lds[pl][l31][tb + 8 * j]);
#pragma unroll
                    for (int k = 0; k < 7; ++k) {
                        const int off = k * DIL;          // compile-time after unrolling
                        const int wbase = (off >> 3) * 4 + ((off & 7) >> 1);   // first 32-bit word of the fragment
                        const bool odd = (off & 1) != 0;
                        u32x4 r;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int w0 = wbase + i, w1 = wbase + i + 1;
                            const unsigned a0 = cw[w0 >> 2][w0 & 3];
                            if (odd) {
                                const unsigned a1 = cw[w1 >> 2][w1 & 3];
                                r[i] = sat_alignbit(a1, a0, 16);
                            } else {
                                r[i] = a0;
                            }
                        }
                        const bf16x8 bf = __builtin_bit_cast(bf16x8, r);
                        acc[k] = sat_mfma_32x32x16_bf16(af[0], bf, acc[k]);
                        if (pl == 0) acc[k] = sat_mfma_32x32x16_bf16(af[1], bf, acc[k]);
                    }
                }
            }
        }
        __syncthreads();
    }

    if (want_rs && (tid & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + (tid >> 4) + 16 * i;
            if (m < p.M) p.rowsum[(size_t)m * p.nsplit + split] = rs[i];
        }
    }
    if (wave_on) {
        float* ob = p.out + (size_t)split * p.so_split;
        const int n = n0 + l31;
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + m_w + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < p.M && n < p.N) ob[(size_t)m * p.so_m + (size_t)n * p.so_n + (size_t)k * p.so_k] = acc[k][r];
            }
    }
}

#include "conv_wgrad7_bf16x3_pipe.h"     // the 8-wave pipelined kernel (128 x 64 tiles) for N >= 64

struct SatWgBfPlan { int nsplit, cps, nchunks, nT; bool pipe; };
static void sat_wgbf_plan(int B, int M, int N, int T, SatWgBfPlan* pl) {
    pl->nT = sat_cdiv(T, SAT_WB_TT);
    pl->nchunks = B * pl->nT;
    pl->pipe = N >= SAT_WP_NI && (T & 3) == 0;     // at least one full 64-channel column block; 16-byte dy loads
    const int tiles = sat_cdiv(M, SAT_CO_T) * sat_cdiv(N, pl->pipe ? SAT_WP_NI : 32);
    int want = 512 / tiles;                        // pipelined kernel: 1 workgroup per CU, two FULL rounds at most; 4-wave kernel: 2 per CU, one round
    if (want > pl->nchunks) want = pl->nchunks;
    if (want < 1) want = 1;
    if (want > 512) want = 512;
    pl->cps = sat_cdiv(pl->nchunks, want);
    pl->nsplit = sat_cdiv(pl->nchunks, pl->cps);
}
extern "C" int sat_conv_wgrad7_bf16x3_nsplit(int B, int M, int N, int T) {
    SatWgBfPlan pl;
    sat_wgbf_plan(B, M, N, T, &pl);
    return pl.nsplit;
}
extern "C" int sat_conv_wgrad7_bf16x3_fuses_rowsum(int B, int M, int N, int T) {
    SatWgBfPlan pl;
    sat_wgbf_plan(B, M, N, T, &pl);
    // the 4-wave kernel produces dy_rowsum; the pipelined one does not (four more live registers cost it ~13 % in situ: profiles/EXPERIMENTS.md)
    return pl.pipe ? 0 : 1;
}
// dW[m][n][k] for a K = 7, stride-1 conv with dilation in {1, 3, 9}: dy (B, M, T), x (B, N, T) pre-activation,
// alpha/beta = SnakeBeta log-params of the conv input (or NULL).  Writes nsplit slabs (element (m,n,k) at
// m*so_m + n*so_n + k*so_k, slab stride M*N*7); sum them with sat_reduce_splits.
extern "C" int sat_conv_wgrad7_bf16x3(const float* dy, const float* x, const float* alpha, const float* beta, float* partial,
                                      long long so_m, long long so_n, long long so_k, int B, int M, int N, int T, int dil,
                                      int pad, float* dy_rowsum, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0 || T <= 0) { sat_set_error("sat_conv_wgrad7_bf16x3: empty shape"); return 1; }
    if (dil != 1 && dil != 3 && dil != 9) { sat_set_error("sat_conv_wgrad7_bf16x3: dilation must be 1, 3 or 9 (the Oobleck ResidualUnits)"); return 1; }
    if ((alpha == nullptr) != (beta == nullptr)) { sat_set_error("sat_conv_wgrad7_bf16x3: alpha/beta must both be given"); return 1; }
    SatWgBfPlan pl;
    sat_wgbf_plan(B, M, N, T, &pl);
    SatWgBfParams p{dy, x, alpha, beta, partial, (long long)M * N * 7, so_m, so_n, so_k, B, M, N, T, pad, pl.cps, pl.nchunks, pl.nT,
                    dy_rowsum, pl.nsplit};
    if (pl.pipe) {
        dim3 grid(sat_cdiv(M, SAT_CO_T), sat_cdiv(N, SAT_WP_NI), pl.nsplit);
        if (dy_rowsum) { sat_set_error("sat_conv_wgrad7_bf16x3: this shape's kernel does not fuse the row sums (sat_conv_wgrad7_bf16x3_fuses_rowsum)"); return 1; }
        if (dil == 1) SAT_LAUNCH((sat_wgrad7_bf16x3_pipe_kernel<1>), grid, dim3(SAT_WP_NT), stream, p);
        else if (dil == 3) SAT_LAUNCH((sat_wgrad7_bf16x3_pipe_kernel<3>), grid, dim3(SAT_WP_NT), stream, p);
        else SAT_LAUNCH((sat_wgrad7_bf16x3_pipe_kernel<9>), grid, dim3(SAT_WP_NT), stream, p);
        return sat_check_launch("sat_conv_wgrad7_bf16x3");
    }
    dim3 grid(sat_cdiv(M, SAT_CO_T), sat_cdiv(N, 32), pl.nsplit);
    if (dil == 1) SAT_LAUNCH(sat_wgrad7_bf16x3_kernel<1>, grid, dim3(256), stream, p);
    else if (dil == 3) SAT_LAUNCH(sat_wgrad7_bf16x3_kernel<3>, grid, dim3(256), stream, p);
    else SAT_LAUNCH(sat_wgrad7_bf16x3_kernel<9>, grid, dim3(256), stream, p);
    return sat_check_launch("sat_conv_wgrad7_bf16x3");
}

// =====================================================================================================================
// The short-kernel weight gradients: k = 1 convs (one tap) and the K = 2*stride down / up convs (two virtual taps
// over space-to-depth rows of the longer tensor) — same contract as sat_conv_wgrad (conv_wgrad.hip):
//
//   dW[m][n][k = j*S + r] = sum_b sum_t  actA(lo[b][m][t]) * actB(hi[b][n][(t + j)*S + r - pad])
//
// With one or two taps a 128 x 32 tile would restage operands far too often per MFMA, so the workgroup tile is
// 128 (lo rows) x 128 (virtual hi rows v = n*S + r), a wave owns 64 x 64 (2 x 2 MFMA tiles per tap).  The hi tile is
// staged by walking each real channel's contiguous samples and scattering them to row n*S + u%S, column u/S.
// =====================================================================================================================
#define SAT_WS_TT 64
#define SAT_WS_LOROW (SAT_WS_TT + 8)   // 144 B rows
#define SAT_WS_HIROW (SAT_WS_TT + 8)   // 144 B rows like lo (conflict-free 16-byte fragment reads; 160-byte rows were 2-way): columns 0..65 are written, the tap-1 chunk reads up to 71

struct SatWgSmallParams {
    const float* lo;     // (B, M, Tlo)
    const float* hi;     // (B, N, Thi)
    const float* alpha;  // snake log-params of lo's channels (snake_on == 1) or hi's (snake_on == 2), or null
    const float* beta;
    float* out;
    long long so_split, so_m, so_n, so_k;
    int B, M, N, Tlo, Thi, pad, snake_on, s_log2;
    int chunks_per_split, nchunks, nT;
    float* rowsum;       // or null: [M][nsplit] per-split sums over (b, t) of the raw lo rows (bias gradient when lo = dy)
    int nsplit;
};

template <int NT>
__global__ void __launch_bounds__(256)
#if !defined(SAT_HIPEMU)
__attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
sat_wgrad_small_bf16x3_kernel(SatWgSmallParams p) {
    constexpr int NCH = NT;                                          // aligned chunks per fragment row read
    __shared__ __attribute__((aligned(16))) short lo_lds[2][SAT_CO_T][SAT_WS_LOROW];
    __shared__ __attribute__((aligned(16))) short hi_lds[2][SAT_CO_T][SAT_WS_HIROW];
    __shared__ float sn_a[SAT_CO_T], sn_ib[SAT_CO_T];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int sl = p.s_log2, S = 1 << sl;
    int mn_tile, split;
    sat_xcd_tile(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y, gridDim.z, &mn_tile, &split);
    const int m0 = (mn_tile % (int)gridDim.x) * SAT_CO_T, v0 = (mn_tile / (int)gridDim.x) * SAT_CO_T;
    const int n_base = v0 >> sl, nc = SAT_CO_T >> sl;                // real hi channels of this tile
    const int NV = p.N << sl;
    const int m_w = (wave >> 1) * 64, v_w = (wave & 1) * 64;
    const bool wave_on = (m0 + m_w) < p.M && (v0 + v_w) < NV;
    const bool snake_lo = p.alpha && p.snake_on == 1, snake_hi = p.alpha && p.snake_on == 2;

    f32x16 acc[2][2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < NT; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][k][r] = 0.0f;

    const bool want_rs = p.rowsum != nullptr && v0 == 0;     // block-uniform
    float rsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // row (tid >> 4) + 16 i, valid in lanes with (tid & 15) == 0

    if (tid < SAT_CO_T) {
        float sa = 1.f, sib = 0.f;
        const int c = snake_lo ? m0 + tid : n_base + tid;
        const bool ok = snake_lo ? c < p.M : (snake_hi && tid < nc && c < p.N);
        if (ok) {
            sa = expf(p.alpha[c]);
            sib = 1.0f / (expf(p.beta[c]) + 1e-9f);
        }
        sn_a[tid] = sa;
        sn_ib[tid] = sib;
    }
    __syncthreads();

    const int c_begin = split * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > p.nchunks) c_end = p.nchunks;
    const int rs = (SAT_WS_TT + NT - 1) << sl;                       // real samples per hi channel and stage
    const int up = rs >> 1;                                          // ... in pairs (rs is even)
    const int total_units = nc * up;

    auto mfma_stage = [&]() {
        if (wave_on) {
            SAT_MFMA_PRIO(1);
#pragma unroll
            for (int ks = 0; ks < SAT_WS_TT / 16; ++ks) {
                const int tb = 16 * ks + 8 * hi;
                bf16x8 af[2][2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        af[mi][pl] = *reinterpret_cast<const bf16x8*>(&lo_lds[pl][m_w + mi * 32 + l31][tb]);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        u32x4 cw[NCH];
#pragma unroll
                        for (int j = 0; j < NCH; ++j) cw[j] = *reinterpret_cast<const u32x4*>(&hi_lds[pl][v_w + ni * 32 + l31][tb + 8 * j]);
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            u32x4 r = cw[0];
                            if (j == 1) {
                                r[0] = sat_alignbit(cw[0][1], cw[0][0], 16);
                                r[1] = sat_alignbit(cw[0][2], cw[0][1], 16);
                                r[2] = sat_alignbit(cw[0][3], cw[0][2], 16);
                                r[3] = sat_alignbit(cw[NCH - 1][0], cw[0][3], 16);
                            }
                            const bf16x8 bf = __builtin_bit_cast(bf16x8, r);
#pragma unroll
                            for (int mi = 0; mi < 2; ++mi) {
                                acc[mi][ni][j] = sat_mfma_32x32x16_bf16(af[mi][0], bf, acc[mi][ni][j]);
                                if (pl == 0) acc[mi][ni][j] = sat_mfma_32x32x16_bf16(af[mi][1], bf, acc[mi][ni][j]);
                            }
                        }
                    }
                }
            }
            SAT_MFMA_PRIO(0);
        }
    };

    // k = 1 convs (NT == 1, no padding, 16-byte aligned rows): both operands are 128 x 64 tiles read as float4, and the
    // stage is software-pipelined — the loads of stage c+1 are issued before the MFMAs of stage c
    const bool fast1 = (NT == 1) && sl == 0 && p.pad == 0 && (p.Tlo & 3) == 0 && p.Thi == p.Tlo;
    if (fast1) {
        float4 vlo[8], vhi[8];
        const int row0 = tid >> 4, c4 = (tid & 15) * 4;
        auto issue = [&](int ch) {
            const int b = ch / p.nT;
            const int tt0 = (ch - b * p.nT) * SAT_WS_TT;
            const float* slo = p.lo + (size_t)b * p.M * p.Tlo;
            const float* shi = p.hi + (size_t)b * p.N * p.Thi;
            const bool t_ok = tt0 + c4 < p.Tlo;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int m = m0 + row0 + 16 * u, n = v0 + row0 + 16 * u;
                const bool okm = t_ok && m < p.M, okn = t_ok && n < p.N;
                const float4 a = *reinterpret_cast<const float4*>(slo + (size_t)(okm ? m : 0) * p.Tlo + (okm ? tt0 + c4 : 0));
                const float4 c = *reinterpret_cast<const float4*>(shi + (size_t)(okn ? n : 0) * p.Thi + (okn ? tt0 + c4 : 0));
                vlo[u] = okm ? a : make_float4(0.f, 0.f, 0.f, 0.f);
                vhi[u] = okn ? c : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto convert = [&]() {
            typedef uint32_t u2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = row0 + 16 * u;
                float4 q = vlo[u];
                if (want_rs) rsum[u] += sat_row16_sum(q);
                if (snake_lo) {
                    const float sa = sn_a[row], sib = sn_ib[row];
                    q.x = sat_snake(q.x, sa, sib); q.y = sat_snake(q.y, sa, sib);
                    q.z = sat_snake(q.z, sa, sib); q.w = sat_snake(q.w, sa, sib);
                }
                uint32_t h0, h1, l0, l1;
                sat_split2_pk(q.x, q.y, &h0, &l0);
                sat_split2_pk(q.z, q.w, &h1, &l1);
                *reinterpret_cast<u2*>(&lo_lds[0][row][c4]) = u2{h0, h1};
                *reinterpret_cast<u2*>(&lo_lds[1][row][c4]) = u2{l0, l1};
                q = vhi[u];
                if (snake_hi) {
                    const float sa = sn_a[row], sib = sn_ib[row];
                    q.x = sat_snake(q.x, sa, sib); q.y = sat_snake(q.y, sa, sib);
                    q.z = sat_snake(q.z, sa, sib); q.w = sat_snake(q.w, sa, sib);
                }
                sat_split2_pk(q.x, q.y, &h0, &l0);
                sat_split2_pk(q.z, q.w, &h1, &l1);
                *reinterpret_cast<u2*>(&hi_lds[0][row][c4]) = u2{h0, h1};
                *reinterpret_cast<u2*>(&hi_lds[1][row][c4]) = u2{l0, l1};
            }
        };
        issue(c_begin);
        for (int ch = c_begin; ch < c_end; ++ch) {
            convert();
            __syncthreads();
            if (ch + 1 < c_end) issue(ch + 1);
            mfma_stage();
            __syncthreads();
        }
    } else if ((NT == 2) && sl >= 1 && sl <= 3 && (p.Tlo & 3) == 0) {
        // K = 2 S convs with S = 2, 4, 8 (every down / up conv of the Oobleck stack), round 4: the same software pipeline as the k = 1
        // path — the global loads of stage c+1 are in flight under the MFMAs of stage c — with the hi tile (128 virtual rows x 66
        // columns = nc real channels x 66 S contiguous samples) read as 16-byte QUADS (2112 per stage, 8.25 per thread; dword-aligned
        // global_load_dwordx4: the padding offset makes the rows start off the 16-byte grid) and scattered to the space-to-depth image
        // with 4-byte LDS stores: sample u of a channel is row u % S, column u / S; a quad holds 4 rows of one column (S >= 4) and the
        // lane QC = S / 4 quads further holds the same rows of the next column, so one cross-lane word per plane pairs (column c,
        // column c + 1) into the two 32-bit words a lane stores; for S = 2 a quad is rows (0, 1) of two columns and needs no partner.
        // (The first version staged this tile with 4-byte loads, 2-byte LDS stores and a division per sample pair, not overlapped
        // with the MFMAs: 1.3 ms per launch against 0.1 ms of matrix work — profiles/EXPERIMENTS.md.)
        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        constexpr int NQ = 9;                                            // quads per thread and stage
        const int qpc = 33 << (sl - 1);                                  // quads per real channel and stage
        const unsigned magic = 0xffffffffu / (unsigned)qpc + 1u;         // Q / qpc = umulhi(Q, magic) for Q < 2^16
        const int total_q = nc * qpc;                                    // = 2112
        const int qcl = sl >= 2 ? sl - 2 : 0, qc = 1 << qcl;             // quads per column (S = 8: 2)
        // quads IB .. IE-1 of this thread: global -> registers.  (tv = the thread id behind an opaque copy made inside the stage loop: the
        // per-quad index arithmetic — ~10 VALU per quad — would otherwise be hoisted out of the loop as invariant and live across
        // the MFMA stage: that build spilled 52 registers)
        auto load_quads = [&](int tv, const float* shi, int th0, bool interior, f4u* vq, auto ib, auto ie) {
            constexpr int IB = decltype(ib)::value, IE = decltype(ie)::value;
#pragma unroll
            for (int i = IB; i < IE; ++i) {
                f4u v = {0.f, 0.f, 0.f, 0.f};
                if (i < NQ - 1 || wave == 0) {                           // (wave-uniform: the last round is wave 0 only)
                    const int Q = tv + 256 * i;
                    const int cl = (int)sat_umulhi((unsigned)Q, magic), q = Q - cl * qpc;
                    const int n = n_base + cl, t = th0 + 4 * q;
                    const float* sp = shi + (size_t)(n < p.N ? n : p.N - 1) * p.Thi;
                    // BRANCH-FREE loads (a per-lane fallback branch around each load made the compiler wait for every quad in turn): the
                    // stage is either interior for every quad (block-uniform: all but the first and the last two stages of a batch
                    // item) — one dword-aligned 16-byte load — or an edge stage: four clamped 4-byte loads and selects
                    if (interior) {
                        v = *reinterpret_cast<const f4u*>(sp + t);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int te = t + e;
                            const int tc = te < 0 ? 0 : (te < p.Thi ? te : p.Thi - 1);
                            const float x = sp[tc];
                            v[e] = (te == tc) ? x : 0.0f;
                        }
                    }
                    if (n >= p.N) v = f4u{0.f, 0.f, 0.f, 0.f};
                }
                vq[i - IB] = v;
            }
        };
        // ... -> SnakeBeta, hi / lo split, space-to-depth scatter into the stage image
        auto store_quads = [&](int tv, const f4u* vq, auto ib, auto ie) {
            constexpr int IB = decltype(ib)::value, IE = decltype(ie)::value;
#pragma unroll
            for (int i = IB; i < IE; ++i) {
                const int Q = tv + 256 * i;
                if (i == NQ - 1 && Q >= total_q) break;                  // (wave-uniform: the last round is wave 0 only)
                const int cl = (int)sat_umulhi((unsigned)Q, magic), q = Q - cl * qpc;
                f4u o = vq[i - IB];
                if (snake_hi) {
                    const float sa = sn_a[cl], sib = sn_ib[cl];
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = sat_snake(o[e], sa, sib);
                }
                uint32_t h01, l01, h23, l23;
                sat_split2_pk(o[0], o[1], &h01, &l01);
                sat_split2_pk(o[2], o[3], &h23, &l23);
                int row, col;
                uint32_t wh0, wh1, wl0, wl1;
                if (sl == 1) {                                           // rows (0, 1, 0, 1) of columns 2q, 2q + 1
                    row = cl << 1;
                    col = 2 * q;
                    wh0 = (h01 & 0xffffu) | (h23 << 16); wh1 = (h01 >> 16) | (h23 & 0xffff0000u);
                    wl0 = (l01 & 0xffffu) | (l23 << 16); wl1 = (l01 >> 16) | (l23 & 0xffff0000u);
                } else {
                    const int colq = q >> qcl, part = q & (qc - 1);      // this quad: rows 4 part .. + 4 of column colq
                    const bool odd = colq & 1;
                    const uint32_t rh = __shfl_xor(odd ? h01 : h23, qc), rl = __shfl_xor(odd ? l01 : l23, qc);
                    row = (cl << sl) + 4 * part + (odd ? 2 : 0);
                    col = colq & ~1;
                    if (!odd) {                                          // rows 0, 1: (mine, partner's)
                        wh0 = (h01 & 0xffffu) | (rh << 16); wh1 = (h01 >> 16) | (rh & 0xffff0000u);
                        wl0 = (l01 & 0xffffu) | (rl << 16); wl1 = (l01 >> 16) | (rl & 0xffff0000u);
                    } else {                                             // rows 2, 3: (partner's, mine)
                        wh0 = (rh & 0xffffu) | (h23 << 16); wh1 = (rh >> 16) | (h23 & 0xffff0000u);
                        wl0 = (rl & 0xffffu) | (l23 << 16); wl1 = (rl >> 16) | (l23 & 0xffff0000u);
                    }
                }
                *reinterpret_cast<uint32_t*>(&hi_lds[0][row][col]) = wh0;
                *reinterpret_cast<uint32_t*>(&hi_lds[0][row + 1][col]) = wh1;
                *reinterpret_cast<uint32_t*>(&hi_lds[1][row][col]) = wl0;
                *reinterpret_cast<uint32_t*>(&hi_lds[1][row + 1][col]) = wl1;
            }
        };
        using I0 = std::integral_constant<int, 0>; using IN = std::integral_constant<int, NQ>;
        f4u vq[NQ];
        auto issue_hi = [&](int ch) {
            const int b = ch / p.nT;
            const int tt0 = (ch - b * p.nT) * SAT_WS_TT;
            const int th0 = (tt0 << sl) - p.pad;
            const bool interior = th0 >= 0 && th0 + (66 << sl) <= p.Thi;     // every sample of the stage's 66 columns exists
            int tv = tid;
#if !defined(SAT_HIPEMU)
            asm volatile("" : "+v"(tv));
#endif
            load_quads(tv, p.hi + (size_t)b * p.N * p.Thi, th0, interior, vq, I0{}, IN{});
        };
#if !defined(SAT_WGS_NOCARRY)
        issue_hi(c_begin);
#endif
        for (int ch = c_begin; ch < c_end; ++ch) {
            const int b = ch / p.nT;
            const int tt0 = (ch - b * p.nT) * SAT_WS_TT;
            const float* slo = p.lo + (size_t)b * p.M * p.Tlo;
            int tv = tid;
#if !defined(SAT_HIPEMU)
            asm volatile("" : "+v"(tv));
#endif
            const int row0 = tv >> 4, c4 = (tv & 15) * 4;
            float4 vlo[8];
#if defined(SAT_WGS_NOCARRY)
            issue_hi(ch);
#endif
            {
                const bool t_ok = tt0 + c4 < p.Tlo;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int m = m0 + row0 + 16 * u;
                    const bool okm = t_ok && m < p.M;
                    const float4 a = *reinterpret_cast<const float4*>(slo + (size_t)(okm ? m : 0) * p.Tlo + (okm ? tt0 + c4 : 0));
                    vlo[u] = okm ? a : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            SAT_SCHED_FENCE();
            store_quads(tv, vq, I0{}, IN{});
            SAT_SCHED_FENCE();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = row0 + 16 * u;
                float4 q = vlo[u];
                if (want_rs) rsum[u] += sat_row16_sum(q);
                if (snake_lo) {
                    const float sa = sn_a[row], sib = sn_ib[row];
                    q.x = sat_snake(q.x, sa, sib); q.y = sat_snake(q.y, sa, sib);
                    q.z = sat_snake(q.z, sa, sib); q.w = sat_snake(q.w, sa, sib);
                }
                uint32_t h0, h1, l0, l1;
                sat_split2_pk(q.x, q.y, &h0, &l0);
                sat_split2_pk(q.z, q.w, &h1, &l1);
                *reinterpret_cast<u2*>(&lo_lds[0][row][c4]) = u2{h0, h1};
                *reinterpret_cast<u2*>(&lo_lds[1][row][c4]) = u2{l0, l1};
            }
            __syncthreads();
#if !defined(SAT_WGS_NOCARRY)
            if (ch + 1 < c_end) issue_hi(ch + 1);
#endif
            mfma_stage();
            __syncthreads();
        }
    } else
    for (int ch = c_begin; ch < c_end; ++ch) {
        const int b = ch / p.nT;
        const int tt0 = (ch - b * p.nT) * SAT_WS_TT;
        // ---- stage lo: 128 rows x 64 t ----
        {
            const float* src = p.lo + (size_t)b * p.M * p.Tlo;
            for (int half = 0; half < 2; ++half) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = tid + (half * 4 + u) * 256;
                    const int row = idx >> 4, c4 = (idx & 15) * 4;
                    const int m = m0 + row, t = tt0 + c4;
                    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (m < p.M) {
                        const float* s = src + (size_t)m * p.Tlo + t;
                        if (t + 3 < p.Tlo && ((p.Tlo & 3) == 0)) {
                            q = *reinterpret_cast<const float4*>(s);
                        } else {
                            if (t + 0 < p.Tlo) q.x = s[0];
                            if (t + 1 < p.Tlo) q.y = s[1];
                            if (t + 2 < p.Tlo) q.z = s[2];
                            if (t + 3 < p.Tlo) q.w = s[3];
                        }
                    }
                    v[u] = q;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = tid + (half * 4 + u) * 256;
                    const int row = idx >> 4, c4 = (idx & 15) * 4;
                    float4 q = v[u];
                    if (want_rs) rsum[half * 4 + u] += sat_row16_sum(q);
                    if (snake_lo) {
                        const float sa = sn_a[row], sib = sn_ib[row];
                        q.x = sat_snake(q.x, sa, sib);
                        q.y = sat_snake(q.y, sa, sib);
                        q.z = sat_snake(q.z, sa, sib);
                        q.w = sat_snake(q.w, sa, sib);
                    }
                    uint32_t h0, h1, l0, l1;
                    sat_split2_pk(q.x, q.y, &h0, &l0);
                    sat_split2_pk(q.z, q.w, &h1, &l1);
                    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
                    *reinterpret_cast<u2*>(&lo_lds[0][row][c4]) = u2{h0, h1};
                    *reinterpret_cast<u2*>(&lo_lds[1][row][c4]) = u2{l0, l1};
                }
            }
        }
        // ---- stage hi: nc real channels x rs contiguous samples from tt0*S - pad, scattered to (n*S + u%S, u/S) ----
        {
            const float* src = p.hi + (size_t)b * p.N * p.Thi;
            const int th0 = (tt0 << sl) - p.pad;
            for (int base = tid; base < total_units; base += 8 * 256) {
                float v[8][2];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 256;
                    const int cl = idx / up, uu = (idx - cl * up) * 2;
                    const int n = n_base + cl, t = th0 + uu;
                    const bool ok = idx < total_units && n < p.N;
                    const float* s = src + (size_t)(ok ? n : 0) * p.Thi;
                    v[u][0] = (ok && t >= 0 && t < p.Thi) ? s[t] : 0.0f;
                    v[u][1] = (ok && t + 1 >= 0 && t + 1 < p.Thi) ? s[t + 1] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 256;
                    if (idx < total_units) {
                        const int cl = idx / up, uu = (idx - cl * up) * 2;
                        float o0 = v[u][0], o1 = v[u][1];
                        if (snake_hi) {
                            const float sa = sn_a[cl], sib = sn_ib[cl];
                            o0 = sat_snake(o0, sa, sib);
                            o1 = sat_snake(o1, sa, sib);
                        }
                        uint32_t h, l;
                        sat_split2_pk(o0, o1, &h, &l);
                        if (sl == 0) {
                            *reinterpret_cast<uint32_t*>(&hi_lds[0][cl][uu]) = h;
                            *reinterpret_cast<uint32_t*>(&hi_lds[1][cl][uu]) = l;
                        } else {
                            const int row = (cl << sl) + (uu & (S - 1)), col = uu >> sl;   // uu, S even: uu+1 is the next row
                            hi_lds[0][row][col] = (short)(h & 0xffffu);
                            hi_lds[0][row + 1][col] = (short)(h >> 16);
                            hi_lds[1][row][col] = (short)(l & 0xffffu);
                            hi_lds[1][row + 1][col] = (short)(l >> 16);
                        }
                    }
                }
            }
        }
        __syncthreads();
        mfma_stage();
        __syncthreads();
    }

    if (want_rs && (tid & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + (tid >> 4) + 16 * i;
            if (m < p.M) p.rowsum[(size_t)m * p.nsplit + split] = rsum[i];
        }
    }
    if (wave_on) {
        float* ob = p.out + (size_t)split * p.so_split;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int v = v0 + v_w + ni * 32 + l31;
                const int n = v >> sl, ph = v & (S - 1);
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + m_w + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (m < p.M && n < p.N)
                            ob[(size_t)m * p.so_m + (size_t)n * p.so_n + (size_t)(j * S + ph) * p.so_k] = acc[mi][ni][j][r];
                    }
            }
    }
}

static bool sat_wgs_plan(int B, int M, int N, int Tlo, int K, int stride, SatWgBfPlan* pl, int* s_log2, int* nt) {
    if (B <= 0 || M <= 0 || N <= 0 || Tlo <= 0) return false;
    if (K == 1 && stride == 1) { *s_log2 = 0; *nt = 1; }
    else if (K == 2 * stride && stride >= 2 && stride <= 64 && (stride & (stride - 1)) == 0) {
        int l = 0;
        while ((1 << l) < stride) ++l;
        *s_log2 = l;
        *nt = 2;
    } else return false;
    pl->nT = sat_cdiv(Tlo, SAT_WS_TT);
    pl->nchunks = B * pl->nT;
    const int tiles = sat_cdiv(M, SAT_CO_T) * sat_cdiv(N << *s_log2, SAT_CO_T);
    int want = 512 / tiles;            // 2 workgroups per CU in flight: one full wave of the grid (never a partial second one)
    if (want > pl->nchunks) want = pl->nchunks;
    if (want < 1) want = 1;
    if (want > 1024) want = 1024;
    pl->cps = sat_cdiv(pl->nchunks, want);
    pl->nsplit = sat_cdiv(pl->nchunks, pl->cps);
    return true;
}
extern "C" int sat_conv_wgrad_bf16x3_nsplit(int B, int M, int N, int Tlo, int K, int stride) {
    SatWgBfPlan pl;
    int sl, nt;
    return sat_wgs_plan(B, M, N, Tlo, K, stride, &pl, &sl, &nt) ? pl.nsplit : -1;
}
// Same contract as sat_conv_wgrad for K == 1 (stride 1) and K == 2*stride (power-of-two stride, dilation 1), on the
// bf16 matrix cores at fp32 accuracy.  Slab stride M*N*K; nsplit from sat_conv_wgrad_bf16x3_nsplit (-1: unsupported).
extern "C" int sat_conv_wgrad_bf16x3(const float* lo, const float* hi, const float* alpha, const float* beta, int snake_on,
                                     float* partial, long long so_m, long long so_n, long long so_k, int B, int M, int N,
                                     int Tlo, int Thi, int K, int stride, int pad, float* lo_rowsum, void* stream) {
    SatWgBfPlan pl;
    int sl, nt;
    if (!sat_wgs_plan(B, M, N, Tlo, K, stride, &pl, &sl, &nt)) {
        sat_set_error("sat_conv_wgrad_bf16x3: needs K == 1 (stride 1) or K == 2*stride with a power-of-two stride");
        return 1;
    }
    if (Thi <= 0 || pad < 0) { sat_set_error("sat_conv_wgrad_bf16x3: bad shape"); return 1; }
    if (alpha && (snake_on != 1 && snake_on != 2)) { sat_set_error("sat_conv_wgrad_bf16x3: snake_on must be 1 (lo) or 2 (hi)"); return 1; }
    if ((alpha == nullptr) != (beta == nullptr)) { sat_set_error("sat_conv_wgrad_bf16x3: alpha/beta must both be given"); return 1; }
    SatWgSmallParams p{lo, hi, alpha, beta, partial, (long long)M * N * K, so_m, so_n, so_k,
                       B, M, N, Tlo, Thi, pad, alpha ? snake_on : 0, sl, pl.cps, pl.nchunks, pl.nT, lo_rowsum, pl.nsplit};
    dim3 grid(sat_cdiv(M, SAT_CO_T), sat_cdiv(N << sl, SAT_CO_T), pl.nsplit);
    if (nt == 1) SAT_LAUNCH(sat_wgrad_small_bf16x3_kernel<1>, grid, dim3(256), stream, p);
    else SAT_LAUNCH(sat_wgrad_small_bf16x3_kernel<2>, grid, dim3(256), stream, p);
    return sat_check_launch("sat_conv_wgrad_bf16x3");
}
