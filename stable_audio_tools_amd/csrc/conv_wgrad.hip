// conv_wgrad.hip — weight gradients of every Oobleck conv as one split-K GEMM family on the fp32
// matrix cores.
//
//   dW[m][n][k] = sum_b sum_t  actA(lo[b][m][t]) * actB(hi[b][n][t*s + k*d - pad])
//
//   * WNConv1d (stride 1 or s):  lo = dL/dy (Cout rows), hi = conv input (Cin rows)   -> dW[co][ci][k]
//   * WNConvTranspose1d:         lo = conv input (Cin rows), hi = dL/dy (Cout rows)   -> dW[ci][co][k]
// (torch weight layouts: autoencoders.py:23-27).  SnakeBeta of the conv input is recomputed while
// the tile is staged (the activation was never stored), on whichever operand is the conv input.
//
// GEMM view: the MFMA reduction dim is TIME (the two k-slots of 32x32x2 are two consecutive time
// steps), rows = lo channels, cols = hi channels; a wave keeps NSUB x KT accumulator tiles
// (n sub-tiles x taps) so that one staged (lo, hi) time slab feeds every tap.  The (b, t) range is
// split across gridDim.z; partial slabs are summed by sat_reduce_splits (deterministic, no atomics).
#include "conv_common.h"

#define SAT_WG_TT 32
#define SAT_WG_LO_RL (SAT_WG_TT + 1)
#define SAT_WG_HI_FLOATS 9216

struct SatWgradParams {
    const float* lo;     // (B, M, Tlo)
    const float* hi;     // (B, N, Thi)
    const float* alpha;  // snake log-params of the conv input (channels of lo if snake_on==1, of hi if 2)
    const float* beta;
    float* out;          // partial slabs
    long long so_split, so_m, so_n, so_k;
    int B, M, N, Tlo, Thi;
    int K, stride, dil, pad;
    int snake_on;
    int chunks_per_split, nchunks, nT;
    int hi_rl;           // hi slab row length (odd)
    int hi_nj;           // hi samples staged per row per chunk
    int ngroups;         // tap groups
};

template <int NSUB, int KT>
__global__ void __launch_bounds__(256) sat_conv_wgrad_kernel(SatWgradParams p) {
    constexpr int N_T = NSUB * 32;
    __shared__ float lo_lds[SAT_CO_T][SAT_WG_LO_RL];
    __shared__ float hi_lds[SAT_WG_HI_FLOATS];
    __shared__ float sn_a[SAT_CO_T], sn_ib[SAT_CO_T];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int m0 = blockIdx.x * SAT_CO_T;
    const int ntile = blockIdx.y / p.ngroups, grp = blockIdx.y - ntile * p.ngroups;
    const int n0 = ntile * N_T;
    const int k0 = grp * KT;
    int kcount = p.K - k0;
    if (kcount > KT) kcount = KT;
    const int m_w = wave * 32;
    const int S = p.stride, dil = p.dil;
    const int RLH = p.hi_rl;

    f32x16 acc[NSUB][KT];
#pragma unroll
    for (int i = 0; i < NSUB; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    if (p.snake_on) {
        const int nch = (p.snake_on == 1) ? p.M : p.N;
        const int c0 = (p.snake_on == 1) ? m0 : n0;
        if (tid < SAT_CO_T) {
            const int c = c0 + tid;
            float sa = 1.f, sib = 0.f;
            if (c < nch) {
                sa = expf(p.alpha[c]);
                sib = 1.0f / (expf(p.beta[c]) + 1e-9f);
            }
            sn_a[tid] = sa;
            sn_ib[tid] = sib;
        }
    }
    __syncthreads();

    const bool wave_on = (m0 + m_w) < p.M;
    const int c_begin = blockIdx.z * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > p.nchunks) c_end = p.nchunks;

    for (int ch = c_begin; ch < c_end; ++ch) {
        const int b = ch / p.nT;
        const int tt0 = (ch - b * p.nT) * SAT_WG_TT;
        // ---- stage lo: 128 rows x TT ----
        {
            // all 16 loads of a thread are issued before the first is consumed (one HBM round trip per stage)
            const float* lob = p.lo + (size_t)b * p.M * p.Tlo;
            float v[SAT_CO_T * SAT_WG_TT / 256];
#pragma unroll
            for (int u = 0; u < SAT_CO_T * SAT_WG_TT / 256; ++u) {
                const int idx = tid + u * 256;
                const int row = idx / SAT_WG_TT, col = idx - row * SAT_WG_TT;
                const int m = m0 + row, t = tt0 + col;
                v[u] = (m < p.M && t < p.Tlo) ? lob[(size_t)m * p.Tlo + t] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < SAT_CO_T * SAT_WG_TT / 256; ++u) {
                const int idx = tid + u * 256;
                const int row = idx / SAT_WG_TT, col = idx - row * SAT_WG_TT;
                lo_lds[row][col] = (p.snake_on == 1) ? sat_snake(v[u], sn_a[row], sn_ib[row]) : v[u];
            }
        }
        // ---- stage hi: N_T rows x hi_nj, in batches of 8 loads per thread ----
        {
            const float* hib = p.hi + (size_t)b * p.N * p.Thi;
            const int th0 = tt0 * S - p.pad + k0 * dil;
            const int nj = p.hi_nj;
            const int total = N_T * nj;
            for (int base = tid; base < total; base += 8 * 256) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 256;
                    const int row = idx / nj, col = idx - row * nj;
                    const int n = n0 + row, t = th0 + col;
                    v[u] = (idx < total && n < p.N && t >= 0 && t < p.Thi) ? hib[(size_t)n * p.Thi + t] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + u * 256;
                    if (idx < total) {
                        const int row = idx / nj, col = idx - row * nj;
                        hi_lds[row * RLH + col] = (p.snake_on == 2) ? sat_snake(v[u], sn_a[row], sn_ib[row]) : v[u];
                    }
                }
            }
        }
        __syncthreads();
        if (wave_on) {
            if (kcount == KT) {
                // branch-free body (a hand-rolled register prefetch here measured 1.7x SLOWER on MI355X:
                // 47 -> 28 TFLOP/s; the compiler's own schedule of the unrolled loop is kept)
#pragma unroll 4
                for (int tp = 0; tp < SAT_WG_TT; tp += 2) {
                    const float av = lo_lds[m_w + l31][tp + hi];
#pragma unroll
                    for (int ns = 0; ns < NSUB; ++ns) {
                        const float* hr = hi_lds + (ns * 32 + l31) * RLH + (tp + hi) * S;
#pragma unroll
                        for (int kk = 0; kk < KT; ++kk) acc[ns][kk] = sat_mfma_32x32x2_f32(av, hr[kk * dil], acc[ns][kk]);
                    }
                }
            } else {
                for (int tp = 0; tp < SAT_WG_TT; tp += 2) {
                    const float av = lo_lds[m_w + l31][tp + hi];
#pragma unroll
                    for (int ns = 0; ns < NSUB; ++ns) {
                        const float* hr = hi_lds + (ns * 32 + l31) * RLH + (tp + hi) * S;
#pragma unroll
                        for (int kk = 0; kk < KT; ++kk) {
                            if (kk < kcount) acc[ns][kk] = sat_mfma_32x32x2_f32(av, hr[kk * dil], acc[ns][kk]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    if (wave_on) {
        float* ob = p.out + (size_t)blockIdx.z * p.so_split;
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) {
            const int n = n0 + ns * 32 + l31;
#pragma unroll
            for (int kk = 0; kk < KT; ++kk) {
                if (kk >= kcount) break;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + m_w + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (m < p.M && n < p.N)
                        ob[(size_t)m * p.so_m + (size_t)n * p.so_n + (size_t)(k0 + kk) * p.so_k] = acc[ns][kk][r];
                }
            }
        }
    }
}

// out[i] = sum_z partial[z*count + i]
struct SatReduceParams {
    const float* partial;
    float* out;
    long long count;
    int nsplit;
    float scale;
    int accumulate;
};
__global__ void __launch_bounds__(256) sat_reduce_splits_kernel(SatReduceParams p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.count) return;
    float s = 0.f;
    for (int z = 0; z < p.nsplit; ++z) s += p.partial[(size_t)z * p.count + i];
    s *= p.scale;
    if (p.accumulate) s += p.out[i];
    p.out[i] = s;
}
// 16-byte version (count % 4 == 0, 16-byte aligned buffers): a thread owns 4 consecutive elements; the slabs are walked 4 at a time
// with independent accumulators so that four 1-KiB wave loads are in flight per step.  The summation ORDER is fixed (slab 0, 1, 2, ...
// folded pairwise the same way every launch): deterministic.
__global__ void __launch_bounds__(256) sat_reduce_splits_vec_kernel(SatReduceParams p) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long nq = p.count >> 2;
    if (q >= nq) return;
    const f32x4* src = reinterpret_cast<const f32x4*>(p.partial) + q;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    int z = 0;
    for (; z + 4 <= p.nsplit; z += 4) {
        const f32x4 v0 = src[(size_t)(z + 0) * nq], v1 = src[(size_t)(z + 1) * nq], v2 = src[(size_t)(z + 2) * nq], v3 = src[(size_t)(z + 3) * nq];
        a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; z < p.nsplit; ++z) a0 += src[(size_t)z * nq];
    f32x4 s = ((a0 + a1) + (a2 + a3)) * p.scale;
    f32x4* out = reinterpret_cast<f32x4*>(p.out) + q;
    if (p.accumulate) s += *out;
    *out = s;
}

extern "C" int sat_reduce_splits(const float* partial, float* out, long long count, int nsplit, float scale,
                                 int accumulate, void* stream) {
    if (count <= 0 || nsplit <= 0) { sat_set_error("sat_reduce_splits: empty"); return 1; }
    SatReduceParams p{partial, out, count, nsplit, scale, accumulate};
    if ((count & 3) == 0 && (((uintptr_t)partial | (uintptr_t)out) & 15) == 0) {
        dim3 gridv((unsigned)sat_cdivll(count >> 2, 256));
        SAT_LAUNCH(sat_reduce_splits_vec_kernel, gridv, dim3(256), stream, p);
        return sat_check_launch("sat_reduce_splits");
    }
    dim3 grid((unsigned)sat_cdivll(count, 256));
    SAT_LAUNCH(sat_reduce_splits_kernel, grid, dim3(256), stream, p);
    return sat_check_launch("sat_reduce_splits");
}

// per-channel sum over (b, t) of a (B, C, T) tensor: partial[split][c]  (bias gradients)
struct SatRowsumParams {
    const float* x;
    float* partial;
    int B, C, T, nsplit, tper;
};
__global__ void __launch_bounds__(256) sat_rowsum_kernel(SatRowsumParams p) {
    __shared__ float red[4];
    const int c = blockIdx.x, z = blockIdx.y;
    const int t_begin = z * p.tper;                     // tper is a multiple of 4
    int t_end = t_begin + p.tper;
    if (t_end > p.T) t_end = p.T;
    float s = 0.f;
    for (int b = 0; b < p.B; ++b) {
        const float* xr = p.x + ((size_t)b * p.C + c) * p.T;
        if ((p.T & 3) == 0) {                           // rows are 16-byte aligned: float4 loads
            for (int t = t_begin + 4 * threadIdx.x; t < t_end; t += 1024) {
                const float4 v = *reinterpret_cast<const float4*>(xr + t);
                s += (v.x + v.y) + (v.z + v.w);
            }
        } else {
            for (int t = t_begin + threadIdx.x; t < t_end; t += 256) s += xr[t];
        }
    }
    s = sat_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) p.partial[(size_t)c * p.nsplit + z] = red[0] + red[1] + red[2] + red[3];
}

extern "C" int sat_rowsum_nsplit(int T) {
    int n = sat_cdiv(T, 16384);
    return n < 1 ? 1 : n;
}
extern "C" int sat_rowsum(const float* x, float* partial, int B, int C, int T, void* stream) {
    if (B <= 0 || C <= 0 || T <= 0) { sat_set_error("sat_rowsum: empty shape"); return 1; }
    const int nsplit = sat_rowsum_nsplit(T);
    SatRowsumParams p{x, partial, B, C, T, nsplit, sat_cdiv(sat_cdiv(T, nsplit), 4) * 4};
    SAT_LAUNCH(sat_rowsum_kernel, dim3(C, nsplit), dim3(256), stream, p);
    return sat_check_launch("sat_rowsum");
}

// ---------------------------------------------------------------------------------------------
struct SatWgPlan {
    int nsub, kt, ngroups, nsplit, cps, nchunks, nT, hi_rl, hi_nj;
};
static int sat_wgrad_plan(int B, int M, int N, int Tlo, int K, int stride, int dil, SatWgPlan* pl) {
    if (K == 1) { pl->nsub = 4; pl->kt = 1; }
    else if (K == 3) { pl->nsub = 2; pl->kt = 3; }
    else if (K == 4) { pl->nsub = 2; pl->kt = 4; }
    else if (K == 7) { pl->nsub = 1; pl->kt = 7; }
    else { pl->nsub = 1; pl->kt = 8; }
    pl->ngroups = sat_cdiv(K, pl->kt);
    int kspan = K < pl->kt ? K : pl->kt;
    pl->hi_nj = (SAT_WG_TT - 1) * stride + (kspan - 1) * dil + 1;
    pl->hi_rl = pl->hi_nj | 1;
    if (pl->nsub * 32 * pl->hi_rl > SAT_WG_HI_FLOATS) return 1;
    pl->nT = sat_cdiv(Tlo, SAT_WG_TT);
    pl->nchunks = B * pl->nT;
    const int tiles = sat_cdiv(M, SAT_CO_T) * sat_cdiv(N, pl->nsub * 32) * pl->ngroups;
    int want = sat_cdiv(1536, tiles);          // ~6 workgroups per CU in flight overall
    if (want > pl->nchunks) want = pl->nchunks;
    if (want < 1) want = 1;
    if (want > 512) want = 512;
    pl->cps = sat_cdiv(pl->nchunks, want);
    pl->nsplit = sat_cdiv(pl->nchunks, pl->cps);
    return 0;
}

extern "C" int sat_conv_wgrad_nsplit(int B, int M, int N, int Tlo, int K, int stride, int dil) {
    SatWgPlan pl;
    if (sat_wgrad_plan(B, M, N, Tlo, K, stride, dil, &pl)) return -1;
    return pl.nsplit;
}

extern "C" int sat_conv_wgrad(const float* lo, const float* hi, const float* alpha, const float* beta,
                              int snake_on, float* partial, long long so_m, long long so_n, long long so_k,
                              int B, int M, int N, int Tlo, int Thi, int K, int stride, int dil, int pad,
                              void* stream) {
    if (B <= 0 || M <= 0 || N <= 0 || Tlo <= 0 || Thi <= 0 || K <= 0) { sat_set_error("sat_conv_wgrad: empty shape"); return 1; }
    if (snake_on && (!alpha || !beta)) { sat_set_error("sat_conv_wgrad: snake requested without parameters"); return 1; }
    SatWgPlan pl;
    if (sat_wgrad_plan(B, M, N, Tlo, K, stride, dil, &pl)) { sat_set_error("sat_conv_wgrad: receptive field too large for the LDS slab"); return 1; }
    SatWgradParams p;
    p.lo = lo; p.hi = hi; p.alpha = alpha; p.beta = beta; p.out = partial;
    p.so_split = (long long)M * N * K; p.so_m = so_m; p.so_n = so_n; p.so_k = so_k;
    p.B = B; p.M = M; p.N = N; p.Tlo = Tlo; p.Thi = Thi; p.K = K; p.stride = stride; p.dil = dil; p.pad = pad;
    p.snake_on = snake_on;
    p.chunks_per_split = pl.cps; p.nchunks = pl.nchunks; p.nT = pl.nT;
    p.hi_rl = pl.hi_rl; p.hi_nj = pl.hi_nj; p.ngroups = pl.ngroups;
    dim3 grid(sat_cdiv(M, SAT_CO_T), sat_cdiv(N, pl.nsub * 32) * pl.ngroups, pl.nsplit);
    if (pl.nsub == 4) SAT_LAUNCH((sat_conv_wgrad_kernel<4, 1>), grid, dim3(256), stream, p);
    else if (pl.nsub == 2 && pl.kt == 3) SAT_LAUNCH((sat_conv_wgrad_kernel<2, 3>), grid, dim3(256), stream, p);
    else if (pl.nsub == 2 && pl.kt == 4) SAT_LAUNCH((sat_conv_wgrad_kernel<2, 4>), grid, dim3(256), stream, p);
    else if (pl.kt == 7) SAT_LAUNCH((sat_conv_wgrad_kernel<1, 7>), grid, dim3(256), stream, p);
    else SAT_LAUNCH((sat_conv_wgrad_kernel<1, 8>), grid, dim3(256), stream, p);
    return sat_check_launch("sat_conv_wgrad");
}
