// dit_ops.hip — the HBM-bound operators of the DiT block (stable_audio_tools/models/transformer.py):
//   sat_layernorm_fwd/bwd   LayerNorm.forward :236-241 (gamma, zero beta buffer, eps 1e-5, fp32 math) fused with
//                           the adaLN modulation x*(1+scale)+shift of TransformerBlock.forward :682, :697
//   sat_rope_tables / sat_rope_apply   RotaryEmbedding.forward :125-138 + apply_rotary_pos_emb :155-174
//                           (fp32, first rot_dim dims of every head, NeoX half rotation), applied in place
//                           on the q and k slices of the fused qkv projection
//   sat_swiglu_fwd/bwd      GLU.forward :274-275  (x * silu(gate))
//   sat_gate_residual       x * sigmoid(1 - gate) + residual   :684-686, :699-701
// One wave per token row, wavefront-shuffle reductions, 16-byte accesses where the layout allows.
// dtype: 0 = fp32, 1 = bf16 tensors (statistics and math always fp32).
#include <cstdio>
#include "sat_device.h"

template <typename T> struct SatIO;
template <> struct SatIO<float> {
    static SAT_DEVICE float ld(const void* p, long long i) { return ((const float*)p)[i]; }
    static SAT_DEVICE void st(void* p, long long i, float v) { ((float*)p)[i] = v; }
};
template <> struct SatIO<short> {
    static SAT_DEVICE float ld(const void* p, long long i) { return sat_bf16_to_f32(((const short*)p)[i]); }
    static SAT_DEVICE void st(void* p, long long i, float v) { ((short*)p)[i] = sat_f32_to_bf16(v); }
};

// ------------------------------------------------------------------------------------------------
// LayerNorm (+ adaLN modulate)
// ------------------------------------------------------------------------------------------------
struct SatLnParams {
    const void* x;        // (rows, D)
    const float* gamma;   // (D) fp32 master
    const float* beta;    // (D) or null
    const void* scale;    // (B, D) or null : y = ln * (1 + scale[b]) + shift[b]
    const void* shift;    // (B, D)
    void* y;              // (rows, D)
    float* mean;          // (rows) saved for backward (may be null)
    float* rstd;          // (rows)
    // backward
    const void* dy;       // (rows, D)
    void* dx;             // (rows, D)
    float* part;          // [3][nblocks_y][D] partial column sums: d_gamma, d_scale(per batch), d_shift(per batch)
    long long mod_stride; // element stride between batches in scale/shift (>= D)
    int rows, D, rows_per_batch;
    float eps;
    // fp8 output (sat_layernorm_fwd_fp8): e4m3 bytes (rows, D) with one dynamic scale per row
    uint8_t* q;
    float* qscale;        // (rows): row max / 448
    // backward: optional addend of dx, (rows, D) in the activation dtype — the gradient that reached x along the residual path
    const void* dres;
};

template <typename T>
__global__ void __launch_bounds__(256) sat_layernorm_fwd_kernel(SatLnParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= p.rows) return;   // whole wave exits together; no block barrier below
    const long long base = (long long)row * p.D;
    float s = 0.f;
    for (int i = lane; i < p.D; i += 64) s += SatIO<T>::ld(p.x, base + i);
    const float mean = sat_wave_sum(s) / (float)p.D;
    float v = 0.f;
    for (int i = lane; i < p.D; i += 64) {
        const float d = SatIO<T>::ld(p.x, base + i) - mean;
        v += d * d;
    }
    const float rstd = 1.0f / sqrtf(sat_wave_sum(v) / (float)p.D + p.eps);
    const long long mb = p.scale ? (long long)(row / p.rows_per_batch) * p.mod_stride : 0;
    for (int i = lane; i < p.D; i += 64) {
        float o = (SatIO<T>::ld(p.x, base + i) - mean) * rstd * p.gamma[i];
        if (p.beta) o += p.beta[i];
        if (p.scale) o = o * (1.0f + SatIO<T>::ld(p.scale, mb + i)) + SatIO<T>::ld(p.shift, mb + i);
        SatIO<T>::st(p.y, base + i, o);
    }
    if (lane == 0 && p.mean) {
        p.mean[row] = mean;
        p.rstd[row] = rstd;
    }
}

// dx: one wave per row.
template <typename T>
__global__ void __launch_bounds__(256) sat_layernorm_bwd_dx_kernel(SatLnParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= p.rows) return;
    const long long base = (long long)row * p.D;
    const float mean = p.mean[row], rstd = p.rstd[row];
    const long long mb = p.scale ? (long long)(row / p.rows_per_batch) * p.mod_stride : 0;
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < p.D; i += 64) {
        float g = SatIO<T>::ld(p.dy, base + i);
        if (p.scale) g *= 1.0f + SatIO<T>::ld(p.scale, mb + i);
        g *= p.gamma[i];
        const float xh = (SatIO<T>::ld(p.x, base + i) - mean) * rstd;
        s1 += g;
        s2 += g * xh;
    }
    s1 = sat_wave_sum(s1) / (float)p.D;
    s2 = sat_wave_sum(s2) / (float)p.D;
    for (int i = lane; i < p.D; i += 64) {
        float g = SatIO<T>::ld(p.dy, base + i);
        if (p.scale) g *= 1.0f + SatIO<T>::ld(p.scale, mb + i);
        g *= p.gamma[i];
        const float xh = (SatIO<T>::ld(p.x, base + i) - mean) * rstd;
        float o = rstd * (g - s1 - xh * s2);
        if (p.dres) o += SatIO<T>::ld(p.dres, base + i);
        SatIO<T>::st(p.dx, base + i, o);
    }
}

// ---- vector paths: a wave keeps its whole row in registers (16-byte loads / stores), used when D is a multiple of
// 64 lanes x 16 bytes and every pointer is 16-byte aligned (the DiT shapes: d = 1536).  One pass over HBM.
template <typename T> struct SatVec;
template <> struct SatVec<float> {
    static constexpr int N = 4, MAXC = 16;            // floats per 16 bytes; row chunks of 256 elements, D <= 4096
    static SAT_DEVICE void ld(const void* p, long long i, float* o) {
        const f32x4 v = *reinterpret_cast<const f32x4*>((const float*)p + i);
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
    static SAT_DEVICE void st(void* p, long long i, const float* o) {
        *reinterpret_cast<f32x4*>((float*)p + i) = f32x4{o[0], o[1], o[2], o[3]};
    }
};
template <> struct SatVec<short> {
    static constexpr int N = 8, MAXC = 8;             // bf16 per 16 bytes; row chunks of 512 elements, D <= 4096
    static SAT_DEVICE void ld(const void* p, long long i, float* o) {
        const u32x4 v = *reinterpret_cast<const u32x4*>((const short*)p + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[2 * j] = __builtin_bit_cast(float, v[j] << 16);
            o[2 * j + 1] = __builtin_bit_cast(float, v[j] & 0xffff0000u);
        }
    }
    static SAT_DEVICE void st(void* p, long long i, const float* o) {
        u32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = (uint32_t)(uint16_t)sat_f32_to_bf16(o[2 * j]) | ((uint32_t)(uint16_t)sat_f32_to_bf16(o[2 * j + 1]) << 16);
        *reinterpret_cast<u32x4*>((short*)p + i) = v;
    }
};

template <typename T>
__global__ void __launch_bounds__(256) sat_layernorm_fwd_vec_kernel(SatLnParams p) {
    constexpr int N = SatVec<T>::N, MAXC = SatVec<T>::MAXC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= p.rows) return;   // whole wave exits together; no block barrier below
    const long long base = (long long)row * p.D;
    const int nc = p.D / (64 * N);
    float xs[MAXC][N];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nc) {
            SatVec<T>::ld(p.x, base + (c * 64 + lane) * N, xs[c]);
#pragma unroll
            for (int j = 0; j < N; ++j) s += xs[c][j];
        }
    }
    const float mean = sat_wave_sum(s) / (float)p.D;
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nc) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float d = xs[c][j] - mean;
                v += d * d;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(sat_wave_sum(v) / (float)p.D + p.eps);
    const long long mb = p.scale ? (long long)(row / p.rows_per_batch) * p.mod_stride : 0;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nc) {
            const int i0 = (c * 64 + lane) * N;
            float g[N], o[N];
#pragma unroll
            for (int j = 0; j < N; j += 4) SatVec<float>::ld(p.gamma, i0 + j, g + j);
#pragma unroll
            for (int j = 0; j < N; ++j) o[j] = (xs[c][j] - mean) * rstd * g[j];
            if (p.beta) {
#pragma unroll
                for (int j = 0; j < N; j += 4) SatVec<float>::ld(p.beta, i0 + j, g + j);
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] += g[j];
            }
            if (p.scale) {
                float sc[N], sh[N];
                SatVec<T>::ld(p.scale, mb + i0, sc);
                SatVec<T>::ld(p.shift, mb + i0, sh);
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] = o[j] * (1.0f + sc[j]) + sh[j];
            }
            SatVec<T>::st(p.y, base + i0, o);
        }
    }
    if (lane == 0 && p.mean) {
        p.mean[row] = mean;
        p.rstd[row] = rstd;
    }
}

// LayerNorm straight to the fp8 operand of the projection that consumes it (round 4; the N = 6145 sampler): the normalised (and
// adaLN-modulated) row is kept in registers, its max |.| gives the row's dynamic scale, and the row leaves as e4m3 bytes + one fp32 scale
// (the GEMM's row_alpha) — the bf16 LayerNorm output (written, then re-read by the quantiser) and the quantiser's launch disappear.
template <typename T>
__global__ void __launch_bounds__(256) sat_layernorm_fwd_fp8_kernel(SatLnParams p) {
    constexpr int N = SatVec<T>::N, MAXC = SatVec<T>::MAXC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= p.rows) return;   // whole wave exits together; no block barrier below
    const long long base = (long long)row * p.D;
    const int nc = p.D / (64 * N);
    float xs[MAXC][N];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nc) {
            SatVec<T>::ld(p.x, base + (c * 64 + lane) * N, xs[c]);
#pragma unroll
            for (int j = 0; j < N; ++j) s += xs[c][j];
        }
    }
    const float mean = sat_wave_sum(s) / (float)p.D;
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nc) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float d = xs[c][j] - mean;
                v += d * d;
            }
        }
    }
    const float rstd = 1.0f / sqrtf(sat_wave_sum(v) / (float)p.D + p.eps);
    const long long mb = p.scale ? (long long)(row / p.rows_per_batch) * p.mod_stride : 0;
    float am = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nc) {
            const int i0 = (c * 64 + lane) * N;
            float g[N];
#pragma unroll
            for (int j = 0; j < N; j += 4) SatVec<float>::ld(p.gamma, i0 + j, g + j);
#pragma unroll
            for (int j = 0; j < N; ++j) xs[c][j] = (xs[c][j] - mean) * rstd * g[j];
            if (p.beta) {
#pragma unroll
                for (int j = 0; j < N; j += 4) SatVec<float>::ld(p.beta, i0 + j, g + j);
#pragma unroll
                for (int j = 0; j < N; ++j) xs[c][j] += g[j];
            }
            if (p.scale) {
                float sc[N], sh[N];
                SatVec<T>::ld(p.scale, mb + i0, sc);
                SatVec<T>::ld(p.shift, mb + i0, sh);
#pragma unroll
                for (int j = 0; j < N; ++j) xs[c][j] = xs[c][j] * (1.0f + sc[j]) + sh[j];
            }
#pragma unroll
            for (int j = 0; j < N; ++j) am = fmaxf(am, fabsf(xs[c][j]));
        }
    }
    for (int k = 32; k >= 1; k >>= 1) am = fmaxf(am, __shfl_xor(am, k));
    am = fmaxf(am, 1e-12f);
    const float qs = 448.0f / am;
    if (lane == 0) p.qscale[row] = am / 448.0f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nc) {
            uint8_t* dst = p.q + base + (c * 64 + lane) * N;
#pragma unroll
            for (int j = 0; j < N; j += 4)
                *reinterpret_cast<uint32_t*>(dst + j) = sat_f32x4_to_fp8(xs[c][j] * qs, xs[c][j + 1] * qs, xs[c][j + 2] * qs, xs[c][j + 3] * qs);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) sat_layernorm_bwd_dx_vec_kernel(SatLnParams p) {
    constexpr int N = SatVec<T>::N, MAXC = SatVec<T>::MAXC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= p.rows) return;
    const long long base = (long long)row * p.D;
    const int nc = p.D / (64 * N);
    const float mean = p.mean[row], rstd = p.rstd[row];
    const long long mb = p.scale ? (long long)(row / p.rows_per_batch) * p.mod_stride : 0;
    float gs[MAXC][N], xh[MAXC][N];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nc) {
            const int i0 = (c * 64 + lane) * N;
            float gam[N];
            SatVec<T>::ld(p.dy, base + i0, gs[c]);
            SatVec<T>::ld(p.x, base + i0, xh[c]);
#pragma unroll
            for (int j = 0; j < N; j += 4) SatVec<float>::ld(p.gamma, i0 + j, gam + j);
            if (p.scale) {
                float sc[N];
                SatVec<T>::ld(p.scale, mb + i0, sc);
#pragma unroll
                for (int j = 0; j < N; ++j) gs[c][j] *= 1.0f + sc[j];
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                gs[c][j] *= gam[j];
                xh[c][j] = (xh[c][j] - mean) * rstd;
                s1 += gs[c][j];
                s2 += gs[c][j] * xh[c][j];
            }
        }
    }
    s1 = sat_wave_sum(s1) / (float)p.D;
    s2 = sat_wave_sum(s2) / (float)p.D;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nc) {
            float o[N];
#pragma unroll
            for (int j = 0; j < N; ++j) o[j] = rstd * (gs[c][j] - s1 - xh[c][j] * s2);
            if (p.dres) {
                float r[N];
                SatVec<T>::ld(p.dres, base + (c * 64 + lane) * N, r);
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] += r[j];
            }
            SatVec<T>::st(p.dx, base + (c * 64 + lane) * N, o);
        }
    }
}

static bool sat_ln_vec_ok(const SatLnParams& p, int elem_bytes) {
    const int n = 16 / elem_bytes, maxc = elem_bytes == 4 ? 16 : 8;
    if (p.D % (64 * n) != 0 || p.D / (64 * n) > maxc) return false;
    const void* ptrs[] = {p.x, p.gamma, p.beta, p.scale, p.shift, p.y, p.dy, p.dx, p.dres};
    for (const void* q : ptrs)
        if (q && ((uintptr_t)q & 15)) return false;
    return !p.scale || (p.mod_stride % n) == 0;
}

// parameter / modulation gradients: thread per column, a workgroup walks a slab of rows of ONE batch item.
//   part[0] : d_gamma  = sum dy*(1+scale) * xhat
//   part[1] : d_scale  = sum dy * ln   (ln = xhat*gamma + beta)        (adaLN only)
//   part[2] : d_shift  = sum dy                                       (adaLN only)
#define SAT_LN_ROWS_PER_BLOCK 64
template <typename T>
__global__ void __launch_bounds__(256) sat_layernorm_bwd_param_kernel(SatLnParams p) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.z;
    const int r0 = b * p.rows_per_batch + blockIdx.y * SAT_LN_ROWS_PER_BLOCK;
    int r1 = r0 + SAT_LN_ROWS_PER_BLOCK;
    const int rend = (b + 1) * p.rows_per_batch;
    if (r1 > rend) r1 = rend;
    if (col >= p.D) return;
    const float gam = p.gamma[col], bet = p.beta ? p.beta[col] : 0.0f;
    const float sc = p.scale ? 1.0f + SatIO<T>::ld(p.scale, (long long)b * p.mod_stride + col) : 1.0f;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int r = r0; r < r1; ++r) {
        const float dy = SatIO<T>::ld(p.dy, (long long)r * p.D + col);
        const float xh = (SatIO<T>::ld(p.x, (long long)r * p.D + col) - p.mean[r]) * p.rstd[r];
        a0 += dy * sc * xh;
        a1 += dy * (xh * gam + bet);
        a2 += dy;
    }
    const int nby = gridDim.y * gridDim.z;
    const int by = b * gridDim.y + blockIdx.y;
    p.part[((size_t)0 * nby + by) * p.D + col] = a0;
    p.part[((size_t)1 * nby + by) * p.D + col] = a1;
    p.part[((size_t)2 * nby + by) * p.D + col] = a2;
}

// Two adjacent columns per thread (one 4-byte load of a bf16 pair / one 8-byte load of an fp32 pair), four rows in flight per step — the
// one-column kernel above is a chain of 64 dependent 2-byte loads per thread (22.7 us per launch at 4100 x 1536 bf16: 1.1 TB/s, round 6
// profiles/r06_dit_train_b4_kernel_stats.csv).  Without adaLN modulation only part[0] is produced (the caller reads nothing else).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T> struct SatLd2;
template <> struct SatLd2<float> {
    static SAT_DEVICE void ld(const void* p, long long i, float* o) { const f32x2 v = *reinterpret_cast<const f32x2*>((const float*)p + i); o[0] = v[0]; o[1] = v[1]; }
};
template <> struct SatLd2<short> {
    static SAT_DEVICE void ld(const void* p, long long i, float* o) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>((const short*)p + i);
        o[0] = __builtin_bit_cast(float, v << 16);
        o[1] = __builtin_bit_cast(float, v & 0xffff0000u);
    }
};
template <typename T>
__global__ void __launch_bounds__(256) sat_layernorm_bwd_param2_kernel(SatLnParams p) {
    const int col = (blockIdx.x * 256 + threadIdx.x) * 2;
    const int b = blockIdx.z;
    const int r0 = b * p.rows_per_batch + blockIdx.y * SAT_LN_ROWS_PER_BLOCK;
    int r1 = r0 + SAT_LN_ROWS_PER_BLOCK;
    const int rend = (b + 1) * p.rows_per_batch;
    if (r1 > rend) r1 = rend;
    if (col >= p.D) return;                        // D is even: col + 1 < D
    const bool mod = p.scale != nullptr;
    float gam[2], bet[2], sc[2], a0[2] = {0.f, 0.f}, a1[2] = {0.f, 0.f}, a2[2] = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        gam[e] = p.gamma[col + e];
        bet[e] = p.beta ? p.beta[col + e] : 0.0f;
        sc[e] = mod ? 1.0f + SatIO<T>::ld(p.scale, (long long)b * p.mod_stride + col + e) : 1.0f;
    }
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
        float dy[4][2], xv[4][2], mu[4], rs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            SatLd2<T>::ld(p.dy, (long long)(r + u) * p.D + col, dy[u]);
            SatLd2<T>::ld(p.x, (long long)(r + u) * p.D + col, xv[u]);
            mu[u] = p.mean[r + u];
            rs[u] = p.rstd[r + u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float xh = (xv[u][e] - mu[u]) * rs[u];
                a0[e] += dy[u][e] * sc[e] * xh;
                if (mod) {
                    a1[e] += dy[u][e] * (xh * gam[e] + bet[e]);
                    a2[e] += dy[u][e];
                }
            }
    }
    for (; r < r1; ++r) {
        float dy[2], xv[2];
        SatLd2<T>::ld(p.dy, (long long)r * p.D + col, dy);
        SatLd2<T>::ld(p.x, (long long)r * p.D + col, xv);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float xh = (xv[e] - p.mean[r]) * p.rstd[r];
            a0[e] += dy[e] * sc[e] * xh;
            if (mod) {
                a1[e] += dy[e] * (xh * gam[e] + bet[e]);
                a2[e] += dy[e];
            }
        }
    }
    const int nby = gridDim.y * gridDim.z;
    const int by = b * gridDim.y + blockIdx.y;
    *reinterpret_cast<f32x2*>(p.part + ((size_t)0 * nby + by) * p.D + col) = f32x2{a0[0], a0[1]};
    if (mod) {
        *reinterpret_cast<f32x2*>(p.part + ((size_t)1 * nby + by) * p.D + col) = f32x2{a1[0], a1[1]};
        *reinterpret_cast<f32x2*>(p.part + ((size_t)2 * nby + by) * p.D + col) = f32x2{a2[0], a2[1]};
    }
}

static int sat_ln_check(int rows, int D, int rows_per_batch, int dtype, const char* who) {
    if (rows <= 0 || D <= 0 || rows_per_batch <= 0 || rows % rows_per_batch != 0) { sat_set_error(who); return 1; }
    if (dtype != 0 && dtype != 1) { sat_set_error("layernorm: dtype must be 0 (f32) or 1 (bf16)"); return 1; }
    return 0;
}

extern "C" int sat_layernorm_fwd(const void* x, const float* gamma, const float* beta, const void* scale,
                                 const void* shift, long long mod_stride, void* y, float* mean, float* rstd, int rows,
                                 int D, int rows_per_batch, float eps, int dtype, void* stream) {
    if (sat_ln_check(rows, D, rows_per_batch, dtype, "sat_layernorm_fwd: bad shape")) return 1;
    if ((scale == nullptr) != (shift == nullptr)) { sat_set_error("sat_layernorm_fwd: scale/shift must both be given"); return 1; }
    SatLnParams p{};
    p.x = x; p.gamma = gamma; p.beta = beta; p.scale = scale; p.shift = shift; p.y = y; p.mean = mean; p.rstd = rstd;
    p.mod_stride = mod_stride; p.rows = rows; p.D = D; p.rows_per_batch = rows_per_batch; p.eps = eps;
    dim3 grid(sat_cdiv(rows, 4));
    if (sat_ln_vec_ok(p, dtype == 0 ? 4 : 2)) {
        if (dtype == 0) SAT_LAUNCH(sat_layernorm_fwd_vec_kernel<float>, grid, dim3(256), stream, p);
        else SAT_LAUNCH(sat_layernorm_fwd_vec_kernel<short>, grid, dim3(256), stream, p);
    } else if (dtype == 0) SAT_LAUNCH(sat_layernorm_fwd_kernel<float>, grid, dim3(256), stream, p);
    else SAT_LAUNCH(sat_layernorm_fwd_kernel<short>, grid, dim3(256), stream, p);
    return sat_check_launch("sat_layernorm_fwd");
}

// LayerNorm (+ adaLN modulate) with the output quantised per row to fp8 e4m3: q (rows, D) bytes, qscale (rows) = row max / 448.  Needs
// the vector layout of the fast path (D a multiple of 64 lanes x 16 bytes, 16-byte aligned pointers): returns 2 (nothing launched, no error
// set) when the shape does not qualify — the caller takes sat_layernorm_fwd + sat_quant_fp8_rows.
extern "C" int sat_layernorm_fwd_fp8(const void* x, const float* gamma, const float* beta, const void* scale, const void* shift,
                                     long long mod_stride, void* q, float* qscale, int rows, int D, int rows_per_batch, float eps,
                                     int dtype, void* stream) {
    if (sat_ln_check(rows, D, rows_per_batch, dtype, "sat_layernorm_fwd_fp8: bad shape")) return 1;
    if ((scale == nullptr) != (shift == nullptr) || !q || !qscale) { sat_set_error("sat_layernorm_fwd_fp8: bad arguments"); return 1; }
    SatLnParams p{};
    p.x = x; p.gamma = gamma; p.beta = beta; p.scale = scale; p.shift = shift;
    p.mod_stride = mod_stride; p.rows = rows; p.D = D; p.rows_per_batch = rows_per_batch; p.eps = eps;
    p.q = (uint8_t*)q; p.qscale = qscale;
    if (!sat_ln_vec_ok(p, dtype == 0 ? 4 : 2) || ((uintptr_t)q & 7)) return 2;
    dim3 grid(sat_cdiv(rows, 4));
    if (dtype == 0) SAT_LAUNCH(sat_layernorm_fwd_fp8_kernel<float>, grid, dim3(256), stream, p);
    else SAT_LAUNCH(sat_layernorm_fwd_fp8_kernel<short>, grid, dim3(256), stream, p);
    return sat_check_launch("sat_layernorm_fwd_fp8");
}

extern "C" int sat_layernorm_bwd_nblocks(int rows, int rows_per_batch) {
    return (rows / rows_per_batch) * sat_cdiv(rows_per_batch, SAT_LN_ROWS_PER_BLOCK);
}

// dx = LayerNorm backward [+ dres]: dres (rows, D, activation dtype) is the gradient that reached x along the residual connection around the
// normalised branch (x = x + f(LN(x)), transformer.py:703-712) — added in the same pass instead of by an elementwise launch of autograd's.
static int sat_layernorm_bwd_impl(const void* dy, const void* x, const float* gamma, const float* beta, const void* scale, long long mod_stride,
                                  const float* mean, const float* rstd, const void* dres, void* dx, float* part, int rows, int D, int rows_per_batch,
                                  int dtype, void* stream);
extern "C" int sat_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta,
                                 const void* scale, long long mod_stride, const float* mean, const float* rstd,
                                 void* dx, float* part, int rows, int D, int rows_per_batch, int dtype, void* stream) {
    return sat_layernorm_bwd_impl(dy, x, gamma, beta, scale, mod_stride, mean, rstd, nullptr, dx, part, rows, D, rows_per_batch, dtype, stream);
}
extern "C" int sat_layernorm_bwd_res(const void* dy, const void* x, const float* gamma, const float* beta, const void* scale, long long mod_stride,
                                     const float* mean, const float* rstd, const void* dres, void* dx, float* part, int rows, int D,
                                     int rows_per_batch, int dtype, void* stream) {
    return sat_layernorm_bwd_impl(dy, x, gamma, beta, scale, mod_stride, mean, rstd, dres, dx, part, rows, D, rows_per_batch, dtype, stream);
}
static int sat_layernorm_bwd_impl(const void* dy, const void* x, const float* gamma, const float* beta, const void* scale, long long mod_stride,
                                  const float* mean, const float* rstd, const void* dres, void* dx, float* part, int rows, int D, int rows_per_batch,
                                  int dtype, void* stream) {
    if (sat_ln_check(rows, D, rows_per_batch, dtype, "sat_layernorm_bwd: bad shape")) return 1;
    SatLnParams p{};
    p.x = x; p.gamma = gamma; p.beta = beta; p.scale = scale; p.dy = dy; p.dx = dx; p.part = part; p.dres = dres;
    p.mean = const_cast<float*>(mean); p.rstd = const_cast<float*>(rstd);
    p.mod_stride = mod_stride; p.rows = rows; p.D = D; p.rows_per_batch = rows_per_batch;
    dim3 g1(sat_cdiv(rows, 4));
    dim3 g2(sat_cdiv(D, 256), sat_cdiv(rows_per_batch, SAT_LN_ROWS_PER_BLOCK), rows / rows_per_batch);
    const bool vec = sat_ln_vec_ok(p, dtype == 0 ? 4 : 2);
    // two columns per thread: D even and 8-byte aligned rows / partial planes
    const bool pair = (D % 2 == 0) && (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)part) % 8 == 0);
    dim3 g2p(sat_cdiv(D, 512), g2.y, g2.z);
    if (dtype == 0) {
        if (vec) SAT_LAUNCH(sat_layernorm_bwd_dx_vec_kernel<float>, g1, dim3(256), stream, p);
        else SAT_LAUNCH(sat_layernorm_bwd_dx_kernel<float>, g1, dim3(256), stream, p);
        if (pair) SAT_LAUNCH(sat_layernorm_bwd_param2_kernel<float>, g2p, dim3(256), stream, p);
        else SAT_LAUNCH(sat_layernorm_bwd_param_kernel<float>, g2, dim3(256), stream, p);
    } else {
        if (vec) SAT_LAUNCH(sat_layernorm_bwd_dx_vec_kernel<short>, g1, dim3(256), stream, p);
        else SAT_LAUNCH(sat_layernorm_bwd_dx_kernel<short>, g1, dim3(256), stream, p);
        if (pair) SAT_LAUNCH(sat_layernorm_bwd_param2_kernel<short>, g2p, dim3(256), stream, p);
        else SAT_LAUNCH(sat_layernorm_bwd_param_kernel<short>, g2, dim3(256), stream, p);
    }
    return sat_check_launch("sat_layernorm_bwd");
}

// ------------------------------------------------------------------------------------------------
// Rotary position embedding
// ------------------------------------------------------------------------------------------------
struct SatRopeTabParams {
    const float* inv_freq;  // (R/2)
    float* cs;              // (N, R/2, 2) = cos, sin
    int N, half;
    float pos_scale;        // positions are pos_scale * n (the q/k length ratio of transformer.py:498-503)
};
__global__ void __launch_bounds__(256) sat_rope_tables_kernel(SatRopeTabParams p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.N * p.half) return;
    const int n = i / p.half, j = i - n * p.half;
    // freqs = t * inv_freq in fp32 (einsum), then optionally * ratio, then cos/sin in fp32
    const float ang = ((float)n * p.inv_freq[j]) * p.pos_scale;
    p.cs[2 * i + 0] = cosf(ang);
    p.cs[2 * i + 1] = sinf(ang);
}
extern "C" int sat_rope_tables(const float* inv_freq, float* cs, int N, int half, float pos_scale, void* stream) {
    if (N <= 0 || half <= 0) { sat_set_error("sat_rope_tables: empty shape"); return 1; }
    SatRopeTabParams p{inv_freq, cs, N, half, pos_scale};
    SAT_LAUNCH(sat_rope_tables_kernel, dim3(sat_cdiv(N * half, 256)), dim3(256), stream, p);
    return sat_check_launch("sat_rope_tables");
}

struct SatRopeParams {
    void* t;          // element (b, n, h, d) at b*sb + n*sn + h*sh + d
    const float* cs;  // (Ntab, half, 2)
    long long sb, sn, sh;
    int B, N, H, half, tab_off;  // table row = tab_off + n  (freqs[-seq_len:], transformer.py:162)
    float sign;                  // +1 forward, -1 transpose (backward)
};
template <typename T>
__global__ void __launch_bounds__(256) sat_rope_apply_kernel(SatRopeParams p) {
    const long long total = (long long)p.B * p.N * p.H * p.half;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int j = (int)(i % p.half);
        long long q = i / p.half;
        const int h = (int)(q % p.H);
        q /= p.H;
        const int n = (int)(q % p.N), b = (int)(q / p.N);
        const long long base = (long long)b * p.sb + (long long)n * p.sn + (long long)h * p.sh;
        const float c = p.cs[2 * ((long long)(p.tab_off + n) * p.half + j) + 0];
        const float s = p.sign * p.cs[2 * ((long long)(p.tab_off + n) * p.half + j) + 1];
        const float x1 = SatIO<T>::ld(p.t, base + j), x2 = SatIO<T>::ld(p.t, base + p.half + j);
        // t*cos + rotate_half(t)*sin with rotate_half = (-x2, x1)   (transformer.py:149-152, :170)
        SatIO<T>::st(p.t, base + j, x1 * c - x2 * s);
        SatIO<T>::st(p.t, base + p.half + j, x2 * c + x1 * s);
    }
}
extern "C" int sat_rope_apply(void* t, const float* cs, long long sb, long long sn, long long sh, int B, int N, int H,
                              int half, int tab_off, int transpose, int dtype, void* stream) {
    if (B <= 0 || N <= 0 || H <= 0 || half <= 0 || tab_off < 0) { sat_set_error("sat_rope_apply: bad shape"); return 1; }
    if (dtype != 0 && dtype != 1) { sat_set_error("sat_rope_apply: dtype must be 0 (f32) or 1 (bf16)"); return 1; }
    SatRopeParams p{t, cs, sb, sn, sh, B, N, H, half, tab_off, transpose ? -1.0f : 1.0f};
    long long nb = sat_cdivll((long long)B * N * H * half, 256);
    if (nb > 8192) nb = 8192;
    if (dtype == 0) SAT_LAUNCH(sat_rope_apply_kernel<float>, dim3((unsigned)nb), dim3(256), stream, p);
    else SAT_LAUNCH(sat_rope_apply_kernel<short>, dim3((unsigned)nb), dim3(256), stream, p);
    return sat_check_launch("sat_rope_apply");
}

// ------------------------------------------------------------------------------------------------
// SwiGLU and gate/residual
// ------------------------------------------------------------------------------------------------
struct SatGluParams {
    const void* xin;   // (rows, 2F): [x | gate]
    void* out;         // fwd: (rows, F) ;  bwd: d_xin (rows, 2F)
    const void* dout;  // bwd: (rows, F)
    long long rows;
    int F;
};
SAT_DEVICE float sat_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

template <typename T>
__global__ void __launch_bounds__(256) sat_swiglu_fwd_kernel(SatGluParams p) {
    const long long total = p.rows * p.F;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / p.F;
        const int c = (int)(i - r * p.F);
        const float x = SatIO<T>::ld(p.xin, r * 2 * p.F + c), g = SatIO<T>::ld(p.xin, r * 2 * p.F + p.F + c);
        SatIO<T>::st(p.out, i, x * (g * sat_sigmoid(g)));
    }
}
template <typename T>
__global__ void __launch_bounds__(256) sat_swiglu_bwd_kernel(SatGluParams p) {
    const long long total = p.rows * p.F;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / p.F;
        const int c = (int)(i - r * p.F);
        const float x = SatIO<T>::ld(p.xin, r * 2 * p.F + c), g = SatIO<T>::ld(p.xin, r * 2 * p.F + p.F + c);
        const float d = SatIO<T>::ld(p.dout, i);
        const float sg = sat_sigmoid(g);
        SatIO<T>::st(p.out, r * 2 * p.F + c, d * g * sg);
        SatIO<T>::st(p.out, r * 2 * p.F + p.F + c, d * x * (sg * (1.0f + g * (1.0f - sg))));
    }
}
extern "C" int sat_swiglu(const void* xin, const void* dout, void* out, long long rows, int F, int backward, int dtype,
                          void* stream) {
    if (rows <= 0 || F <= 0) { sat_set_error("sat_swiglu: empty shape"); return 1; }
    if (dtype != 0 && dtype != 1) { sat_set_error("sat_swiglu: dtype must be 0 (f32) or 1 (bf16)"); return 1; }
    SatGluParams p{xin, out, dout, rows, F};
    long long nb = sat_cdivll(rows * F, 256);
    if (nb > 8192) nb = 8192;
    dim3 grid((unsigned)nb);
    if (!backward) {
        if (dtype == 0) SAT_LAUNCH(sat_swiglu_fwd_kernel<float>, grid, dim3(256), stream, p);
        else SAT_LAUNCH(sat_swiglu_fwd_kernel<short>, grid, dim3(256), stream, p);
    } else {
        if (dtype == 0) SAT_LAUNCH(sat_swiglu_bwd_kernel<float>, grid, dim3(256), stream, p);
        else SAT_LAUNCH(sat_swiglu_bwd_kernel<short>, grid, dim3(256), stream, p);
    }
    return sat_check_launch("sat_swiglu");
}

struct SatGateParams {
    const void* x;     // (B, N, D) branch output
    const void* gate;  // (B, D) with batch stride gstride
    const void* res;   // (B, N, D)
    void* y;
    long long gstride;
    int B, N, D;
};
template <typename T>
__global__ void __launch_bounds__(256) sat_gate_residual_kernel(SatGateParams p) {
    const long long total = (long long)p.B * p.N * p.D;
    const long long per_b = (long long)p.N * p.D;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / per_b);
        const int c = (int)(i % p.D);
        const float g = SatIO<T>::ld(p.gate, (long long)b * p.gstride + c);
        SatIO<T>::st(p.y, i, SatIO<T>::ld(p.x, i) * sat_sigmoid(1.0f - g) + SatIO<T>::ld(p.res, i));
    }
}
// backward of y = x * sigmoid(1 - gate[b]) + res:  dx = dy * s ; d_res = dy ; d_gate[b][c] = -sum_n dy * x * s * (1 - s)
// thread per column, a workgroup walks SAT_LN_ROWS_PER_BLOCK rows of one batch item; part[b][chunk][D]
struct SatGateBwdParams {
    const void* dy;
    const void* x;
    const void* gate;
    void* dx;
    float* part;
    long long gstride;
    int B, N, D;
};
template <typename T>
__global__ void __launch_bounds__(256) sat_gate_residual_bwd_kernel(SatGateBwdParams p) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * SAT_LN_ROWS_PER_BLOCK;
    int r1 = r0 + SAT_LN_ROWS_PER_BLOCK;
    if (r1 > p.N) r1 = p.N;
    if (col >= p.D) return;
    const float s = sat_sigmoid(1.0f - SatIO<T>::ld(p.gate, (long long)b * p.gstride + col));
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) {
        const long long i = ((long long)b * p.N + r) * p.D + col;
        const float dy = SatIO<T>::ld(p.dy, i);
        SatIO<T>::st(p.dx, i, dy * s);
        acc += dy * SatIO<T>::ld(p.x, i);
    }
    p.part[((size_t)b * gridDim.y + blockIdx.y) * p.D + col] = -acc * s * (1.0f - s);
}
extern "C" int sat_gate_residual_bwd_nchunks(int N) { return sat_cdiv(N, SAT_LN_ROWS_PER_BLOCK); }
extern "C" int sat_gate_residual_bwd(const void* dy, const void* x, const void* gate, long long gstride, void* dx,
                                     float* part, int B, int N, int D, int dtype, void* stream) {
    if (B <= 0 || N <= 0 || D <= 0) { sat_set_error("sat_gate_residual_bwd: empty shape"); return 1; }
    if (dtype != 0 && dtype != 1) { sat_set_error("sat_gate_residual_bwd: dtype must be 0 (f32) or 1 (bf16)"); return 1; }
    SatGateBwdParams p{dy, x, gate, dx, part, gstride, B, N, D};
    dim3 grid(sat_cdiv(D, 256), sat_cdiv(N, SAT_LN_ROWS_PER_BLOCK), B);
    if (dtype == 0) SAT_LAUNCH(sat_gate_residual_bwd_kernel<float>, grid, dim3(256), stream, p);
    else SAT_LAUNCH(sat_gate_residual_bwd_kernel<short>, grid, dim3(256), stream, p);
    return sat_check_launch("sat_gate_residual_bwd");
}

extern "C" int sat_gate_residual(const void* x, const void* gate, long long gstride, const void* res, void* y, int B,
                                 int N, int D, int dtype, void* stream) {
    if (B <= 0 || N <= 0 || D <= 0) { sat_set_error("sat_gate_residual: empty shape"); return 1; }
    if (dtype != 0 && dtype != 1) { sat_set_error("sat_gate_residual: dtype must be 0 (f32) or 1 (bf16)"); return 1; }
    SatGateParams p{x, gate, res, y, gstride, B, N, D};
    long long nb = sat_cdivll((long long)B * N * D, 256);
    if (nb > 8192) nb = 8192;
    if (dtype == 0) SAT_LAUNCH(sat_gate_residual_kernel<float>, dim3((unsigned)nb), dim3(256), stream, p);
    else SAT_LAUNCH(sat_gate_residual_kernel<short>, dim3((unsigned)nb), dim3(256), stream, p);
    return sat_check_launch("sat_gate_residual");
}

// ------------------------------------------------------------------------------------------------
// Classifier-free-guidance combine + CFG rescale + sampler update in one pass
// (reference: models/dit.py:400-410 — cond/uncond chunk, uncond + (cond - uncond) * scale, channel-std rescale with
//  scale_phi; inference/sampling.py — the per-step updates of EVERY sampler there are linear in (x, v, one more tensor, the
//  unconditioned output): v-DDIM :254-307 (eta > 0: + fresh noise; cfg_pp: eps from the unconditioned output), rectified-flow
//  Euler :98-135, RK4 stages :138-177 (third operand = the running k-sum), DPM-Solver++ :179-219 (third operand = the previous
//  step's denoised), ping-pong :222-250 (third operand = fresh noise)).
//   out2: (ncond * B, C, T) model output, conditioned half first (ncond = 2), or the plain output (ncond = 1)
//   v    = ncond == 2 ? rescale(uncond + (cond - uncond) * scale) : out2;   u = the unconditioned half (ncond 2) or v (ncond 1)
//   y0   = c0x * x + c0v * v + c0p * p + c0u * u           (x == NULL: y0 = v)
//   y1   = c1x * x + c1v * v + c1p * p + c1u * u           (optional second output: DDIM's `pred`, DPM++'s `denoised`, ...)
//   coefficient order (host array or device buffer, fp32): c0x c0v c0p c0u c1x c1v c1p c1u;  p optional (its terms drop out)
// One thread per (b, t) column, two passes over the C channels (unbiased channel std, as torch.std(dim=1)).
// ------------------------------------------------------------------------------------------------
struct SatCfgParams {
    const void* out2;
    const void* x;
    const void* prev;
    void* y0;
    void* y1;
    int B, C, T, ncond;
    float scale, phi;
    float c[8];
    const float* coef;    // device coefficients overriding the by-value ones (HIP-graph replay: the launch is frozen)
};
template <typename T>
__global__ void __launch_bounds__(256) sat_cfg_step_kernel(SatCfgParams p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)p.B * p.T) return;
    if (p.coef) {
#pragma unroll
        for (int j = 0; j < 8; ++j) p.c[j] = p.coef[j];
    }
    const int b = (int)(i / p.T), t = (int)(i - (long long)b * p.T);
    const long long cbase = (long long)b * p.C * p.T + t;
    const long long ubase = cbase + (long long)p.B * p.C * p.T;
    float ratio = 1.0f;
    if (p.ncond == 2 && p.phi != 0.0f) {
        // Welford: the one-pass sum / sum-of-squares form loses ~1e-4 of the std when |mean| is comparable to it (torch.std is two-pass)
        float mc = 0.f, qc = 0.f, mg = 0.f, qg = 0.f;
        for (int c = 0; c < p.C; ++c) {
            const float cv = SatIO<T>::ld(p.out2, cbase + (long long)c * p.T), uv = SatIO<T>::ld(p.out2, ubase + (long long)c * p.T);
            const float g = uv + (cv - uv) * p.scale;
            const float rn = 1.0f / (float)(c + 1);
            const float dc = cv - mc, dg = g - mg;
            mc += dc * rn; mg += dg * rn;
            qc += dc * (cv - mc); qg += dg * (g - mg);
        }
        const float dn = (float)(p.C > 1 ? p.C - 1 : 1);
        const float var_c = fmaxf(qc, 0.f) / dn, var_g = fmaxf(qg, 0.f) / dn;
        ratio = p.phi * (sqrtf(var_c) / sqrtf(var_g)) + (1.0f - p.phi);
    }
    for (int c = 0; c < p.C; ++c) {
        const long long o = cbase + (long long)c * p.T;
        float v = SatIO<T>::ld(p.out2, o);
        float u = v;
        if (p.ncond == 2) {
            u = SatIO<T>::ld(p.out2, ubase + (long long)c * p.T);
            v = (u + (v - u) * p.scale) * ratio;
        }
        if (p.x) {
            const float xv = SatIO<T>::ld(p.x, o);
            float r0 = p.c[0] * xv + p.c[1] * v, r1 = p.c[4] * xv + p.c[5] * v;
            if (p.prev) {
                const float pv = SatIO<T>::ld(p.prev, o);
                r0 += p.c[2] * pv;
                r1 += p.c[6] * pv;
            }
            if (p.c[3] != 0.f) r0 += p.c[3] * u;
            if (p.c[7] != 0.f) r1 += p.c[7] * u;
            SatIO<T>::st(p.y0, o, r0);
            if (p.y1) SatIO<T>::st(p.y1, o, r1);
        } else {
            SatIO<T>::st(p.y0, o, v);
        }
    }
}
static int sat_cfg_launch(const char* who, const void* out2, const void* x, const void* prev, void* y0, void* y1, int B, int C, int T,
                          int ncond, float scale, float phi, const float* host_coef, const float* dev_coef, int dtype, void* stream) {
    char msg[96];
    if (B <= 0 || C <= 0 || T <= 0 || (ncond != 1 && ncond != 2)) { snprintf(msg, sizeof msg, "%s: bad shape", who); sat_set_error(msg); return 1; }
    if (dtype != 0 && dtype != 1) { snprintf(msg, sizeof msg, "%s: dtype must be 0 (f32) or 1 (bf16)", who); sat_set_error(msg); return 1; }
    if (!out2 || !y0 || ((y1 || prev) && !x)) { snprintf(msg, sizeof msg, "%s: missing buffer", who); sat_set_error(msg); return 1; }
    SatCfgParams p{out2, x, prev, y0, y1, B, C, T, ncond, scale, phi, {0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dev_coef};
    if (host_coef) for (int j = 0; j < 8; ++j) p.c[j] = host_coef[j];
    const dim3 grid((unsigned)sat_cdivll((long long)B * T, 256));
    if (dtype == 0) SAT_LAUNCH(sat_cfg_step_kernel<float>, grid, dim3(256), stream, p);
    else SAT_LAUNCH(sat_cfg_step_kernel<short>, grid, dim3(256), stream, p);
    return sat_check_launch(who);
}
extern "C" int sat_cfg_step(const void* out2, const void* x, void* y0, void* y1, int B, int C, int T, int ncond, float scale,
                            float phi, float c0x, float c0v, float c1x, float c1v, int dtype, void* stream) {
    const float c[8] = {c0x, c0v, 0.f, 0.f, c1x, c1v, 0.f, 0.f};
    return sat_cfg_launch("sat_cfg_step", out2, x, nullptr, y0, y1, B, C, T, ncond, scale, phi, c, nullptr, dtype, stream);
}
// The general form: third operand `prev` (may be NULL) and the unconditioned-output terms; coef[8] on the HOST.
extern "C" int sat_sampler_step(const void* out2, const void* x, const void* prev, void* y0, void* y1, int B, int C, int T, int ncond,
                                float scale, float phi, const float* coef, int dtype, void* stream) {
    if (!coef) { sat_set_error("sat_sampler_step: coef is NULL"); return 1; }
    return sat_cfg_launch("sat_sampler_step", out2, x, prev, y0, y1, B, C, T, ncond, scale, phi, coef, nullptr, dtype, stream);
}
// The same with the coefficients read from DEVICE memory (coef[8] fp32): a sampler step captured into a HIP graph is replayed
// with new coefficients by rewriting that buffer (sampling.GraphedDenoiser).
extern "C" int sat_sampler_step_dev(const void* out2, const void* x, const void* prev, void* y0, void* y1, int B, int C, int T, int ncond,
                                    float scale, float phi, const float* coef, int dtype, void* stream) {
    if (!coef || !x) { sat_set_error("sat_sampler_step_dev: missing buffer"); return 1; }
    return sat_cfg_launch("sat_sampler_step_dev", out2, x, prev, y0, y1, B, C, T, ncond, scale, phi, nullptr, coef, dtype, stream);
}
