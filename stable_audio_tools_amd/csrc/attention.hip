// attention.hip — dense non-causal scaled-dot-product attention for the DiT (head dim 64), flash
// style on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16), self- and grouped-query cross-attention.
//
// Replaces Attention.apply_attn (stable_audio_tools/models/transformer.py:406-441): flash_attn_func /
// F.scaled_dot_product_attention with scale 1/sqrt(d), no mask (masks are disabled by the caller,
// models/dit.py:283), and the GQA repeat_interleave of k/v (transformer.py:408-411) done by indexing.
//
// Structure (one wave = 32 query rows, 4 waves per workgroup, 64-key tiles staged through LDS):
//   * "swapped" products: S^T = K Q^T and O^T = V^T P^T, so a lane owns ONE query column — its
//     softmax statistics are lane-local (one lane^32 exchange per tile for the max), the O rescale
//     is a per-lane scalar, and the P^T accumulator registers feed the second MFMA as its B operand
//     with no cross-lane traffic: the MFMA k-slot order is simply defined to be the accumulator's
//     row order, and the V^T fragment is read from LDS in that same order.
//   * Q fragments live in registers for the whole kernel; K tile row-major (padded rows ->
//     conflict-free ds_read_b128), V tile transposed in LDS.
//   * fp32 inputs (the 1e-3 parity mode) are split into bf16 hi + lo parts and every product is
//     three MFMAs (hi*hi + hi*lo + lo*hi): ~2^-16 relative error per product, fp32 accumulate.
//     bf16 inputs use one MFMA per product.
// Outputs O in (B, Nq, H*64) — heads already merged for the to_out projection — and the
// log-sum-exp per (b, h, q) for the backward pass.
#include "sat_device.h"

#define SAT_ATT_D 64
#define SAT_ATT_KT 64            // keys per tile
#define SAT_ATT_KROW 72          // bf16 per K row in LDS (64 + 8 pad -> 144 B stride)
#define SAT_ATT_VROW 72          // bf16 per V^T row in LDS (64 keys + 8 pad)

struct SatAttnParams {
    const void* q;   // element (b, h, n, d) at b*sqb + h*sqh + n*sqn + d   (element strides)
    const void* k;   // (b, hk, n, d)
    const void* v;
    void* o;         // (B, Nq, H*64)
    float* lse;      // (B, H, Nq) or null
    long long sqb, sqh, sqn, skb, skh, skn, svb, svh, svn;
    int B, H, Hkv, Nq, Nk;
    float scale;
};

template <typename T> struct SatLoad;
template <> struct SatLoad<float> {
    static SAT_DEVICE float at(const void* p, long long i) { return ((const float*)p)[i]; }
    static SAT_DEVICE void put(void* p, long long i, float v) { ((float*)p)[i] = v; }
};
template <> struct SatLoad<short> {  // bf16 bits
    static SAT_DEVICE float at(const void* p, long long i) { return sat_bf16_to_f32(((const short*)p)[i]); }
    static SAT_DEVICE void put(void* p, long long i, float v) { ((short*)p)[i] = sat_f32_to_bf16(v); }
};

SAT_DEVICE void sat_split_bf16(float x, short* hi, short* lo) {
    const short h = sat_f32_to_bf16(x);
    *hi = h;
    *lo = sat_f32_to_bf16(x - sat_bf16_to_f32(h));
}

template <typename T, bool SPLIT>
__global__ void __launch_bounds__(256) sat_attn_fwd_kernel(SatAttnParams p) {
    constexpr int NP = SPLIT ? 2 : 1;  // hi (+ lo) planes
    __shared__ __attribute__((aligned(16))) short k_lds[NP][SAT_ATT_KT][SAT_ATT_KROW];   // [key][d]
    __shared__ __attribute__((aligned(16))) short v_lds[NP][SAT_ATT_D][SAT_ATT_VROW];    // [d][key]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int hk = h / (p.H / p.Hkv);
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int qrow = q0 + l31;
    const bool q_ok = qrow < p.Nq;

    // ---- Q fragments: B operand of S^T = K Q^T : lane (q = l31, hi) holds d = 16*s + 8*hi + e ----
    bf16x8 qf[NP][4];
    {
        const long long base = (long long)b * p.sqb + (long long)h * p.sqh + (long long)qrow * p.sqn;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = q_ok ? SatLoad<T>::at(p.q, base + 16 * s + 8 * hi + e) : 0.0f;
                if (SPLIT) {
                    short a, c;
                    sat_split_bf16(x, &a, &c);
                    qf[0][s][e] = a;
                    qf[NP - 1][s][e] = c;
                } else {
                    qf[0][s][e] = sat_f32_to_bf16(x);
                }
            }
        }
    }

    f32x16 oacc[2];   // O^T: rows d (2 tiles of 32), cols q
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.0f;
    float m_run = -INFINITY;   // running max of raw scores (this lane's query)
    float l_run = 0.0f;        // running sum over THIS half's keys (combined with lane^32 at the end)
    const float sl2 = p.scale * 1.4426950408889634f;   // scale * log2(e)

    const long long kbase = (long long)b * p.skb + (long long)hk * p.skh;
    const long long vbase = (long long)b * p.svb + (long long)hk * p.svh;

    for (int k0 = 0; k0 < p.Nk; k0 += SAT_ATT_KT) {
        __syncthreads();   // previous tile fully consumed
        // ---- stage K [key][d] and V^T [d][key] (zero beyond Nk) ----
        for (int i = tid; i < SAT_ATT_KT * (SAT_ATT_D / 4); i += 256) {
            const int key = i >> 4, d4 = (i & 15) * 4;
            const bool ok = (k0 + key) < p.Nk;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float kx = ok ? SatLoad<T>::at(p.k, kbase + (long long)(k0 + key) * p.skn + d4 + e) : 0.0f;
                const float vx = ok ? SatLoad<T>::at(p.v, vbase + (long long)(k0 + key) * p.svn + d4 + e) : 0.0f;
                if (SPLIT) {
                    short a, c;
                    sat_split_bf16(kx, &a, &c);
                    k_lds[0][key][d4 + e] = a;
                    k_lds[NP - 1][key][d4 + e] = c;
                    sat_split_bf16(vx, &a, &c);
                    v_lds[0][d4 + e][key] = a;
                    v_lds[NP - 1][d4 + e][key] = c;
                } else {
                    k_lds[0][key][d4 + e] = sat_f32_to_bf16(kx);
                    v_lds[0][d4 + e][key] = sat_f32_to_bf16(vx);
                }
            }
        }
        __syncthreads();

        // ---- S^T = K Q^T for the two 32-key sub-tiles ----
        f32x16 sacc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                // A operand: K[key = kb*32 + l31][d = 16 s + 8 hi + e]
                const bf16x8 ka = *reinterpret_cast<const bf16x8*>(&k_lds[0][kb * 32 + l31][16 * s + 8 * hi]);
                sacc[kb] = sat_mfma_32x32x16_bf16(ka, qf[0][s], sacc[kb]);
                if (SPLIT) {
                    const bf16x8 kl = *reinterpret_cast<const bf16x8*>(&k_lds[NP - 1][kb * 32 + l31][16 * s + 8 * hi]);
                    sacc[kb] = sat_mfma_32x32x16_bf16(ka, qf[NP - 1][s], sacc[kb]);
                    sacc[kb] = sat_mfma_32x32x16_bf16(kl, qf[0][s], sacc[kb]);
                }
            }
        }
        // ---- online softmax for this lane's query over its 32 (of 64) keys ----
        float tmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (key >= p.Nk) sacc[kb][r] = -INFINITY;
                tmax = fmaxf(tmax, sacc[kb][r]);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);            // finite: every tile has >= 1 valid key
        const float alpha = exp2f((m_run - m_new) * sl2);  // first tile: exp2(-inf) = 0
        m_run = m_new;
        float psum = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = exp2f((sacc[kb][r] - m_new) * sl2);
                sacc[kb][r] = pv;
                psum += pv;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;

        // ---- O^T += V^T P^T.  k-slot (hi, e) of MFMA u in sub-tile kb  <->  accumulator register 8u+e,
        //      i.e. key kb*32 + 16u + 4hi + (e&3) + 8(e>>2) ----
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bf16x8 pb[NP];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = sacc[kb][8 * u + e];
                    if (SPLIT) {
                        short a, c;
                        sat_split_bf16(pv, &a, &c);
                        pb[0][e] = a;
                        pb[NP - 1][e] = c;
                    } else {
                        pb[0][e] = sat_f32_to_bf16(pv);
                    }
                }
                const int kofs = kb * 32 + 16 * u + 4 * hi;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const short* vr = &v_lds[0][t * 32 + l31][kofs];
                    bf16x8 va;
                    typedef short s4 __attribute__((ext_vector_type(4)));
                    const s4 lo4 = *reinterpret_cast<const s4*>(vr);
                    const s4 hi4 = *reinterpret_cast<const s4*>(vr + 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        va[e] = lo4[e];
                        va[4 + e] = hi4[e];
                    }
                    oacc[t] = sat_mfma_32x32x16_bf16(va, pb[0], oacc[t]);
                    if (SPLIT) {
                        const short* vl = &v_lds[NP - 1][t * 32 + l31][kofs];
                        bf16x8 vb;
                        const s4 lo4b = *reinterpret_cast<const s4*>(vl);
                        const s4 hi4b = *reinterpret_cast<const s4*>(vl + 8);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            vb[e] = lo4b[e];
                            vb[4 + e] = hi4b[e];
                        }
                        oacc[t] = sat_mfma_32x32x16_bf16(va, pb[NP - 1], oacc[t]);
                        oacc[t] = sat_mfma_32x32x16_bf16(vb, pb[0], oacc[t]);
                    }
                }
            }
        }
    }

    // ---- epilogue ----
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv_l = 1.0f / l_tot;
    if (q_ok) {
        const long long obase = ((long long)b * p.Nq + qrow) * ((long long)p.H * SAT_ATT_D) + (long long)h * SAT_ATT_D;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                SatLoad<T>::put(p.o, obase + d, oacc[t][r] * inv_l);
            }
        if (p.lse && hi == 0) p.lse[((long long)b * p.H + h) * p.Nq + qrow] = m_run * p.scale + logf(l_tot);
    }
}

extern "C" int sat_attention_fwd(const void* q, const void* k, const void* v, void* o, float* lse, long long sqb,
                                 long long sqh, long long sqn, long long skb, long long skh, long long skn,
                                 long long svb, long long svh, long long svn, int B, int H, int Hkv, int Nq, int Nk,
                                 int head_dim, float scale, int dtype, void* stream) {
    if (B <= 0 || H <= 0 || Hkv <= 0 || Nq <= 0 || Nk <= 0) { sat_set_error("sat_attention_fwd: empty shape"); return 1; }
    if (head_dim != SAT_ATT_D) { sat_set_error("sat_attention_fwd: only head_dim == 64 (the Stable Audio DiT) is implemented"); return 1; }
    if (H % Hkv != 0) { sat_set_error("sat_attention_fwd: H must be a multiple of Hkv"); return 1; }
    if (dtype != 0 && dtype != 1) { sat_set_error("sat_attention_fwd: dtype must be 0 (f32) or 1 (bf16)"); return 1; }
    SatAttnParams p{q, k, v, o, lse, sqb, sqh, sqn, skb, skh, skn, svb, svh, svn, B, H, Hkv, Nq, Nk, scale};
    dim3 grid(sat_cdiv(Nq, 128), H, B);
    if (dtype == 0) SAT_LAUNCH((sat_attn_fwd_kernel<float, true>), grid, dim3(256), stream, p);
    else SAT_LAUNCH((sat_attn_fwd_kernel<short, false>), grid, dim3(256), stream, p);
    return sat_check_launch("sat_attention_fwd");
}
