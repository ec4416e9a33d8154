// EXPERIMENT (not part of libsat_amd.so): variants of the bf16 attention forward kernel of csrc/attention.hip, built as
// tools/exp/libattn_x.so and timed by tools/attn_x_bench.py on the same operand planes as the product kernel.
//   FLAGS bit 0 (A): scalar fp32 softmax math instead of v_pk_*_f32
//   FLAGS bit 1 (B): Q pre-scaled by scale*log2(e) (one bf16 rounding more on Q), running max folded into the QK^T MFMA's C operand
//                    (S arrives as s*c - m: no per-score fma), first tile peeled
//   FLAGS bit 2 (C): (needs B) no per-tile row max: the tile's row sum detects an outgrown running max; the rare case recomputes QK^T
//   FLAGS bit 3 (P): s_setprio 1 around the MFMA groups
#include "sat_device.h"
#include <stdlib.h>
#include <cstdio>

#define D 64
#define TK 64
#define ROW 72

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct P {
    const short* q; const short* k; const short* vt; void* o; float* lse;
    int B, H, Hkv, Nq, Nk, Nqp, Nkp; float scale;
};

SAT_DEVICE int kperm_of(int a) { return (a & 0x13) | ((a & 4) << 1) | ((a & 8) >> 1); }
SAT_DEVICE float halfmax(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}
SAT_DEVICE float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
SAT_DEVICE bf16x8 frag(short (*t)[ROW], int row, int koff) { return *reinterpret_cast<const bf16x8*>(&t[row][koff]); }
SAT_DEVICE bf16x8 pack8(const f32x16& acc, int u) {
    typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
    typedef float f8 __attribute__((ext_vector_type(8)));
    const f8 v = {acc[8 * u], acc[8 * u + 1], acc[8 * u + 2], acc[8 * u + 3], acc[8 * u + 4], acc[8 * u + 5], acc[8 * u + 6], acc[8 * u + 7]};
    return __builtin_bit_cast(bf16x8, __builtin_convertvector(v, bf8));
}
#define DEFER 4.0f
#define SUMLIM 512.0f          // 32 scores per lane, each <= 2^DEFER when the running max is in range

// ---- the product kernel's tile (reference arm of the experiment; FLAGS bit 0 switches the packed math off) ----
template <int FLAGS, int NKB, bool MASK>
SAT_DEVICE void tile_v0(short (*k_lds)[ROW], short (*v_lds)[ROW], const bf16x8 (&qf)[4], f32x16 (&oacc)[2], float& m_run, float& l_run,
                        float sl2, int l31, int hi, int kperm, int nvalid) {
    f32x16 sacc[NKB];
    if (FLAGS & 8) SAT_SETPRIO(1);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 4; ++s) sacc[kb] = sat_mfma_32x32x16_bf16(frag(k_lds, kb * 32 + kperm, 16 * s + 8 * hi), qf[s], sacc[kb]);
    }
    if (FLAGS & 8) SAT_SETPRIO(0);
    if (MASK) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + (r & 7) + 8 * hi + 16 * (r >> 3);
                if (key >= nvalid) sacc[kb][r] = -INFINITY;
            }
    }
    float tmax = sacc[0][0];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = (kb == 0 ? 1 : 0); r < 16; ++r) tmax = fmaxf(tmax, sacc[kb][r]);
    tmax = halfmax(tmax);
    if (sat_wave_any((tmax - m_run) * sl2 > DEFER)) {
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = ex2((m_run - m_new) * sl2);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
    }
    const float mb = m_run * sl2;
    if (FLAGS & 1) {
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = ex2(__builtin_fmaf(sacc[kb][2 * j], sl2, -mb)), b = ex2(__builtin_fmaf(sacc[kb][2 * j + 1], sl2, -mb));
                ps0 += a; ps1 += b;
                sacc[kb][2 * j] = a; sacc[kb][2 * j + 1] = b;
            }
        l_run += ps0 + ps1;
    } else {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 sl2v = {sl2, sl2}, mbv = {mb, mb};
        f32x2 ps = {0.0f, 0.0f};
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f32x2 t = {sacc[kb][2 * j], sacc[kb][2 * j + 1]};
                t = t * sl2v - mbv;
                t[0] = ex2(t[0]);
                t[1] = ex2(t[1]);
                ps += t;
                sacc[kb][2 * j] = t[0];
                sacc[kb][2 * j + 1] = t[1];
            }
        l_run += ps[0] + ps[1];
    }
    if (FLAGS & 8) SAT_SETPRIO(1);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bf16x8 pb = pack8(sacc[kb], u);
#pragma unroll
            for (int t = 0; t < 2; ++t) oacc[t] = sat_mfma_32x32x16_bf16(frag(v_lds, t * 32 + l31, kb * 32 + 16 * u + 8 * hi), pb, oacc[t]);
        }
    if (FLAGS & 8) SAT_SETPRIO(0);
}

// ---- variant B / C: scores arrive shifted (x = s*c - mb through the MFMA's C operand), mb = running max in the exp2 domain ----
// FIRST: mb is not known yet — C = 0, true row max taken.  Otherwise negm = -mb in all 16 registers.
template <int FLAGS, int NKB, bool MASK, bool FIRST>
SAT_DEVICE void tile_b(short (*k_lds)[ROW], short (*v_lds)[ROW], const bf16x8 (&qf)[4], f32x16 (&oacc)[2], f32x16& negm, float& mb, float& l_run,
                       int l31, int hi, int kperm, int nvalid) {
    f32x16 x[NKB];
    auto qk = [&]() {
        if (FLAGS & 8) SAT_SETPRIO(1);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (FIRST) {
#pragma unroll
                for (int r = 0; r < 16; ++r) x[kb][r] = 0.0f;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
                x[kb] = sat_mfma_32x32x16_bf16(frag(k_lds, kb * 32 + kperm, 16 * s + 8 * hi), qf[s], (!FIRST && s == 0) ? negm : x[kb]);
        }
        if (FLAGS & 8) SAT_SETPRIO(0);
    };
    auto rowmax = [&]() {
        float tmax = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + (r & 7) + 8 * hi + 16 * (r >> 3);
                if (!MASK || key < nvalid) tmax = fmaxf(tmax, x[kb][r]);
            }
        return halfmax(tmax);
    };
    // move the running max by d >= 0 (per lane): rescale O and l, shift this tile's scores, refresh the C block
    auto shift = [&](float d) {
        const float alpha = ex2(-d);
        mb += d;
        l_run *= alpha;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[kb][r] -= d;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -mb;
    };
    qk();
    if (FIRST) {
        const float tmax = rowmax();
        mb = tmax;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[kb][r] -= tmax;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -mb;
    } else if (!(FLAGS & 4)) {
        const float tmax = rowmax();
        if (sat_wave_any(tmax > DEFER)) shift(fmaxf(tmax, 0.0f));
    }
    float ps0 = 0.f, ps1 = 0.f;
    auto expsum = [&]() {
        ps0 = 0.f; ps1 = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a = ex2(x[kb][2 * j]), b = ex2(x[kb][2 * j + 1]);
                if (MASK) {
                    const int key = kb * 32 + ((2 * j) & 7) + 8 * hi + 16 * ((2 * j) >> 3);
                    if (key >= nvalid) a = 0.f;
                    if (key + 1 >= nvalid) b = 0.f;
                }
                ps0 += a; ps1 += b;
                x[kb][2 * j] = a; x[kb][2 * j + 1] = b;
            }
    };
    expsum();
    if ((FLAGS & 4) && !FIRST) {
        // the row sum of this lane's 32 scores bounds every one of them: above the limit (or inf) the running max is stale
        if (sat_wave_any(!(ps0 + ps1 <= SUMLIM))) {
            qk();                                    // rare: the scores were overwritten by their exponentials
            const float tmax = rowmax();
            shift(fmaxf(tmax, 0.0f));
            expsum();
        }
    }
    l_run += ps0 + ps1;
    if (FLAGS & 8) SAT_SETPRIO(1);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bf16x8 pb = pack8(x[kb], u);
#pragma unroll
            for (int t = 0; t < 2; ++t) oacc[t] = sat_mfma_32x32x16_bf16(frag(v_lds, t * 32 + l31, kb * 32 + 16 * u + 8 * hi), pb, oacc[t]);
        }
    if (FLAGS & 8) SAT_SETPRIO(0);
}

template <int FLAGS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) attn_fwd_x(P p) {
    __shared__ __attribute__((aligned(16))) short k_lds2[2][TK][ROW];
    __shared__ __attribute__((aligned(16))) short v_lds2[2][D][ROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int kperm = kperm_of(l31);
    const int b = blockIdx.z, h = blockIdx.y;
    const int hk = h / (p.H / p.Hkv);
    const int qrow = blockIdx.x * 128 + wave * 32 + l31;
    const bool q_in = qrow < p.Nqp, q_ok = qrow < p.Nq;
    const bool w_ok = blockIdx.x * 128 + wave * 32 < p.Nq;
    const size_t qplane = ((size_t)b * p.H + h) * (size_t)p.Nqp * D;
    const size_t kplane = ((size_t)b * p.Hkv + hk) * (size_t)p.Nkp * D;
    const float sl2 = p.scale * 1.4426950408889634f;
    constexpr bool VB = (FLAGS & 2) != 0;

    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (q_in) qf[s] = *reinterpret_cast<const bf16x8*>(p.q + qplane + (size_t)qrow * D + 16 * s + 8 * hi);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[s][e] = 0;
        }
        if (VB) {       // Q <- Q * scale * log2(e), rounded to bf16 once per launch
            u32x4 w = __builtin_bit_cast(u32x4, qf[s]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = __builtin_bit_cast(float, w[j] << 16) * sl2, hi2 = __builtin_bit_cast(float, w[j] & 0xffff0000u) * sl2;
                w[j] = sat_cvt2_pk(lo, hi2);
            }
            qf[s] = __builtin_bit_cast(bf16x8, w);
        }
    }
    f32x16 oacc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f, mb = 0.0f;
    f32x16 negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.0f;

    bf16x8 kreg[2], vreg[2];
    auto tile_load = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = threadIdx.x + j * 256, r = c >> 3, part = c & 7;
            kreg[j] = *reinterpret_cast<const bf16x8*>(p.k + kplane + (size_t)(k0 + r) * D + part * 8);
            vreg[j] = *reinterpret_cast<const bf16x8*>(p.vt + kplane + (size_t)r * p.Nkp + k0 + part * 8);
        }
    };
    auto tile_store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = threadIdx.x + j * 256, r = c >> 3, part = c & 7;
            *reinterpret_cast<bf16x8*>(&k_lds2[buf][r][part * 8]) = kreg[j];
            *reinterpret_cast<bf16x8*>(&v_lds2[buf][r][part * 8]) = vreg[j];
        }
    };
    tile_load(0);
    tile_store(0);
    if (TK < p.Nk) tile_load(TK);
    __syncthreads();
    int buf = 0, k0 = 0;
    if (VB && p.Nk >= TK) {     // peeled first tile (full): establishes the running max
        if (TK < p.Nk) {
            tile_store(1);
            if (2 * TK < p.Nk) tile_load(2 * TK);
        }
        if (w_ok) tile_b<FLAGS, 2, false, true>(k_lds2[0], v_lds2[0], qf, oacc, negm, mb, l_run, l31, hi, kperm, TK);
        __syncthreads();
        k0 = TK; buf = 1;
    }
    for (; k0 + TK <= p.Nk; k0 += TK, buf ^= 1) {
        if (k0 + TK < p.Nk) {
            tile_store(buf ^ 1);
            if (k0 + 2 * TK < p.Nk) tile_load(k0 + 2 * TK);
        }
        if (w_ok) {
            if (VB) tile_b<FLAGS, 2, false, false>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, TK);
            else tile_v0<FLAGS, 2, false>(k_lds2[buf], v_lds2[buf], qf, oacc, m_run, l_run, sl2, l31, hi, kperm, TK);
        }
        __syncthreads();
    }
    if (k0 < p.Nk && w_ok) {
        const int rem = p.Nk - k0;
        if (VB) {
            if (k0 == 0) {      // fewer than 64 keys in total: the ragged tile is also the first
                if (rem > 32) tile_b<FLAGS, 2, true, true>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
                else tile_b<FLAGS, 1, true, true>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
            } else if (rem > 32) tile_b<FLAGS, 2, true, false>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
            else tile_b<FLAGS, 1, true, false>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
        } else {
            if (rem > 32) tile_v0<FLAGS, 2, true>(k_lds2[buf], v_lds2[buf], qf, oacc, m_run, l_run, sl2, l31, hi, kperm, rem);
            else tile_v0<FLAGS, 1, true>(k_lds2[buf], v_lds2[buf], qf, oacc, m_run, l_run, sl2, l31, hi, kperm, rem);
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv_l = 1.0f / l_tot;
    if (q_ok) {
        const long long obase = ((long long)b * p.Nq + qrow) * ((long long)p.H * D) + (long long)h * D;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {oacc[t][4 * g] * inv_l, oacc[t][4 * g + 1] * inv_l, oacc[t][4 * g + 2] * inv_l, oacc[t][4 * g + 3] * inv_l};
                const long long idx = obase + t * 32 + 8 * g + 4 * hi;
                *(u32x2*)((short*)p.o + idx) = u32x2{sat_cvt2_pk(v[0], v[1]), sat_cvt2_pk(v[2], v[3])};
            }
        if (p.lse && hi == 0)
            p.lse[((long long)b * p.H + h) * p.Nq + qrow] = VB ? (mb + log2f(l_tot)) * 0.6931471805599453f : m_run * p.scale + logf(l_tot);
    }
}

extern "C" int satx_attention_fwd(int variant, const short* q, const short* k, const short* vt, void* o, float* lse, int B, int H, int Hkv,
                                  int Nq, int Nk, int Nqp, int Nkp, float scale, void* stream) {
    P p{q, k, vt, o, lse, B, H, Hkv, Nq, Nk, Nqp, Nkp, scale};
    dim3 grid((Nq + 127) / 128, H, B);
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
        case 0: hipLaunchKernelGGL(attn_fwd_x<0>, grid, dim3(256), 0, st, p); break;
        case 1: hipLaunchKernelGGL(attn_fwd_x<1>, grid, dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL(attn_fwd_x<3>, grid, dim3(256), 0, st, p); break;
        case 7: hipLaunchKernelGGL(attn_fwd_x<7>, grid, dim3(256), 0, st, p); break;
        case 8: hipLaunchKernelGGL(attn_fwd_x<8>, grid, dim3(256), 0, st, p); break;
        case 11: hipLaunchKernelGGL(attn_fwd_x<11>, grid, dim3(256), 0, st, p); break;
        case 15: hipLaunchKernelGGL(attn_fwd_x<15>, grid, dim3(256), 0, st, p); break;
        default: return 2;
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
