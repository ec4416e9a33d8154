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
#include <type_traits>

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

SAT_DEVICE bf16x8 fragk(const short* t, int row, int chunk) {      // [64][64] bf16, 16-byte slot s of row r holds chunk s ^ ((r >> 1) & 7)
    return *reinterpret_cast<const bf16x8*>(t + row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3));
}
template <int FLAGS, int NKB, bool MASK, bool FIRST>
SAT_DEVICE void tile_bk(const short* k_lds, short (*v_lds)[ROW], const bf16x8 (&qf)[4], f32x16 (&oacc)[2], f32x16& negm, float& mb, float& l_run,
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
                x[kb] = sat_mfma_32x32x16_bf16(fragk(k_lds, kb * 32 + kperm, 2 * s + hi), qf[s], (!FIRST && s == 0) ? negm : x[kb]);
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


// ---- variant PIPE (needs B + C): two-tile software pipeline inside the wave — iteration k issues QK^T of tile k+1, then the PV MFMAs
// of tile k INTERLEAVED with the exponentials / row sums of tile k+1 (sched_group_barrier pins one MFMA + one VALU slice per step);
// P(k) lives as 4 packed bf16x8 (16 registers), so no second score tile is alive.  K stream one tile ahead of the V stream; two K and
// two V buffers.  The ragged last tile runs un-pipelined after the drain.
template <int FLAGS>
SAT_DEVICE void attn_fwd_p_body(const P& p, short (*k_lds2)[TK][ROW], short (*v_lds2)[D][ROW]) {
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
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (q_in) qf[s] = *reinterpret_cast<const bf16x8*>(p.q + qplane + (size_t)qrow * D + 16 * s + 8 * hi);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[s][e] = 0;
        }
        u32x4 w = __builtin_bit_cast(u32x4, qf[s]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = __builtin_bit_cast(float, w[j] << 16) * sl2, hi2 = __builtin_bit_cast(float, w[j] & 0xffff0000u) * sl2;
            w[j] = sat_cvt2_pk(lo, hi2);
        }
        qf[s] = __builtin_bit_cast(bf16x8, w);
    }
    f32x16 oacc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.0f;
    float l_run = 0.0f, mb = 0.0f;
    f32x16 negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.0f;

    const int nfull = p.Nk / TK, rem = p.Nk - nfull * TK, ntiles = nfull + (rem > 0 ? 1 : 0);
    bf16x8 kreg[2], vreg[2];
    auto loadK = [&](int t) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = threadIdx.x + j * 256, r = c >> 3, part = c & 7;
            kreg[j] = *reinterpret_cast<const bf16x8*>(p.k + kplane + (size_t)(t * TK + r) * D + part * 8);
        }
    };
    auto loadV = [&](int t) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = threadIdx.x + j * 256, r = c >> 3, part = c & 7;
            vreg[j] = *reinterpret_cast<const bf16x8*>(p.vt + kplane + (size_t)r * p.Nkp + t * TK + part * 8);
        }
    };
    auto storeK = [&](int t) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = threadIdx.x + j * 256, r = c >> 3, part = c & 7;
            *reinterpret_cast<bf16x8*>(&k_lds2[t & 1][r][part * 8]) = kreg[j];
        }
    };
    auto storeV = [&](int t) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = threadIdx.x + j * 256, r = c >> 3, part = c & 7;
            *reinterpret_cast<bf16x8*>(&v_lds2[t & 1][r][part * 8]) = vreg[j];
        }
    };
    if (nfull == 0) {          // fewer than 64 keys: one ragged tile, un-pipelined
        loadK(0); loadV(0); storeK(0); storeV(0);
        __syncthreads();
        if (w_ok) {
            if (rem > 32) tile_b<FLAGS, 2, true, true>(k_lds2[0], v_lds2[0], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
            else tile_b<FLAGS, 1, true, true>(k_lds2[0], v_lds2[0], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
        }
    } else {
        f32x16 x[2];
        bf16x8 pb[4];
        float ps0 = 0.f, ps1 = 0.f;
        auto qk = [&](short (*k_lds)[ROW], bool first) {
            if (FLAGS & 8) SAT_SETPRIO(1);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                if (first) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) x[kb][r] = 0.0f;
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    x[kb] = sat_mfma_32x32x16_bf16(frag(k_lds, kb * 32 + kperm, 16 * s + 8 * hi), qf[s], (!first && s == 0) ? negm : x[kb]);
            }
            if (FLAGS & 8) SAT_SETPRIO(0);
        };
        auto rowmax = [&]() {
            float tmax = x[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = (kb == 0 ? 1 : 0); r < 16; ++r) tmax = fmaxf(tmax, x[kb][r]);
            return halfmax(tmax);
        };
        auto expsum_all = [&]() {
            ps0 = 0.f; ps1 = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = ex2(x[kb][2 * j]), c = ex2(x[kb][2 * j + 1]);
                    ps0 += a; ps1 += c;
                    x[kb][2 * j] = a; x[kb][2 * j + 1] = c;
                }
        };
        auto packall = [&]() {
#pragma unroll
            for (int i = 0; i < 4; ++i) pb[i] = pack8(x[i >> 1], i & 1);
        };
        // prologue: K0, V0 -> LDS; K1 -> registers; tile 0: true row max, exponentials, P(0)
        loadK(0); loadV(0); storeK(0); storeV(0);
        if (1 < ntiles) loadK(1);
        __syncthreads();
        if (w_ok) {
            qk(k_lds2[0], true);
            const float tmax = rowmax();
            mb = tmax;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) x[kb][r] -= tmax;
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[r] = -mb;
            expsum_all();
            l_run += ps0 + ps1;
            packall();
        }
        if (1 < ntiles) storeK(1);
        if (2 < ntiles) loadK(2);
        if (1 < ntiles) loadV(1);
        __syncthreads();
        for (int k = 0; k + 1 < nfull; ++k) {
            // LDS holds K(k+1), V(k); registers hold K(k+2), V(k+1)
            if (k + 2 < ntiles) storeK(k + 2);
            if (k + 1 < ntiles) storeV(k + 1);
            if (k + 3 < ntiles) loadK(k + 3);
            if (k + 2 < ntiles) loadV(k + 2);
            if (w_ok) {
                qk(k_lds2[(k + 1) & 1], false);
                short (*v_lds)[ROW] = v_lds2[k & 1];
                ps0 = 0.f; ps1 = 0.f;
                // PV(k): 8 MFMAs, each followed by one slice (4 scores) of tile k+1's exponentials and row sum
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int ks = i >> 1, t = i & 1;          // k-step (16 keys) and output d block
                    oacc[t] = sat_mfma_32x32x16_bf16(frag(v_lds, t * 32 + l31, 16 * ks + 8 * hi), pb[ks], oacc[t]);
                    const int kb = i >> 2, j0 = 2 * (i & 3);
#pragma unroll
                    for (int j = j0; j < j0 + 2; ++j) {
                        const float a = ex2(x[kb][2 * j]), c = ex2(x[kb][2 * j + 1]);
                        ps0 += a; ps1 += c;
                        x[kb][2 * j] = a; x[kb][2 * j + 1] = c;
                    }
                    if (FLAGS & 32) {
                        SAT_SCHED_GROUP(0x008, 1);      // one MFMA
                        SAT_SCHED_GROUP(0x002, 8);      // its VALU slice: 4 v_exp + 4 v_add
                    }
                }
                if (sat_wave_any(!(ps0 + ps1 <= SUMLIM))) {          // the running max went stale on tile k+1 (rare)
                    qk(k_lds2[(k + 1) & 1], false);
                    const float d = fmaxf(rowmax(), 0.0f), alpha = ex2(-d);
                    mb += d;
                    l_run *= alpha;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                        for (int r = 0; r < 16; ++r) x[kb][r] -= d;
#pragma unroll
                    for (int r = 0; r < 16; ++r) negm[r] = -mb;
                    expsum_all();
                }
                l_run += ps0 + ps1;
                packall();
            }
            __syncthreads();
        }
        // drain: V(nfull) (the ragged tile's) -> LDS, PV of the last full tile
        if (nfull < ntiles) storeV(nfull);
        if (w_ok) {
            short (*v_lds)[ROW] = v_lds2[(nfull - 1) & 1];
            if (FLAGS & 8) SAT_SETPRIO(1);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ks = i >> 1, t = i & 1;
                oacc[t] = sat_mfma_32x32x16_bf16(frag(v_lds, t * 32 + l31, 16 * ks + 8 * hi), pb[ks], oacc[t]);
            }
            if (FLAGS & 8) SAT_SETPRIO(0);
        }
        if (rem > 0) {
            __syncthreads();
            if (w_ok) {
                if (rem > 32) tile_b<FLAGS, 2, true, false>(k_lds2[nfull & 1], v_lds2[nfull & 1], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
                else tile_b<FLAGS, 1, true, false>(k_lds2[nfull & 1], v_lds2[nfull & 1], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
            }
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
        if (p.lse && hi == 0) p.lse[((long long)b * p.H + h) * p.Nq + qrow] = (mb + log2f(l_tot)) * 0.6931471805599453f;
    }
}


// ---- V2: issue-slot diet.  The SQ counters say the product kernel is bound by instruction ISSUE (the per-wave ACTIVE quad-cycles of
// the three waves of a SIMD add up to the SIMD's time): every instruction, whatever its pipe, costs >= one 4-cycle slot, v_exp_f32 two.
//   * K / V^T tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: 4 instructions per wave and tile instead of 4 global loads +
//     4 ds_write_b128 at ~13 cycles each, and 16 staging VGPRs freed); rows are 128 B unpadded, 16-byte slot s of row r holds chunk
//     s ^ ((r >> 1) & 7) (the swizzle rides on the DMA's per-lane SOURCE address), fragment reads stay conflict-free ds_read_b128
//   * FLAGS & 64: the row sums run on the MATRIX pipe — four MFMAs with an all-ones A operand per tile (l[q] = sum_k P[k][q] in every
//     accumulator row) instead of 32 v_add_f32; the sums then use the same bf16-rounded P as the numerator
//   * B + C + setprio as before (Q pre-scaled, running max in the QK^T C operand, stale-max detection from the row sum)
#define ROW2 64
SAT_DEVICE bf16x8 frag2(const short* t, int row, int chunk) {      // t: [64][64] bf16, swizzled slots
    return *reinterpret_cast<const bf16x8*>(t + row * ROW2 + ((chunk ^ ((row >> 1) & 7)) << 3));
}
template <int FLAGS, int NKB, bool MASK, bool FIRST>
SAT_DEVICE void tile_v2(const short* k_lds, const short* v_lds, const bf16x8 (&qf)[4], f32x16 (&oacc)[2], f32x16& negm, f32x16& lacc, float& mb,
                        float& l_run, int l31, int hi, int kperm, int nvalid) {
    constexpr bool LM = (FLAGS & 64) != 0;
    f32x16 x[NKB];
    bf16x8 pb[2 * NKB];
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    auto qk = [&]() {
        SAT_SETPRIO(1);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (FIRST) {
#pragma unroll
                for (int r = 0; r < 16; ++r) x[kb][r] = 0.0f;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
                x[kb] = sat_mfma_32x32x16_bf16(frag2(k_lds, kb * 32 + kperm, 2 * s + hi), qf[s], (!FIRST && s == 0) ? negm : x[kb]);
        }
        SAT_SETPRIO(0);
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
    float ps = 0.f;
    auto exps = [&]() {          // exponentials (+ VALU row sum when the matrix pipe does not take it), pack to bf16
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a = ex2(x[kb][2 * j]), c = ex2(x[kb][2 * j + 1]);
                if (MASK) {
                    const int key = kb * 32 + ((2 * j) & 7) + 8 * hi + 16 * ((2 * j) >> 3);
                    if (key >= nvalid) a = 0.f;
                    if (key + 1 >= nvalid) c = 0.f;
                }
                if (!LM) { ps0 += a; ps1 += c; }
                x[kb][2 * j] = a; x[kb][2 * j + 1] = c;
            }
        ps = ps0 + ps1;
#pragma unroll
        for (int i = 0; i < 2 * NKB; ++i) pb[i] = pack8(x[i >> 1], i & 1);
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
    }
    exps();
    float l_before = 0.f;
    auto lsum = [&]() {          // row sums of this tile on the matrix pipe: every row of the result = sum over the tile's keys
        l_before = lacc[0];
        SAT_SETPRIO(1);
#pragma unroll
        for (int i = 0; i < 2 * NKB; ++i) lacc = sat_mfma_32x32x16_bf16(ones, pb[i], lacc);
        SAT_SETPRIO(0);
    };
    if (LM) lsum();
    if (!FIRST) {
        const float grown = LM ? lacc[0] - l_before : ps;       // LM: the full row sum of 64 keys; else this lane's 32
        if (sat_wave_any(!(grown <= (LM ? 2.0f * SUMLIM : SUMLIM)))) {          // stale running max (rare): redo the tile with the true one
            if (LM) {
#pragma unroll
                for (int r = 0; r < 16; ++r) lacc[r] = l_before;
            }
            qk();
            const float d = fmaxf(rowmax(), 0.0f), alpha = ex2(-d);
            mb += d;
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) lacc[r] *= alpha;
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
            exps();
            if (LM) lsum();
        }
    }
    if (!LM) l_run += ps;
    SAT_SETPRIO(1);
#pragma unroll
    for (int i = 0; i < 2 * NKB; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t) oacc[t] = sat_mfma_32x32x16_bf16(frag2(v_lds, t * 32 + l31, 2 * i + hi), pb[i], oacc[t]);
    SAT_SETPRIO(0);
}

template <int FLAGS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) attn_fwd_v2(P p) {
    __shared__ __attribute__((aligned(1024))) short k_lds2[2][TK * ROW2];
    __shared__ __attribute__((aligned(1024))) short v_lds2[2][D * ROW2];
    constexpr bool LM = (FLAGS & 64) != 0;
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
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (q_in) qf[s] = *reinterpret_cast<const bf16x8*>(p.q + qplane + (size_t)qrow * D + 16 * s + 8 * hi);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[s][e] = 0;
        }
        u32x4 w = __builtin_bit_cast(u32x4, qf[s]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = __builtin_bit_cast(float, w[j] << 16) * sl2, hi2 = __builtin_bit_cast(float, w[j] & 0xffff0000u) * sl2;
            w[j] = sat_cvt2_pk(lo, hi2);
        }
        qf[s] = __builtin_bit_cast(bf16x8, w);
    }
    f32x16 oacc[2], negm, lacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.0f; oacc[1][r] = 0.0f; negm[r] = 0.0f; lacc[r] = 0.0f; }
    float l_run = 0.0f, mb = 0.0f;

    // DMA staging: piece pb = wave * 2 + j covers rows 8 pb .. 8 pb + 7; lane -> (row, slot); slot holds chunk slot ^ ((row >> 1) & 7)
    int srow[2], schunk[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        srow[j] = 8 * (wave * 2 + j) + (lane >> 3);
        schunk[j] = (lane & 7) ^ ((srow[j] >> 1) & 7);
    }
    auto stage = [&](int k0, int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            sat_glds16(p.k + kplane + (size_t)(k0 + srow[j]) * D + schunk[j] * 8, (char*)k_lds2[buf] + (wave * 2 + j) * 1024);
            sat_glds16(p.vt + kplane + (size_t)srow[j] * p.Nkp + k0 + schunk[j] * 8, (char*)v_lds2[buf] + (wave * 2 + j) * 1024);
        }
    };
    stage(0, 0);
    SAT_WAIT_VMCNT(0);
    __syncthreads();
    int buf = 0, k0 = 0;
    if (p.Nk >= TK) {
        if (TK < p.Nk) stage(TK, 1);
        if (w_ok) tile_v2<FLAGS, 2, false, true>(k_lds2[0], v_lds2[0], qf, oacc, negm, lacc, mb, l_run, l31, hi, kperm, TK);
        SAT_WAIT_VMCNT(0);
        __syncthreads();
        k0 = TK; buf = 1;
    }
    for (; k0 + TK <= p.Nk; k0 += TK, buf ^= 1) {
        if (k0 + TK < p.Nk) stage(k0 + TK, buf ^ 1);
        if (w_ok) tile_v2<FLAGS, 2, false, false>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, lacc, mb, l_run, l31, hi, kperm, TK);
        SAT_WAIT_VMCNT(0);
        __syncthreads();
    }
    if (k0 < p.Nk && w_ok) {
        const int rem = p.Nk - k0;
        if (k0 == 0) {
            if (rem > 32) tile_v2<FLAGS, 2, true, true>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, lacc, mb, l_run, l31, hi, kperm, rem);
            else tile_v2<FLAGS, 1, true, true>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, lacc, mb, l_run, l31, hi, kperm, rem);
        } else if (rem > 32) tile_v2<FLAGS, 2, true, false>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, lacc, mb, l_run, l31, hi, kperm, rem);
        else tile_v2<FLAGS, 1, true, false>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, lacc, mb, l_run, l31, hi, kperm, rem);
    }
    const float l_tot = LM ? lacc[0] : l_run + __shfl_xor(l_run, 32);
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
        if (p.lse && hi == 0) p.lse[((long long)b * p.H + h) * p.Nq + qrow] = (mb + log2f(l_tot)) * 0.6931471805599453f;
    }
}

template <int FLAGS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) attn_fwd_p(P p) {
    __shared__ __attribute__((aligned(16))) short k_lds2[2][TK][ROW];
    __shared__ __attribute__((aligned(16))) short v_lds2[2][D][ROW];
    attn_fwd_p_body<FLAGS>(p, k_lds2, v_lds2);
}
template <int FLAGS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) attn_fwd_p2(P p) {      // 256 registers: two waves per SIMD
    __shared__ __attribute__((aligned(16))) short k_lds2[2][TK][ROW];
    __shared__ __attribute__((aligned(16))) short v_lds2[2][D][ROW];
    attn_fwd_p_body<FLAGS>(p, k_lds2, v_lds2);
}


// ---- KS: key split inside the workgroup, for occupancy BALANCE at short sequences.  At N = 1025, B*H = 48 the 128-query workgroups are
// 432 four-wave units on 256 CUs: two on 176 CUs, one on 80 — the busiest SIMD walks 2 x 17 tiles, the average one 26.  Here a workgroup
// is 64 queries x 2 key halves (wave = (q-block, half)): 17 x 48 = 816 workgroups, three per CU, every SIMD three waves of ~8.5 tiles;
// the two halves of a q-block merge (O, l, m) through LDS at the end.  One K / V^T buffer per half (the same 36.9 KB per workgroup as the
// product kernel), the next tile in registers, two barriers per tile.
template <int FLAGS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) attn_fwd_ks(P p) {
    __shared__ __attribute__((aligned(1024))) short k_lds2[2][2][TK * 64];  // [half][buffer][key][d]: unpadded, swizzled slots, filled by LDS-DMA
    __shared__ __attribute__((aligned(16))) short v_lds2[2][D][ROW];        // [half][d][key]: one buffer, next tile in registers
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int kperm = kperm_of(l31);
    const int qb = wave & 1, half = wave >> 1;
    const int b = blockIdx.z, h = blockIdx.y;
    const int hk = h / (p.H / p.Hkv);
    const int qrow = blockIdx.x * 64 + qb * 32 + l31;
    const bool q_in = qrow < p.Nqp, q_ok = qrow < p.Nq;
    const bool w_ok = blockIdx.x * 64 + qb * 32 < p.Nq;
    const size_t qplane = ((size_t)b * p.H + h) * (size_t)p.Nqp * D;
    const size_t kplane = ((size_t)b * p.Hkv + hk) * (size_t)p.Nkp * D;
    const float sl2 = p.scale * 1.4426950408889634f;
    bf16x8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        if (q_in) qf[s] = *reinterpret_cast<const bf16x8*>(p.q + qplane + (size_t)qrow * D + 16 * s + 8 * hi);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[s][e] = 0;
        }
        u32x4 w = __builtin_bit_cast(u32x4, qf[s]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = __builtin_bit_cast(float, w[j] << 16) * sl2, hi2 = __builtin_bit_cast(float, w[j] & 0xffff0000u) * sl2;
            w[j] = sat_cvt2_pk(lo, hi2);
        }
        qf[s] = __builtin_bit_cast(bf16x8, w);
    }
    f32x16 oacc[2], negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc[0][r] = 0.0f; oacc[1][r] = 0.0f; negm[r] = 0.0f; }
    float l_run = 0.0f, mb = -INFINITY;

    const int ntiles = (p.Nk + TK - 1) / TK, rem = p.Nk - (ntiles - 1) * TK;      // rem in 1..64: keys of the last tile
    const int n0 = ntiles / 2;
    const int base = half ? n0 : 0, count = half ? ntiles - n0 : n0, steps = ntiles - n0;
    // staging: the 128 threads of a half load that half's tile: 512 K pieces + 512 V pieces of 16 bytes = 4 + 4 per thread
    const int th = threadIdx.x & 127;
    bf16x8 vreg[4];
    // K: 8 DMA pieces of 1 KiB per tile, 4 per wave of the half (piece = qb * 4 + j: rows 8 piece .. 8 piece + 7)
    auto k_dma = [&](int t, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = qb * 4 + j, r = piece * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
            sat_glds16(p.k + kplane + (size_t)(t * TK + r) * D + c * 8, (char*)k_lds2[half][buf] + piece * 1024);
        }
    };
    auto v_load = [&](int t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = th + j * 128, r = c >> 3, part = c & 7;
            vreg[j] = *reinterpret_cast<const bf16x8*>(p.vt + kplane + (size_t)r * p.Nkp + t * TK + part * 8);
        }
    };
    auto v_store = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = th + j * 128, r = c >> 3, part = c & 7;
            *reinterpret_cast<bf16x8*>(&v_lds2[half][r][part * 8]) = vreg[j];
        }
    };
    if (count > 0) { k_dma(base, 0); v_load(base); }
    // every step: [V registers -> LDS, K DMA landed] barrier [next tile: K DMA + V loads] compute barrier.  Step 0 (first tile: true max),
    // the plain middle steps and the last step (ragged tile) are separate code so that the hot loop holds ONE tile body.
    auto open_step = [&](int i) {
        if (i < count) v_store();                 // (waits for the V loads of tile i)
        SAT_WAIT_VMCNT(0);                        // this wave's K DMA pieces of tile i have landed
        __syncthreads();
        if (i + 1 < count) { k_dma(base + i + 1, (i + 1) & 1); v_load(base + i + 1); }
    };
    auto ragged_tile = [&](int i, auto first) {
        constexpr bool F = decltype(first)::value;
        if (rem > 32) tile_bk<FLAGS, 2, true, F>(k_lds2[half][i & 1], v_lds2[half], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
        else tile_bk<FLAGS, 1, true, F>(k_lds2[half][i & 1], v_lds2[half], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
    };
    {   // step 0
        open_step(0);
        if (count > 0 && w_ok) {
            if (base == ntiles - 1 && rem < TK) ragged_tile(0, std::true_type{});
            else tile_bk<FLAGS, 2, false, true>(k_lds2[half][0], v_lds2[half], qf, oacc, negm, mb, l_run, l31, hi, kperm, TK);
        }
        __syncthreads();
    }
    for (int i = 1; i + 1 < steps; ++i) {
        open_step(i);
        if (i < count && w_ok) tile_bk<FLAGS, 2, false, false>(k_lds2[half][i & 1], v_lds2[half], qf, oacc, negm, mb, l_run, l31, hi, kperm, TK);
        __syncthreads();
    }
    if (steps > 1) {   // last step
        const int i = steps - 1;
        open_step(i);
        if (i < count && w_ok) {
            if (base + i == ntiles - 1 && rem < TK) ragged_tile(i, std::false_type{});
            else tile_bk<FLAGS, 2, false, false>(k_lds2[half][i & 1], v_lds2[half], qf, oacc, negm, mb, l_run, l31, hi, kperm, TK);
        }
        __syncthreads();
    }
    // merge the two key halves of each q-block through LDS (the K buffers are free after the last barrier)
    float l_h = l_run + __shfl_xor(l_run, 32);
    float* mg = reinterpret_cast<float*>(&k_lds2[0][0][0]) + qb * 34 * 64;      // 17.4 KB of the 32-KB K area
    if (half == 1) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mg[(t * 16 + r) * 64 + lane] = oacc[t][r];
        mg[32 * 64 + lane] = l_h;
        mg[33 * 64 + lane] = mb;
    }
    __syncthreads();
    if (half == 0 && q_ok) {
        const float l1 = mg[32 * 64 + lane], m1 = mg[33 * 64 + lane];
        const float m = fmaxf(mb, m1);
        const float a0 = (count > 0) ? ex2(mb - m) : 0.0f, a1 = ex2(m1 - m);
        const float l_tot = l_h * a0 + l1 * a1;
        const float inv_l = 1.0f / l_tot;
        const long long obase = ((long long)b * p.Nq + qrow) * ((long long)p.H * D) + (long long)h * D;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (oacc[t][4 * g + e] * a0 + mg[(t * 16 + 4 * g + e) * 64 + lane] * a1) * inv_l;
                const long long idx = obase + t * 32 + 8 * g + 4 * hi;
                *(u32x2*)((short*)p.o + idx) = u32x2{sat_cvt2_pk(v[0], v[1]), sat_cvt2_pk(v[2], v[3])};
            }
        if (p.lse && hi == 0) p.lse[((long long)b * p.H + h) * p.Nq + qrow] = (m + log2f(l_tot)) * 0.6931471805599453f;
    }
}


// ---- W64: 64 queries per wave (two q-blocks share every K / V^T fragment read and the tile's scalar / staging work), workgroup = 2 waves
// = 128 queries, two waves per SIMD (256 registers).  Per 32-key half-tile: 4 K reads + 8 MFMAs (both q-blocks) -> 2 x 16 exponentials ->
// 4 V^T reads + 8 MFMAs.  Issue slots per 32-query tile: 16 MFMA + 32 exp (x2) + 32 add + 16 cvt + 8 LDS + ~8 other, vs 16 + 20 + ~15 LDS /
// scalar in the 32-query kernel.
template <int NKB, bool MASK, bool FIRST>
SAT_DEVICE void tile_w64(short (*k_lds)[ROW], short (*v_lds)[ROW], const bf16x8 (&qf)[2][4], f32x16 (&oacc)[2][2], f32x16 (&negm)[2], float (&mb)[2],
                         float (&l_run)[2], int l31, int hi, int kperm, int nvalid) {
#pragma unroll 1
    for (int kb = 0; kb < NKB; ++kb) {
        f32x16 x[2];
        auto qk = [&]() {
            SAT_SETPRIO(1);
            if (FIRST) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { x[0][r] = 0.0f; x[1][r] = 0.0f; }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 ka = frag(k_lds, kb * 32 + kperm, 16 * s + 8 * hi);
#pragma unroll
                for (int g = 0; g < 2; ++g) x[g] = sat_mfma_32x32x16_bf16(ka, qf[g][s], (!FIRST && s == 0) ? negm[g] : x[g]);
            }
            SAT_SETPRIO(0);
        };
        auto rowmax = [&](int g) {
            float tmax = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + (r & 7) + 8 * hi + 16 * (r >> 3);
                if (!MASK || key < nvalid) tmax = fmaxf(tmax, x[g][r]);
            }
            return halfmax(tmax);
        };
        float ps[2];
        auto expsum = [&]() {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float p0 = 0.f, p1 = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float a = ex2(x[g][2 * j]), c = ex2(x[g][2 * j + 1]);
                    if (MASK) {
                        const int key = kb * 32 + ((2 * j) & 7) + 8 * hi + 16 * ((2 * j) >> 3);
                        if (key >= nvalid) a = 0.f;
                        if (key + 1 >= nvalid) c = 0.f;
                    }
                    p0 += a; p1 += c;
                    x[g][2 * j] = a; x[g][2 * j + 1] = c;
                }
                ps[g] = p0 + p1;
            }
        };
        qk();
        if (FIRST && kb == 0) {          // the wave's very first half-tile: true row max
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float tmax = rowmax(g);
                mb[g] = tmax;
#pragma unroll
                for (int r = 0; r < 16; ++r) x[g][r] -= tmax;
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[g][r] = -mb[g];
            }
        } else if (FIRST) {              // second half of the first tile: shifted by hand (C was 0)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) x[g][r] -= mb[g];
        }
        expsum();
        if (!(FIRST && kb == 0)) {
            if (sat_wave_any(!(ps[0] <= 0.5f * SUMLIM) || !(ps[1] <= 0.5f * SUMLIM))) {      // 16 scores per lane and q-block here
                const bool was_first = FIRST;
                // rare: recompute the scores of this half-tile (C = -mb, or 0 + manual shift in the first tile), take the true max
                SAT_SETPRIO(1);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bf16x8 ka = frag(k_lds, kb * 32 + kperm, 16 * s + 8 * hi);
#pragma unroll
                    for (int g = 0; g < 2; ++g) x[g] = sat_mfma_32x32x16_bf16(ka, qf[g][s], s == 0 ? negm[g] : x[g]);
                }
                SAT_SETPRIO(0);
                (void)was_first;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float d = fmaxf(rowmax(g), 0.0f), alpha = ex2(-d);
                    mb[g] += d;
                    l_run[g] *= alpha;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) oacc[g][t][r] *= alpha;
#pragma unroll
                    for (int r = 0; r < 16; ++r) x[g][r] -= d;
#pragma unroll
                    for (int r = 0; r < 16; ++r) negm[g][r] = -mb[g];
                }
                expsum();
            }
        }
        l_run[0] += ps[0];
        l_run[1] += ps[1];
        bf16x8 pb[2][2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int u = 0; u < 2; ++u) pb[g][u] = pack8(x[g], u);
        SAT_SETPRIO(1);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8 va = frag(v_lds, t * 32 + l31, kb * 32 + 16 * u + 8 * hi);
#pragma unroll
                for (int g = 0; g < 2; ++g) oacc[g][t] = sat_mfma_32x32x16_bf16(va, pb[g][u], oacc[g][t]);
            }
        SAT_SETPRIO(0);
    }
}

SAT_DEVICE void attn_fwd_w64_body(const P& p, short (*k_lds2)[TK][ROW], short (*v_lds2)[D][ROW]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int kperm = kperm_of(l31);
    const int b = blockIdx.z, h = blockIdx.y;
    const int hk = h / (p.H / p.Hkv);
    const int q0 = blockIdx.x * 128 + wave * 64;
    const bool w_ok = q0 < p.Nq;
    const size_t qplane = ((size_t)b * p.H + h) * (size_t)p.Nqp * D;
    const size_t kplane = ((size_t)b * p.Hkv + hk) * (size_t)p.Nkp * D;
    const float sl2 = p.scale * 1.4426950408889634f;
    bf16x8 qf[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int qrow = q0 + g * 32 + l31;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (qrow < p.Nqp) qf[g][s] = *reinterpret_cast<const bf16x8*>(p.q + qplane + (size_t)qrow * D + 16 * s + 8 * hi);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) qf[g][s][e] = 0;
            }
            u32x4 w = __builtin_bit_cast(u32x4, qf[g][s]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = __builtin_bit_cast(float, w[j] << 16) * sl2, hi2 = __builtin_bit_cast(float, w[j] & 0xffff0000u) * sl2;
                w[j] = sat_cvt2_pk(lo, hi2);
            }
            qf[g][s] = __builtin_bit_cast(bf16x8, w);
        }
    }
    f32x16 oacc[2][2], negm[2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) { oacc[g][0][r] = 0.0f; oacc[g][1][r] = 0.0f; negm[g][r] = 0.0f; }
    float l_run[2] = {0.0f, 0.0f}, mb[2] = {0.0f, 0.0f};

    bf16x8 kreg[4], vreg[4];      // 512 + 512 pieces over 128 threads
    auto tile_load = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = threadIdx.x + j * 128, r = c >> 3, part = c & 7;
            kreg[j] = *reinterpret_cast<const bf16x8*>(p.k + kplane + (size_t)(k0 + r) * D + part * 8);
            vreg[j] = *reinterpret_cast<const bf16x8*>(p.vt + kplane + (size_t)r * p.Nkp + k0 + part * 8);
        }
    };
    auto tile_store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = threadIdx.x + j * 128, r = c >> 3, part = c & 7;
            *reinterpret_cast<bf16x8*>(&k_lds2[buf][r][part * 8]) = kreg[j];
            *reinterpret_cast<bf16x8*>(&v_lds2[buf][r][part * 8]) = vreg[j];
        }
    };
    tile_load(0);
    tile_store(0);
    if (TK < p.Nk) tile_load(TK);
    __syncthreads();
    int buf = 0, k0 = 0;
    if (p.Nk >= TK) {
        if (TK < p.Nk) {
            tile_store(1);
            if (2 * TK < p.Nk) tile_load(2 * TK);
        }
        if (w_ok) tile_w64<2, false, true>(k_lds2[0], v_lds2[0], qf, oacc, negm, mb, l_run, l31, hi, kperm, TK);
        __syncthreads();
        k0 = TK; buf = 1;
    }
    for (; k0 + TK <= p.Nk; k0 += TK, buf ^= 1) {
        if (k0 + TK < p.Nk) {
            tile_store(buf ^ 1);
            if (k0 + 2 * TK < p.Nk) tile_load(k0 + 2 * TK);
        }
        if (w_ok) tile_w64<2, false, false>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, TK);
        __syncthreads();
    }
    if (k0 < p.Nk && w_ok) {
        const int rem = p.Nk - k0;
        if (k0 == 0) {
            if (rem > 32) tile_w64<2, true, true>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
            else tile_w64<1, true, true>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
        } else if (rem > 32) tile_w64<2, true, false>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
        else tile_w64<1, true, false>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int qrow = q0 + g * 32 + l31;
        const float l_tot = l_run[g] + __shfl_xor(l_run[g], 32);
        const float inv_l = 1.0f / l_tot;
        if (qrow < p.Nq) {
            const long long obase = ((long long)b * p.Nq + qrow) * ((long long)p.H * D) + (long long)h * D;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const f32x4 v = {oacc[g][t][4 * gg] * inv_l, oacc[g][t][4 * gg + 1] * inv_l, oacc[g][t][4 * gg + 2] * inv_l, oacc[g][t][4 * gg + 3] * inv_l};
                    const long long idx = obase + t * 32 + 8 * gg + 4 * hi;
                    *(u32x2*)((short*)p.o + idx) = u32x2{sat_cvt2_pk(v[0], v[1]), sat_cvt2_pk(v[2], v[3])};
                }
            if (p.lse && hi == 0) p.lse[((long long)b * p.H + h) * p.Nq + qrow] = (mb[g] + log2f(l_tot)) * 0.6931471805599453f;
        }
    }
}

__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2))) attn_fwd_w64(P p) {
    __shared__ __attribute__((aligned(16))) short k_lds2[2][TK][ROW];
    __shared__ __attribute__((aligned(16))) short v_lds2[2][D][ROW];
    attn_fwd_w64_body(p, k_lds2, v_lds2);
}
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1))) attn_fwd_w64_1(P p) {      // 512 registers, one wave per SIMD
    __shared__ __attribute__((aligned(16))) short k_lds2[2][TK][ROW];
    __shared__ __attribute__((aligned(16))) short v_lds2[2][D][ROW];
    attn_fwd_w64_body(p, k_lds2, v_lds2);
}

extern "C" int satx_attention_fwd(int variant, const short* q, const short* k, const short* vt, void* o, float* lse, int B, int H, int Hkv,
                                  int Nq, int Nk, int Nqp, int Nkp, float scale, void* stream) {
    P p{q, k, vt, o, lse, B, H, Hkv, Nq, Nk, Nqp, Nkp, scale};
    dim3 grid((Nq + 127) / 128, H, B);
    hipStream_t st = (hipStream_t)stream;
    if (variant == 500) { hipLaunchKernelGGL(attn_fwd_w64, dim3((Nq + 127) / 128, H, B), dim3(128), 0, st, p); return hipGetLastError() == hipSuccess ? 0 : 1; }
    if (variant == 501) { hipLaunchKernelGGL(attn_fwd_w64_1, dim3((Nq + 127) / 128, H, B), dim3(128), 0, st, p); return hipGetLastError() == hipSuccess ? 0 : 1; }
    if (variant == 415) { hipLaunchKernelGGL(attn_fwd_ks<15>, dim3((Nq + 63) / 64, H, B), dim3(256), 0, st, p); return hipGetLastError() == hipSuccess ? 0 : 1; }
    switch (variant) {
        case 0: hipLaunchKernelGGL(attn_fwd_x<0>, grid, dim3(256), 0, st, p); break;
        case 1: hipLaunchKernelGGL(attn_fwd_x<1>, grid, dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL(attn_fwd_x<3>, grid, dim3(256), 0, st, p); break;
        case 7: hipLaunchKernelGGL(attn_fwd_x<7>, grid, dim3(256), 0, st, p); break;
        case 8: hipLaunchKernelGGL(attn_fwd_x<8>, grid, dim3(256), 0, st, p); break;
        case 11: hipLaunchKernelGGL(attn_fwd_x<11>, grid, dim3(256), 0, st, p); break;
        case 15: hipLaunchKernelGGL(attn_fwd_x<15>, grid, dim3(256), 0, st, p); break;
        case 200: hipLaunchKernelGGL(attn_fwd_v2<0>, grid, dim3(256), 0, st, p); break;
        case 264: hipLaunchKernelGGL(attn_fwd_v2<64>, grid, dim3(256), 0, st, p); break;
        case 22: hipLaunchKernelGGL(attn_fwd_p<6>, grid, dim3(256), 0, st, p); break;
        case 30: hipLaunchKernelGGL(attn_fwd_p<14>, grid, dim3(256), 0, st, p); break;
        case 54: hipLaunchKernelGGL(attn_fwd_p<38>, grid, dim3(256), 0, st, p); break;
        case 62: hipLaunchKernelGGL(attn_fwd_p<46>, grid, dim3(256), 0, st, p); break;
        case 122: hipLaunchKernelGGL(attn_fwd_p2<6>, grid, dim3(256), 0, st, p); break;
        case 130: hipLaunchKernelGGL(attn_fwd_p2<14>, grid, dim3(256), 0, st, p); break;
        case 154: hipLaunchKernelGGL(attn_fwd_p2<38>, grid, dim3(256), 0, st, p); break;
        case 162: hipLaunchKernelGGL(attn_fwd_p2<46>, grid, dim3(256), 0, st, p); break;
        default: return 2;
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
