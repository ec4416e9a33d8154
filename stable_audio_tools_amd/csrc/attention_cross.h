// attention_cross.h — attention against a SHORT key sequence (the DiT's cross-attention: 130 conditioning tokens, 24 query
// heads on 12 key / value heads) — included by attention.hip, bf16 planes only (the fp32 two-plane mode stays on the general kernels).
//
// Replaces, for Nk <= 256, the flash-style kernels above on the reference's cross-attention call
// (stable_audio_tools/models/transformer.py:351-357, :459-472: to_q / to_kv projections, :408-411 repeat_interleave of the kv heads,
// :440 scaled_dot_product_attention without a mask).  The general forward walks 64-key tiles with an online softmax, re-stages K / V^T
// per workgroup and tile and knows nothing about the two query heads that share a kv head: at 130 keys that is three tiles (one of
// them 97 % padding) of machinery per 32 queries — 11.4 us per launch at B x H = 48, 0.057 of the bf16 peak (round 5).  Here:
//   * ALL keys of a (batch item, kv head) are staged into LDS ONCE per workgroup (K row-major + V^T, 2 x 23 KB at 160 padded keys);
//     the workgroup then walks "wave blocks" = (query head of the GQA group, 32-query block) — both query heads are served from
//     the one copy; a wave's blocks are independent, so after the staging barrier there is no barrier at all;
//   * EXACT softmax in one pass: the NKB x 16 scores of a lane's query row are all in registers (keys padded to a multiple of 32,
//     not 64), so there is no running max, no rescale, no deferred-max branch;
//   * backward: a dQ kernel of the same shape (K, V, K^T resident, probabilities recomputed from the LSE, one key block at a time)
//     and a dK / dV kernel whose waves own one 32-key block each and walk a RANGE of the group's query tiles — the ranges'
//     partial sums go to fp32 slabs that sat_attn_cross_reduce adds in index order (no atomics: bit-reproducible).
// Fragment conventions are the file's (swapped products, sat_att_kperm, accumulator rows = MFMA k-slots).

struct SatXAttnParams {
    const short* q_rm;      // [B][H][Nqp][64]
    const short* k_rm;      // [B][Hkv][Nkp][64]
    const short* v_rm;      // [B][Hkv][Nkp][64]   (backward)
    const short* k_tr;      // [B][Hkv][64][Nkp]   (backward)
    const short* v_tr;      // [B][Hkv][64][Nkp]   (forward)
    const short* q_tr;      // [B][H][64][Nqp]     (backward)
    const short* do_rm;     // [B][H][Nqp][64]
    const short* do_tr;     // [B][H][64][Nqp]
    void* o;                // (B, Nq, H*64) bf16
    float* lse;             // (B, H, Nq)
    const float* dsum;      // (B, H, Nq)
    void* dq;               // (B, H, Nq, 64) bf16
    float* part;            // dK / dV slabs [nsplit][2][B][Hkv][Nk][64] fp32
    int B, H, Hkv, Nq, Nk, Nqp, Nkp;
    int nqb;                // 32-query blocks per head
    int per_wg;             // wave blocks (fwd, dQ) / 64-query tiles (dK, dV) per workgroup
    float scale;
};

// stage a plane tile into padded LDS rows with 256 threads, 16 bytes per piece.  NPT = pieces per thread (compile time): ALL loads are
// issued before the first LDS store — a rolled copy loop waits for each load in turn (the first version of these kernels staged
// 2 x 5 pieces per thread behind ten serial L2 / HBM latencies).  `parts` = 16-byte pieces per row; rows * parts == 256 * NPT.
template <int NPT, int LROW>
SAT_DEVICE void sat_xatt_stage(short (*dst)[LROW], const short* src, size_t rstride, int parts) {
    bf16x8 reg[NPT];
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int c = threadIdx.x + j * 256;
        const int r = c / parts, part = c - r * parts;
        reg[j] = *reinterpret_cast<const bf16x8*>(src + (size_t)r * rstride + part * 8);
    }
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int c = threadIdx.x + j * 256;
        const int r = c / parts, part = c - r * parts;
        *reinterpret_cast<bf16x8*>(&dst[r][part * 8]) = reg[j];
    }
}
// max(a, b, c) as ONE v_max3_f32.  fmaxf on MFMA results makes hipcc emit a canonicalising v_max_f32 v, v, v per operand first
// (cdna_hip_programming.md, "fused attention prefill": 75 of them for the 80 scores of a query row here); the asm form takes the
// accumulator registers as they are.  NaN scores propagate differently from fmaxf — a NaN row is NaN either way.
SAT_DEVICE float sat_xatt_max3(float a, float b, float c) {
#if defined(SAT_HIPEMU)
    return fmaxf(fmaxf(a, b), c);
#else
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
#endif
}
// max of the 16 accumulator registers as a tree of eight v_max3 (a serial fmaxf chain over 80 scores is 80 dependent VALU latencies)
SAT_DEVICE float sat_xatt_max16(const f32x16& v) {
    const float a0 = sat_xatt_max3(v[0], v[1], v[2]), a1 = sat_xatt_max3(v[3], v[4], v[5]), a2 = sat_xatt_max3(v[6], v[7], v[8]);
    const float a3 = sat_xatt_max3(v[9], v[10], v[11]), a4 = sat_xatt_max3(v[12], v[13], v[14]);
    return sat_xatt_max3(sat_xatt_max3(a0, a1, a2), sat_xatt_max3(a3, a4, v[15]), v[15]);
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int NKB>
__global__ void __launch_bounds__(256)
#if !defined(SAT_HIPEMU)
__attribute__((amdgpu_waves_per_eu(2, 2)))      // 256 registers: the scores stay in VGPRs (no v_accvgpr_read per score)
#endif
sat_attn_cross_fwd_kernel(SatXAttnParams p) {
    constexpr int NKEY = NKB * 32, VROW = NKEY + 8;      // V^T rows of 16 (4 NKB + 1) bytes: odd multiple of 16 -> conflict-free b128 reads
    __shared__ __attribute__((aligned(16))) short k_lds[NKEY][SAT_ATT_ROW];     // [key][d]
    __shared__ __attribute__((aligned(16))) short v_lds[SAT_ATT_D][VROW];       // [d][key]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int kperm = sat_att_kperm(l31);
    const int b = blockIdx.z, hk = blockIdx.y;
    const int group = p.H / p.Hkv;
    const int nwb = group * p.nqb;
    const int wb0 = blockIdx.x * p.per_wg;
    const int wb1 = (wb0 + p.per_wg < nwb) ? wb0 + p.per_wg : nwb;
    const size_t kplane = ((size_t)b * p.Hkv + hk) * (size_t)p.Nkp * SAT_ATT_D;
    const float sl2 = p.scale * 1.4426950408889634f;
    const int nvalid = p.Nk - (NKB - 1) * 32;            // valid keys of the last 32-key block: 1..32

    u32x4 qraw[4];
    auto load_q = [&](int wb) {
        const int g = wb / p.nqb, qb = wb - g * p.nqb;
        const size_t qplane = ((size_t)b * p.H + (size_t)hk * group + g) * (size_t)p.Nqp * SAT_ATT_D;
        const short* src = p.q_rm + qplane + (size_t)(qb * 32 + l31) * SAT_ATT_D + 8 * hi;
#pragma unroll
        for (int s = 0; s < 4; ++s) qraw[s] = *reinterpret_cast<const u32x4*>(src + 16 * s);
    };
    int wb = wb0 + wave;
    if (wb < wb1) load_q(wb);
    sat_xatt_stage<NKB, SAT_ATT_ROW>(k_lds, p.k_rm + kplane, SAT_ATT_D, 8);                 // NKEY rows x 8 pieces
    sat_xatt_stage<NKB, VROW>(v_lds, p.v_tr + kplane, (size_t)p.Nkp, NKEY / 8);             // 64 rows x NKEY / 8 pieces
    __syncthreads();

    for (; wb < wb1; wb += 4) {
        const int g = wb / p.nqb, qb = wb - g * p.nqb;
        const int h = hk * group + g;
        const int qrow = qb * 32 + l31;
        bf16x8 qf[4];      // B operand of x = K (Q c)^T, c = scale * log2(e): the scores leave the matrix pipe in the exp2 domain
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 w = qraw[s];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e0 = __builtin_bit_cast(float, w[j] << 16), e1 = __builtin_bit_cast(float, w[j] & 0xffff0000u);
                w[j] = sat_cvt2_pk(e0 * sl2, e1 * sl2);
            }
            qf[s] = __builtin_bit_cast(bf16x8, w);
        }
        if (wb + 4 < wb1) load_q(wb + 4);      // the next block's rows travel under this block's math

        f32x16 sacc[NKB];
        SAT_SETPRIO(1);
        const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < 4; ++s)            // k-step outside: the NKB accumulator chains interleave (a dependent MFMA waits 64 cycles)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
                sacc[kb] = sat_mfma_32x32x16_bf16(sat_att_frag_rm(k_lds, kb * 32 + kperm, 16 * s + 8 * hi), qf[s], s == 0 ? zero : sacc[kb]);
        SAT_SETPRIO(0);
        // register r of block kb is key kb * 32 + (r & 7) + 8 hi + 16 (r >> 3); only the last block holds padded keys
        if (nvalid < 32) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r & 7) + 8 * hi + 16 * (r >> 3) >= nvalid) sacc[NKB - 1][r] = -INFINITY;
        }
        float tmax = sat_xatt_max16(sacc[0]);
#pragma unroll
        for (int kb = 1; kb < NKB; ++kb) {
            const float m = sat_xatt_max16(sacc[kb]);
            tmax = sat_xatt_max3(tmax, m, m);
        }
        const float mb = sat_att_halfmax(tmax);      // finite: key 0 is always valid
        f32x16 oacc[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            oacc[0][r] = 0.0f;
            oacc[1][r] = 0.0f;
        }
        float ps0 = 0.0f, ps1 = 0.0f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = sat_exp2(sacc[kb][r] - mb);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (kb == NKB - 1 && u == 1 && nvalid <= 16) break;      // keys 16..31 of the last block are all padding: P = 0
                bf16x8 pb[1];
                sat_att_pack<1>(sacc[kb], u, pb);
                const u32x4 w = __builtin_bit_cast(u32x4, pb[0]);        // the normaliser is the sum of the ROUNDED probabilities
                ps0 = sat_att_sum2(w[0], ps0);
                ps1 = sat_att_sum2(w[1], ps1);
                ps0 = sat_att_sum2(w[2], ps0);
                ps1 = sat_att_sum2(w[3], ps1);
                SAT_SETPRIO(1);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    oacc[t] = sat_mfma_32x32x16_bf16(sat_att_frag_rm(v_lds, t * 32 + l31, kb * 32 + 16 * u + 8 * hi), pb[0], oacc[t]);
                SAT_SETPRIO(0);
            }
        }
        const float l_run = ps0 + ps1;
        const float l_tot = l_run + __shfl_xor(l_run, 32);
        const float inv_l = 1.0f / l_tot;
        if (qrow < p.Nq) {
            const long long obase = ((long long)b * p.Nq + qrow) * ((long long)p.H * SAT_ATT_D) + (long long)h * SAT_ATT_D;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const long long idx = obase + t * 32 + 8 * gq + 4 * hi;
                    *(u32x2*)((short*)p.o + idx) = u32x2{sat_cvt2_pk(oacc[t][4 * gq] * inv_l, oacc[t][4 * gq + 1] * inv_l),
                                                         sat_cvt2_pk(oacc[t][4 * gq + 2] * inv_l, oacc[t][4 * gq + 3] * inv_l)};
                }
            if (p.lse && hi == 0) p.lse[((long long)b * p.H + h) * p.Nq + qrow] = (mb + log2f(l_tot)) * 0.6931471805599453f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward, dQ:  S^T = K (Q c)^T - lse ; P^T = exp2(S^T) ; dP^T - D = V dO^T - D ; dS^T / scale = P^T (dP^T - D) ; dQ^T += K^T dS^T
// one 32-key block at a time (the LSE is known: no cross-block dependency), K / V / K^T resident in LDS
// ---------------------------------------------------------------------------------------------
template <int NKB>
__global__ void __launch_bounds__(256)
#if !defined(SAT_HIPEMU)
__attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
sat_attn_cross_dq_kernel(SatXAttnParams p) {
    constexpr int NKEY = NKB * 32, VROW = NKEY + 8;
    __shared__ __attribute__((aligned(16))) short k_lds[NKEY][SAT_ATT_ROW];     // [key][d]
    __shared__ __attribute__((aligned(16))) short v_lds[NKEY][SAT_ATT_ROW];     // [key][d]
    __shared__ __attribute__((aligned(16))) short kt_lds[SAT_ATT_D][VROW];      // [d][key]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int kperm = sat_att_kperm(l31);
    const int b = blockIdx.z, hk = blockIdx.y;
    const int group = p.H / p.Hkv;
    const int nwb = group * p.nqb;
    const int wb0 = blockIdx.x * p.per_wg;
    const int wb1 = (wb0 + p.per_wg < nwb) ? wb0 + p.per_wg : nwb;
    const size_t kplane = ((size_t)b * p.Hkv + hk) * (size_t)p.Nkp * SAT_ATT_D;
    const float l2e = 1.4426950408889634f;
    const float sl2 = p.scale * l2e;

    u32x4 qraw[4], graw[4];
    float lse_raw = 0.0f, ds_raw = 0.0f;
    auto load_q = [&](int wb) {
        const int g = wb / p.nqb, qb = wb - g * p.nqb;
        const int h = hk * group + g, qrow = qb * 32 + l31;
        const size_t off = ((size_t)b * p.H + h) * (size_t)p.Nqp * SAT_ATT_D + (size_t)qrow * SAT_ATT_D + 8 * hi;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qraw[s] = *reinterpret_cast<const u32x4*>(p.q_rm + off + 16 * s);
            graw[s] = *reinterpret_cast<const u32x4*>(p.do_rm + off + 16 * s);
        }
        const bool ok = qrow < p.Nq;
        lse_raw = ok ? p.lse[((long long)b * p.H + h) * p.Nq + qrow] : 0.0f;
        ds_raw = ok ? p.dsum[((long long)b * p.H + h) * p.Nq + qrow] : 0.0f;
    };
    int wb = wb0 + wave;
    if (wb < wb1) load_q(wb);
    sat_xatt_stage<NKB, SAT_ATT_ROW>(k_lds, p.k_rm + kplane, SAT_ATT_D, 8);
    sat_xatt_stage<NKB, SAT_ATT_ROW>(v_lds, p.v_rm + kplane, SAT_ATT_D, 8);
    sat_xatt_stage<NKB, VROW>(kt_lds, p.k_tr + kplane, (size_t)p.Nkp, NKEY / 8);
    __syncthreads();

    for (; wb < wb1; wb += 4) {
        const int g = wb / p.nqb, qb = wb - g * p.nqb;
        const int h = hk * group + g;
        const int qrow = qb * 32 + l31;
        bf16x8 qf[4], gf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 w = qraw[s];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e0 = __builtin_bit_cast(float, w[j] << 16), e1 = __builtin_bit_cast(float, w[j] & 0xffff0000u);
                w[j] = sat_cvt2_pk(e0 * sl2, e1 * sl2);
            }
            qf[s] = __builtin_bit_cast(bf16x8, w);
            gf[s] = __builtin_bit_cast(bf16x8, graw[s]);
        }
        const float nl = -lse_raw * l2e, nd = -ds_raw;
        if (wb + 4 < wb1) load_q(wb + 4);
        f32x16 negl, negd, dq[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            negl[r] = nl;
            negd[r] = nd;
            dq[0][r] = 0.0f;
            dq[1][r] = 0.0f;
        }
        // padded keys need no mask: their K^T columns are zero (sat_attn_prepare), so whatever dS^T holds there adds nothing.
        // A real loop: unrolled, the scheduler hoists every block's fragment reads and spills (NKB >= 3: 56-208 registers).
#pragma unroll 1
        for (int kb = 0; kb < NKB; ++kb) {
            f32x16 sacc, pacc;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                sacc = sat_mfma_32x32x16_bf16(sat_att_frag_rm(k_lds, kb * 32 + kperm, 16 * s + 8 * hi), qf[s], s == 0 ? negl : sacc);
                pacc = sat_mfma_32x32x16_bf16(sat_att_frag_rm(v_lds, kb * 32 + kperm, 16 * s + 8 * hi), gf[s], s == 0 ? negd : pacc);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[r] = sat_exp2(sacc[r]) * pacc[r];      // dS^T / scale
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bf16x8 sb[1];
                sat_att_pack<1>(sacc, u, sb);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    dq[t] = sat_mfma_32x32x16_bf16(sat_att_frag_rm(kt_lds, t * 32 + l31, kb * 32 + 16 * u + 8 * hi), sb[0], dq[t]);
            }
        }
        if (qrow < p.Nq) {
            // lane (q, hi) holds d = 32 t + 8 gq + 4 hi + {0..3} in registers 4 gq + {0..3}: 8-byte stores
            const long long obase = (((long long)b * p.H + h) * p.Nq + qrow) * SAT_ATT_D;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const long long idx = obase + t * 32 + 8 * gq + 4 * hi;
                    *(u32x2*)((short*)p.dq + idx) = u32x2{sat_cvt2_pk(dq[t][4 * gq] * p.scale, dq[t][4 * gq + 1] * p.scale),
                                                          sat_cvt2_pk(dq[t][4 * gq + 2] * p.scale, dq[t][4 * gq + 3] * p.scale)};
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward, dK / dV.  Wave w < NKB owns keys [32 w, 32 w + 32) (lanes = keys); the workgroup walks the 64-query tiles
// [it0, it1) of the kv group's query heads (tile index = head-in-group * nqt + tile), staged through a double-buffered LDS
// ring as in sat_attn_bwd_dkv_bf16_kernel; the range's sums go to slab blockIdx.x.
//   S = Q K^T ; P = exp2(S * scale * log2 e - lse) ; dV^T += dO^T P ; dP - D = dO V^T - D ; dS / scale = P (dP - D) ; dK^T += Q^T dS
// PRECONDITION (sat_attn_prepare): rows [N, Np) of every operand plane are zero (a padded query row adds exactly zero).
// ---------------------------------------------------------------------------------------------
template <int NKB>
__global__ void __launch_bounds__((NKB < 4 ? 4 : NKB) * 64)
#if !defined(SAT_HIPEMU)
__attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
sat_attn_cross_dkv_kernel(SatXAttnParams p) {
    constexpr int NW = NKB < 4 ? 4 : NKB, NT = NW * 64;
    constexpr int TQ = 64, TROW = TQ + 8;
    constexpr int PP = (512 + NT - 1) / NT;      // 16-byte pieces per thread of a [64][64] bf16 tile
    __shared__ __attribute__((aligned(16))) short q_lds2[2][TQ][SAT_ATT_ROW];      // [buffer][q][d]
    __shared__ __attribute__((aligned(16))) short g_lds2[2][TQ][SAT_ATT_ROW];      // dO [q][d]
    __shared__ __attribute__((aligned(16))) short qt_lds2[2][SAT_ATT_D][TROW];     // [d][q]
    __shared__ __attribute__((aligned(16))) short gt_lds2[2][SAT_ATT_D][TROW];     // dO^T [d][q]
    __shared__ __attribute__((aligned(16))) float nl_lds2[2][TQ], nd_lds2[2][TQ];  // -lse * log2(e), -D of the tile's queries
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, hk = blockIdx.y;
    const int group = p.H / p.Hkv;
    const int krow = wave * 32 + l31;
    const bool w_on = wave < NKB;                       // wave-uniform: this wave owns a key block
    const bool k_ok = w_on && krow < p.Nk;
    const size_t kplane = ((size_t)b * p.Hkv + hk) * (size_t)p.Nkp * SAT_ATT_D;
    const float l2e = 1.4426950408889634f;
    const float sl2 = p.scale * l2e;

    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        u32x4 w = u32x4{0u, 0u, 0u, 0u}, v = u32x4{0u, 0u, 0u, 0u};
        if (w_on) {      // krow < NKB * 32 <= Nkp
            w = *reinterpret_cast<const u32x4*>(p.k_rm + kplane + (size_t)krow * SAT_ATT_D + 16 * s + 8 * hi);
            v = *reinterpret_cast<const u32x4*>(p.v_rm + kplane + (size_t)krow * SAT_ATT_D + 16 * s + 8 * hi);
        }
        kf[s] = __builtin_bit_cast(bf16x8, w);
        vf[s] = __builtin_bit_cast(bf16x8, v);
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dk[t][r] = 0.0f;
            dv[t][r] = 0.0f;
        }

    const int nqt = (p.Nq + TQ - 1) / TQ, ntiles = group * nqt;
    const int it0 = blockIdx.x * p.per_wg;
    const int it1 = (it0 + p.per_wg < ntiles) ? it0 + p.per_wg : ntiles;
    const size_t gplane = ((size_t)b * p.H + (size_t)hk * group) * (size_t)p.Nqp * SAT_ATT_D;
    bf16x8 rq[PP], rg[PP], rqt[PP], rgt[PP];
    float r_nl = 0.0f, r_nd = 0.0f;
    auto tile_load = [&](int it) {
        const int hg = it / nqt, q0 = (it - hg * nqt) * TQ;
        const size_t hplane = gplane + (size_t)hg * (size_t)p.Nqp * SAT_ATT_D;
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const int c = threadIdx.x + j * NT;
            if (c < 512) {
                const int r = c >> 3, part = c & 7;
                rq[j] = *reinterpret_cast<const bf16x8*>(p.q_rm + hplane + (size_t)(q0 + r) * SAT_ATT_D + part * 8);
                rg[j] = *reinterpret_cast<const bf16x8*>(p.do_rm + hplane + (size_t)(q0 + r) * SAT_ATT_D + part * 8);
                rqt[j] = *reinterpret_cast<const bf16x8*>(p.q_tr + hplane + (size_t)r * p.Nqp + q0 + part * 8);
                rgt[j] = *reinterpret_cast<const bf16x8*>(p.do_tr + hplane + (size_t)r * p.Nqp + q0 + part * 8);
            }
        }
        if (threadIdx.x < TQ) {
            const int q = q0 + threadIdx.x;
            const bool ok = q < p.Nq;
            const long long i = ((long long)b * p.H + (hk * group + hg)) * p.Nq + q;
            r_nl = ok ? -p.lse[i] * l2e : 0.0f;
            r_nd = ok ? -p.dsum[i] : 0.0f;
        }
    };
    auto tile_store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < PP; ++j) {
            const int c = threadIdx.x + j * NT;
            if (c < 512) {
                const int r = c >> 3, part = c & 7;
                *reinterpret_cast<bf16x8*>(&q_lds2[buf][r][part * 8]) = rq[j];
                *reinterpret_cast<bf16x8*>(&g_lds2[buf][r][part * 8]) = rg[j];
                *reinterpret_cast<bf16x8*>(&qt_lds2[buf][r][part * 8]) = rqt[j];
                *reinterpret_cast<bf16x8*>(&gt_lds2[buf][r][part * 8]) = rgt[j];
            }
        }
        if (threadIdx.x < TQ) {
            nl_lds2[buf][threadIdx.x] = r_nl;
            nd_lds2[buf][threadIdx.x] = r_nd;
        }
    };

    if (it0 < it1) {
        tile_load(it0);
        tile_store(0);
        if (it0 + 1 < it1) tile_load(it0 + 1);
    }
    __syncthreads();
    for (int it = it0, buf = 0; it < it1; ++it, buf ^= 1) {
        short (*q_lds)[SAT_ATT_ROW] = q_lds2[buf];
        short (*g_lds)[SAT_ATT_ROW] = g_lds2[buf];
        short (*qt_lds)[TROW] = qt_lds2[buf];
        short (*gt_lds)[TROW] = gt_lds2[buf];
        const float* nl_lds = nl_lds2[buf];
        const float* nd_lds = nd_lds2[buf];
        if (it + 1 < it1) {
            tile_store(buf ^ 1);
            if (it + 2 < it1) tile_load(it + 2);
        }
        if (w_on) {
#pragma unroll
            for (int qb = 0; qb < TQ / 32; ++qb) {
                // register r = 4 g + e of an accumulator is query row qb * 32 + 8 g + 4 hi + e: four consecutive floats per g
                f32x4 a4[4], c4[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    a4[g] = *reinterpret_cast<const f32x4*>(&nl_lds[qb * 32 + 8 * g + 4 * hi]);
                    c4[g] = *reinterpret_cast<const f32x4*>(&nd_lds[qb * 32 + 8 * g + 4 * hi]);
                }
                const f32x16 nl = sat_att_cat4(a4[0], a4[1], a4[2], a4[3]);
                f32x16 pacc = sat_att_cat4(c4[0], c4[1], c4[2], c4[3]);
                const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                f32x16 sacc;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    sacc = sat_mfma_32x32x16_bf16(sat_att_frag_rm(q_lds, qb * 32 + l31, 16 * s + 8 * hi), kf[s], s == 0 ? zero : sacc);   // S[q][key]
                    pacc = sat_mfma_32x32x16_bf16(sat_att_frag_rm(g_lds, qb * 32 + l31, 16 * s + 8 * hi), vf[s], pacc);                   // dP[q][key] - D
                }
                f32x16 pr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pr[r] = sat_exp2(fmaf(sacc[r], sl2, nl[r]));
                    sacc[r] = pr[r] * pacc[r];                          // dS[q][key] / scale
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    bf16x8 pb[1], sb[1];
                    sat_att_pack<1>(pr, u, pb);
                    sat_att_pack<1>(sacc, u, sb);
                    const int qofs = qb * 32 + 16 * u + 4 * hi;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        dv[t] = sat_mfma_32x32x16_bf16(sat_att_frag_acc(gt_lds, t * 32 + l31, qofs), pb[0], dv[t]);
                        dk[t] = sat_mfma_32x32x16_bf16(sat_att_frag_acc(qt_lds, t * 32 + l31, qofs), sb[0], dk[t]);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (k_ok) {
        // slab blockIdx.x: [2][B][Hkv][Nk][64] fp32 (dK / scale first, then dV); lane (key, hi) holds d = 32 t + 8 g + 4 hi + {0..3}
        const size_t slab = (size_t)p.B * p.Hkv * p.Nk * SAT_ATT_D;
        float* base = p.part + (size_t)blockIdx.x * 2 * slab + (((size_t)b * p.Hkv + hk) * p.Nk + krow) * SAT_ATT_D;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = t * 32 + 8 * g + 4 * hi;
                *reinterpret_cast<f32x4*>(base + d) = f32x4{dk[t][4 * g], dk[t][4 * g + 1], dk[t][4 * g + 2], dk[t][4 * g + 3]};
                *reinterpret_cast<f32x4*>(base + slab + d) = f32x4{dv[t][4 * g], dv[t][4 * g + 1], dv[t][4 * g + 2], dv[t][4 * g + 3]};
            }
    }
}

// dk = bf16(scale * sum_s part[s][0]), dv = bf16(sum_s part[s][1]), slabs added in index order; 4 elements per thread
struct SatXAttnReduceParams {
    const float* part;
    short* dk;
    short* dv;
    long long n4;      // B * Hkv * Nk * 64 / 4
    int nsplit;
    float scale;
};
__global__ void __launch_bounds__(256) sat_attn_cross_reduce_kernel(SatXAttnReduceParams p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * p.n4) return;
    const bool is_v = i >= p.n4;
    const long long e = is_v ? i - p.n4 : i;
    const size_t slab = (size_t)p.n4 * 4;
    const float* src = p.part + (is_v ? slab : 0) + (size_t)e * 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.nsplit; ++s) a += *reinterpret_cast<const f32x4*>(src + (size_t)s * 2 * slab);
    const float c = is_v ? 1.0f : p.scale;
    *reinterpret_cast<u32x2*>((is_v ? p.dv : p.dk) + e * 4) = u32x2{sat_cvt2_pk(a[0] * c, a[1] * c), sat_cvt2_pk(a[2] * c, a[3] * c)};
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
#define SAT_XATT_MAXK 256

// 1 when the short-key kernels serve this shape (bf16 planes, head dim 64, Nk <= 256), else 0
extern "C" int sat_attention_cross_ok(int H, int Hkv, int Nk, int head_dim, int dtype) {
    return (dtype == 1 && head_dim == SAT_ATT_D && Hkv > 0 && H % Hkv == 0 && Nk > 0 && Nk <= SAT_XATT_MAXK) ? 1 : 0;
}

// wave blocks per workgroup: enough workgroups to put ~4 on every CU (three are co-resident at 160 keys; a finer grain keeps the last
// round's imbalance small) while a workgroup's one-time K / V staging is shared by at least four wave blocks (one per wave)
static int sat_xatt_per_wg(int units_bhk, int nwb) {
    const int target = 4 * sat_cu_count();
    int nchunks = sat_cdiv(target, units_bhk);
    if (nchunks < 1) nchunks = 1;
    int per = sat_cdiv(nwb, nchunks);
    per = sat_cdiv(per, 4) * 4;
    if (per < 4) per = 4;
    return per;
}

// number of dK / dV slabs sat_attention_cross_bwd writes for this shape (the workspace holds nsplit * 2 * B * Hkv * Nk * 64 floats)
static int sat_xatt_dkv_plan(int B, int H, int Hkv, int Nq, int* per_wg) {
    const int ntiles = (H / Hkv) * sat_cdiv(Nq, 64);
    int want = sat_cdiv(2 * sat_cu_count(), B * Hkv);
    if (want < 1) want = 1;
    if (want > ntiles) want = ntiles;
    const int per = sat_cdiv(ntiles, want);
    *per_wg = per;
    return sat_cdiv(ntiles, per);
}
extern "C" long long sat_attention_cross_bwd_ws(int B, int H, int Hkv, int Nq, int Nk) {
    if (B <= 0 || H <= 0 || Hkv <= 0 || Nq <= 0 || Nk <= 0 || H % Hkv) return -1;
    int per;
    const int ns = sat_xatt_dkv_plan(B, H, Hkv, Nq, &per);
    return (long long)ns * 2 * B * Hkv * Nk * SAT_ATT_D * (long long)sizeof(float);
}

#define SAT_XATT_DISPATCH(KERNEL, nkb, grid, block, stream, p)                                   \
    switch (nkb) {                                                                               \
        case 1: SAT_LAUNCH((KERNEL<1>), grid, block(1), stream, p); break;                       \
        case 2: SAT_LAUNCH((KERNEL<2>), grid, block(2), stream, p); break;                       \
        case 3: SAT_LAUNCH((KERNEL<3>), grid, block(3), stream, p); break;                       \
        case 4: SAT_LAUNCH((KERNEL<4>), grid, block(4), stream, p); break;                       \
        case 5: SAT_LAUNCH((KERNEL<5>), grid, block(5), stream, p); break;                       \
        case 6: SAT_LAUNCH((KERNEL<6>), grid, block(6), stream, p); break;                       \
        case 7: SAT_LAUNCH((KERNEL<7>), grid, block(7), stream, p); break;                       \
        default: SAT_LAUNCH((KERNEL<8>), grid, block(8), stream, p); break;                      \
    }
#define SAT_XATT_B256(n) dim3(256)
#define SAT_XATT_BNW(n) dim3(((n) < 4 ? 4 : (n)) * 64)

// forward: q_rm (B,H,Nqp,64), k_rm (B,Hkv,Nkp,64), v_tr (B,Hkv,64,Nkp) bf16 planes (sat_attn_prepare / the projection epilogue)
// -> o (B, Nq, H*64) bf16 [, lse (B,H,Nq)]
extern "C" int sat_attention_cross_fwd(const short* q_rm, const short* k_rm, const short* v_tr, void* o, float* lse, int B, int H,
                                       int Hkv, int Nq, int Nk, int Nqp, int Nkp, int head_dim, float scale, void* stream) {
    if (sat_attn_check(B, H, Hkv, Nq, Nk, Nqp, Nkp, head_dim, 1, "sat_attention_cross_fwd: empty shape")) return 1;
    if (!sat_attention_cross_ok(H, Hkv, Nk, head_dim, 1)) { sat_set_error("sat_attention_cross_fwd: Nk > 256 (sat_attention_cross_ok)"); return 1; }
    if (!q_rm || !k_rm || !v_tr || !o) { sat_set_error("sat_attention_cross_fwd: missing buffer"); return 1; }
    SatXAttnParams p{};
    p.q_rm = q_rm; p.k_rm = k_rm; p.v_tr = v_tr; p.o = o; p.lse = lse;
    p.B = B; p.H = H; p.Hkv = Hkv; p.Nq = Nq; p.Nk = Nk; p.Nqp = Nqp; p.Nkp = Nkp; p.scale = scale;
    p.nqb = sat_cdiv(Nq, 32);
    const int nwb = (H / Hkv) * p.nqb;
    p.per_wg = sat_xatt_per_wg(B * Hkv, nwb);
    dim3 grid(sat_cdiv(nwb, p.per_wg), Hkv, B);
    SAT_XATT_DISPATCH(sat_attn_cross_fwd_kernel, sat_cdiv(Nk, 32), grid, SAT_XATT_B256, stream, p);
    return sat_check_launch("sat_attention_cross_fwd");
}

// backward: planes[16] as sat_attention_bwd ({q_rm, k_rm, v_rm, k_tr, q_tr, do_rm, do_tr, -} x {hi, lo}; only the hi planes are
// read); ws: sat_attention_cross_bwd_ws bytes, 16-byte aligned.  dq (B,H,Nq,64), dk / dv (B,Hkv,Nk,64) bf16.
extern "C" int sat_attention_cross_bwd(const short* const* planes, const float* lse, const float* dsum, void* dq, void* dk,
                                       void* dv, void* ws, long long ws_bytes, int B, int H, int Hkv, int Nq, int Nk, int Nqp,
                                       int Nkp, int head_dim, float scale, void* stream) {
    if (sat_attn_check(B, H, Hkv, Nq, Nk, Nqp, Nkp, head_dim, 1, "sat_attention_cross_bwd: empty shape")) return 1;
    if (!sat_attention_cross_ok(H, Hkv, Nk, head_dim, 1)) { sat_set_error("sat_attention_cross_bwd: Nk > 256 (sat_attention_cross_ok)"); return 1; }
    if (!planes || !lse || !dsum || !dq || !dk || !dv || !ws) { sat_set_error("sat_attention_cross_bwd: missing buffer"); return 1; }
    if (ws_bytes < sat_attention_cross_bwd_ws(B, H, Hkv, Nq, Nk) || ((uintptr_t)ws & 15)) { sat_set_error("sat_attention_cross_bwd: workspace too small or misaligned (sat_attention_cross_bwd_ws)"); return 1; }
    SatXAttnParams p{};
    p.q_rm = planes[0]; p.k_rm = planes[2]; p.v_rm = planes[4]; p.k_tr = planes[6]; p.q_tr = planes[8]; p.do_rm = planes[10]; p.do_tr = planes[12];
    if (!p.q_rm || !p.k_rm || !p.v_rm || !p.k_tr || !p.q_tr || !p.do_rm || !p.do_tr) { sat_set_error("sat_attention_cross_bwd: missing plane"); return 1; }
    p.lse = const_cast<float*>(lse); p.dsum = dsum; p.dq = dq; p.part = (float*)ws;
    p.B = B; p.H = H; p.Hkv = Hkv; p.Nq = Nq; p.Nk = Nk; p.Nqp = Nqp; p.Nkp = Nkp; p.scale = scale;
    p.nqb = sat_cdiv(Nq, 32);
    const int nkb = sat_cdiv(Nk, 32);
    const int nwb = (H / Hkv) * p.nqb;
    p.per_wg = sat_xatt_per_wg(B * Hkv, nwb);
    dim3 g1(sat_cdiv(nwb, p.per_wg), Hkv, B);
    SAT_XATT_DISPATCH(sat_attn_cross_dq_kernel, nkb, g1, SAT_XATT_B256, stream, p);
    int per;
    const int nsplit = sat_xatt_dkv_plan(B, H, Hkv, Nq, &per);
    p.per_wg = per;
    dim3 g2(nsplit, Hkv, B);
    SAT_XATT_DISPATCH(sat_attn_cross_dkv_kernel, nkb, g2, SAT_XATT_BNW, stream, p);
    SatXAttnReduceParams r{(const float*)ws, (short*)dk, (short*)dv, (long long)B * Hkv * Nk * SAT_ATT_D / 4, nsplit, scale};
    SAT_LAUNCH(sat_attn_cross_reduce_kernel, dim3((unsigned)sat_cdivll(2 * r.n4, 256)), dim3(256), stream, r);
    return sat_check_launch("sat_attention_cross_bwd");
}
