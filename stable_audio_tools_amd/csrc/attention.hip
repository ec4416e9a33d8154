// attention.hip — dense non-causal scaled-dot-product attention for the DiT (head dim 64), flash
// style on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16): forward, and backward (dQ / dK,dV),
// self- and grouped-query cross-attention.
//
// Replaces Attention.apply_attn (stable_audio_tools/models/transformer.py:406-441): flash_attn_func /
// F.scaled_dot_product_attention with scale 1/sqrt(d), no mask (masks are disabled by the caller,
// models/dit.py:283), the GQA repeat_interleave of k/v (transformer.py:408-411) done by indexing, and
// the autograd of all of it.
//
// Data flow
//   sat_attn_prepare   (b,h,n,64) strided fp32|bf16  ->  bf16 "planes": row-major [B][H][Np][64] and/or
//                      transposed [B][H][64][Np], Np = N rounded up to 64, zero padded.  fp32 sources
//                      are split into hi + lo planes (x ~ hi + lo to ~2^-17): every product in the
//                      kernels is then three MFMAs (hi*hi + hi*lo + lo*hi) with fp32 accumulation —
//                      the 1e-3 "rel fp32" parity mode.  bf16 sources give one plane, one MFMA.
//   kernels            every LDS tile is a straight 16-byte-per-lane copy of a plane tile into padded
//                      rows (conflict-free ds_read_b128 / b64); no conversion or transposition in the loop.
//
// Kernel structure (one wave = 32 rows of its "own" sequence, 4 waves per workgroup, 64-wide tiles of
// the other sequence staged through LDS): all products are arranged "swapped" so that the softmax row
// a lane works on is a COLUMN of the accumulator tile — its statistics are lane-local (one lane^32
// exchange), rescaling O is a per-lane scalar, and accumulator registers feed the next MFMA directly as
// its B operand: the MFMA k-slot order is defined to be the accumulator row order
// (slot (hi,e) of step u <-> row 16u + 4hi + (e&3) + 8(e>>2)) and the LDS-side operand is read in that
// same order.
#include "sat_device.h"

#define SAT_ATT_D 64
#define SAT_ATT_T 64              // tile width (keys in fwd/dQ, queries in dK/dV)
#define SAT_ATT_ROW 72            // bf16 per LDS row (64 + 8 pad -> 144 B stride, conflict-free b128/b64)

typedef short s4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// prepare: strided source -> bf16 planes
// ---------------------------------------------------------------------------------------------
struct SatPrepParams {
    const void* src;            // element (b,h,n,d) at b*sb + h*sh + n*sn + d
    long long sb, sh, sn;
    short* rm_hi; short* rm_lo; // [B][H][Np][64]   (null = skip)
    short* tr_hi; short* tr_lo; // [B][H][64][Np]
    int B, H, N, Np;
};

template <typename T> struct SatSrc;
template <> struct SatSrc<float> { static SAT_DEVICE float at(const void* p, long long i) { return ((const float*)p)[i]; } };
template <> struct SatSrc<short> { static SAT_DEVICE float at(const void* p, long long i) { return sat_bf16_to_f32(((const short*)p)[i]); } };

// One 64 (n) x 64 (d) tile per workgroup; a thread converts 8 consecutive d of one row (16-byte plane stores), the
// transposed planes go through an LDS tile and are written 8 consecutive n at a time.
template <typename T>
__global__ void __launch_bounds__(256) sat_attn_prepare_kernel(SatPrepParams p) {
    __shared__ __attribute__((aligned(16))) short t_hi[SAT_ATT_T][SAT_ATT_D + 8];   // [n][d]
    __shared__ __attribute__((aligned(16))) short t_lo[SAT_ATT_T][SAT_ATT_D + 8];
    const int n0 = blockIdx.x * SAT_ATT_T, h = blockIdx.y, b = blockIdx.z;
    const long long base = (long long)b * p.sb + (long long)h * p.sh;
    const size_t plane = ((size_t)b * p.H + h) * (size_t)p.Np * SAT_ATT_D;
    const bool want_tr = p.tr_hi != nullptr || p.tr_lo != nullptr;
    const bool vec_src = ((p.sn & 7) == 0) && ((p.sb & 7) == 0) && ((p.sh & 7) == 0) &&
                         (((uintptr_t)p.src & 15) == 0);                              // 16-byte loads allowed
    for (int i = threadIdx.x; i < SAT_ATT_T * SAT_ATT_D / 8; i += 256) {
        const int r = i >> 3, d0 = (i & 7) * 8;
        const int n = n0 + r;
        float x[8];
        if (n < p.N) {
            const long long off = base + (long long)n * p.sn + d0;
            if (vec_src) {
                if (sizeof(T) == 2) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>((const short*)p.src + off);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        x[2 * j] = __builtin_bit_cast(float, v[j] << 16);
                        x[2 * j + 1] = __builtin_bit_cast(float, v[j] & 0xffff0000u);
                    }
                } else {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>((const float*)p.src + off);
                    const f32x4 v1 = *reinterpret_cast<const f32x4*>((const float*)p.src + off + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { x[j] = v0[j]; x[4 + j] = v1[j]; }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = SatSrc<T>::at(p.src, off + j);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = 0.0f;
        }
        uint32_t wh[4], wl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sat_split2_pk(x[2 * j], x[2 * j + 1], &wh[j], &wl[j]);
        const u32x4 vh{wh[0], wh[1], wh[2], wh[3]}, vl{wl[0], wl[1], wl[2], wl[3]};
        const size_t o = plane + (size_t)n * SAT_ATT_D + d0;
        if (p.rm_hi) *reinterpret_cast<u32x4*>(p.rm_hi + o) = vh;
        if (p.rm_lo) *reinterpret_cast<u32x4*>(p.rm_lo + o) = vl;
        if (want_tr) {
            *reinterpret_cast<u32x4*>(&t_hi[r][d0]) = vh;
            *reinterpret_cast<u32x4*>(&t_lo[r][d0]) = vl;
        }
    }
    if (!want_tr) return;   // block-uniform
    __syncthreads();
    for (int i = threadIdx.x; i < SAT_ATT_T * SAT_ATT_D / 8; i += 256) {
        const int d = i >> 3, r0 = (i & 7) * 8;
        bf16x8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            vh[j] = t_hi[r0 + j][d];
            vl[j] = t_lo[r0 + j][d];
        }
        const size_t o = plane + (size_t)d * p.Np + n0 + r0;
        if (p.tr_hi) *reinterpret_cast<bf16x8*>(p.tr_hi + o) = vh;
        if (p.tr_lo) *reinterpret_cast<bf16x8*>(p.tr_lo + o) = vl;
    }
}

extern "C" int sat_attn_prepare(const void* src, long long sb, long long sh, long long sn, short* rm_hi, short* rm_lo,
                                short* tr_hi, short* tr_lo, int B, int H, int N, int Np, int dtype, void* stream) {
    if (B <= 0 || H <= 0 || N <= 0) { sat_set_error("sat_attn_prepare: empty shape"); return 1; }
    if (Np < N || (Np % SAT_ATT_T) != 0) { sat_set_error("sat_attn_prepare: Np must be N rounded up to a multiple of 64"); return 1; }
    if (dtype != 0 && dtype != 1) { sat_set_error("sat_attn_prepare: dtype must be 0 (f32) or 1 (bf16)"); return 1; }
    SatPrepParams p{src, sb, sh, sn, rm_hi, rm_lo, tr_hi, tr_lo, B, H, N, Np};
    dim3 grid(Np / SAT_ATT_T, H, B);
    if (dtype == 0) SAT_LAUNCH(sat_attn_prepare_kernel<float>, grid, dim3(256), stream, p);
    else SAT_LAUNCH(sat_attn_prepare_kernel<short>, grid, dim3(256), stream, p);
    return sat_check_launch("sat_attn_prepare");
}

// ---------------------------------------------------------------------------------------------
// shared device helpers
// ---------------------------------------------------------------------------------------------
// copy a ROWS x COLS bf16 tile (rows `rstride` elements apart in global) into padded LDS rows; 16 B per lane
template <int ROWS, int COLS, int LROW>
SAT_DEVICE void sat_att_stage(short (*dst)[LROW], const short* src, size_t rstride) {
    constexpr int PARTS = COLS / 8;
    for (int c = threadIdx.x; c < ROWS * PARTS; c += 256) {
        const int r = c / PARTS, part = c - r * PARTS;
        *reinterpret_cast<bf16x8*>(&dst[r][part * 8]) = *reinterpret_cast<const bf16x8*>(src + (size_t)r * rstride + part * 8);
    }
}
// A/B fragment with 8 CONSECUTIVE k (16 B): row-major tile, row = l31 (+32 sub), k offset = 16*s + 8*hi
template <int LROW>
SAT_DEVICE bf16x8 sat_att_frag_rm(short (*t)[LROW], int row, int koff) {
    return *reinterpret_cast<const bf16x8*>(&t[row][koff]);
}
// fragment in ACCUMULATOR-ROW order: elements e=0..3 at kofs+{0..3}, e=4..7 at kofs+8+{0..3}
template <int LROW>
SAT_DEVICE bf16x8 sat_att_frag_acc(short (*t)[LROW], int row, int kofs) {
    const s4v a = *reinterpret_cast<const s4v*>(&t[row][kofs]);
    const s4v b = *reinterpret_cast<const s4v*>(&t[row][kofs + 8]);
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = a[e];
        v[4 + e] = b[e];
    }
    return v;
}
template <int NP>
SAT_DEVICE void sat_att_pack(const f32x16& acc, int u, bf16x8 (&out)[NP]) {
#if !defined(SAT_HIPEMU)
    if (NP == 1) {   // one 8-wide convert: four v_cvt_pk_bf16_f32 straight into the operand's register quad
        typedef __bf16 sat_bf8 __attribute__((ext_vector_type(8)));
        typedef float sat_f8 __attribute__((ext_vector_type(8)));
        const sat_f8 v = {acc[8 * u], acc[8 * u + 1], acc[8 * u + 2], acc[8 * u + 3], acc[8 * u + 4], acc[8 * u + 5], acc[8 * u + 6], acc[8 * u + 7]};
        out[0] = __builtin_bit_cast(bf16x8, __builtin_convertvector(v, sat_bf8));
        return;
    }
#endif
    uint32_t wh[4], wl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                        // packed RNE converts
        if (NP == 2) sat_split2_pk(acc[8 * u + 2 * j], acc[8 * u + 2 * j + 1], &wh[j], &wl[j]);
        else wh[j] = sat_cvt2_pk(acc[8 * u + 2 * j], acc[8 * u + 2 * j + 1]);
    }
    out[0] = __builtin_bit_cast(bf16x8, u32x4{wh[0], wh[1], wh[2], wh[3]});
    if (NP == 2) out[NP - 1] = __builtin_bit_cast(bf16x8, u32x4{wl[0], wl[1], wl[2], wl[3]});
}
// 2^x on the transcendental unit (v_exp_f32); the arguments here are <= 0 and underflow to 0 as they should
SAT_DEVICE float sat_exp2(float x) {
#if defined(SAT_HIPEMU)
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);
#endif
}
// acc += A * B with the 1- or 3-MFMA product
template <int NP>
SAT_DEVICE f32x16 sat_att_mma(const bf16x8 (&a)[NP], const bf16x8 (&b)[NP], f32x16 acc) {
    acc = sat_mfma_32x32x16_bf16(a[0], b[0], acc);
    if (NP == 2) {
        acc = sat_mfma_32x32x16_bf16(a[0], b[NP - 1], acc);
        acc = sat_mfma_32x32x16_bf16(a[NP - 1], b[0], acc);
    }
    return acc;
}

struct SatAttnParams {
    // planes (hi, lo); lo pointers unused when NP == 1
    const short* q_rm[2];   // [B][H][Nqp][64]
    const short* k_rm[2];   // [B][Hkv][Nkp][64]
    const short* v_rm[2];   // [B][Hkv][Nkp][64]
    const short* k_tr[2];   // [B][Hkv][64][Nkp]
    const short* v_tr[2];   // [B][Hkv][64][Nkp]
    const short* q_tr[2];   // [B][H][64][Nqp]
    const short* do_rm[2];  // [B][H][Nqp][64]
    const short* do_tr[2];  // [B][H][64][Nqp]
    void* o;                // fwd out (B, Nq, H*64), dtype of the model
    float* lse;             // (B, H, Nq): fwd out / bwd in
    const float* dsum;      // (B, H, Nq) rowsum(dO * O)   (bwd)
    void* dq;               // bwd out (B, H, Nq, 64)
    void* dk;               // bwd out (B, Hkv, Nk, 64)
    void* dv;
    int B, H, Hkv, Nq, Nk, Nqp, Nkp;
    float scale;
};

template <typename T> struct SatOut;
template <> struct SatOut<float> { static SAT_DEVICE void put(void* p, long long i, float v) { ((float*)p)[i] = v; } };
template <> struct SatOut<short> { static SAT_DEVICE void put(void* p, long long i, float v) { ((short*)p)[i] = sat_f32_to_bf16(v); } };

// 16-byte load through a raw buffer descriptor: address = descriptor base + per-lane VGPR byte offset + scalar byte offset — the
// per-tile address arithmetic of the tile loops is one scalar add (buffer_load_dwordx4 v, voff, s[rsrc], soff offen) instead of one
// 64-bit per-lane add per piece.
#if defined(SAT_HIPEMU)
struct SatBuf { const char* base; };
SAT_DEVICE SatBuf sat_buf_make(const void* p) { return SatBuf{(const char*)p}; }
SAT_DEVICE bf16x8 sat_buf_load16(SatBuf b, unsigned voff, unsigned soff) { return *reinterpret_cast<const bf16x8*>(b.base + voff + soff); }
#else
typedef __amdgpu_buffer_rsrc_t SatBuf;
SAT_DEVICE SatBuf sat_buf_make(const void* p) {      // raw buffer: stride 0, 2 GiB - 1 bytes, gfx9 untyped dword3 (0x00020000)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
SAT_DEVICE bf16x8 sat_buf_load16(SatBuf b, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(b, voff, soff, 0));
}
#endif


// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
// K row read by lane l31 as MFMA row a = l31 of S^T = K Q^T: key(a) = a with bits 2 and 3 exchanged.  A free per-lane choice that makes
// accumulator registers [8u, 8u+8) of a lane 8 CONSECUTIVE keys (16u + 8 hi + e): the B-operand k-slots of the P V MFMA then line up
// with V^T in natural key order and the V^T fragments are single conflict-free 16-byte reads.
SAT_DEVICE int sat_att_kperm(int a) { return (a & 0x13) | ((a & 4) << 1) | ((a & 8) >> 1); }
// max over the two 32-lane halves, in every lane
SAT_DEVICE float sat_att_halfmax(float x) {
#if defined(SAT_HIPEMU)
    return fmaxf(x, __shfl_xor(x, 32));
#else
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
#endif
}
// Round 4 — the forward kernel's instruction diet (profiles/EXPERIMENTS.md: the SQ counters put the kernel's time in the SIMD's issue
// slots, not in one pipe; what was measured to help is listed here, what was measured NOT to help is in EXPERIMENTS.md):
//   * Q is pre-scaled by scale * log2(e) once per launch (bf16 mode: one more bf16 rounding of Q; fp32 mode: hi + lo are recombined,
//     scaled and re-split, so the product stays at 2^-17), so the scores leave the matrix pipe in the exp2 domain;
//   * the running row max mb (exp2 domain) enters through the C operand of the first QK^T MFMA (a 16-register block holding -mb): the
//     accumulators ARE x = s - mb, no per-score multiply-subtract;
//   * no per-tile row max: after the first tile (true max) the max only moves when the tile's row sum says it went stale —
//     the sum of exp2(x) over this lane's scores above SAT_ATT_SUMLIM (32 scores of 2^SAT_ATT_DEFER each), or inf / NaN — and that rare
//     path recomputes QK^T (the scores were overwritten by their exponentials), takes the true max, rescales O / l and goes on.  P is
//     therefore bounded by SAT_ATT_SUMLIM instead of 1 — harmless in fp32 / bf16;
//   * s_setprio 1 around the MFMA groups, and the file is built with -fno-slp-vectorize (v_pk_*_f32 cost more issue time than the
//     scalar pairs they replace; Makefile).
// Measured (tools/attn_x_bench.py, profiles/r04_experiments/): N = 1025, B*H = 48: 31.7 -> 24.0 us; N = 6145: 568 -> 492 us.
#define SAT_ATT_DEFER 4.0f
#define SAT_ATT_SUMLIM 512.0f      // 32 scores per lane, each <= 2^SAT_ATT_DEFER while the running max is in range

// one 64-key tile (NKB = 2) or its first 32 keys (NKB = 1): x = K (Q c)^T - mb, P = exp2(x), O^T += V^T P^T.
// FIRST: the wave's first tile — mb is not known yet: C = 0 and the true row max is taken.
// acc + the two bf16 of a packed word (v_dot2c_f32_bf16 against (1, 1)): the row sum of P taken from the ROUNDED probabilities — the values
// the P V product multiplies — at one instruction per two scores
SAT_DEVICE float sat_att_sum2(uint32_t w, float acc) {
#if defined(SAT_HIPEMU)
    return acc + sat_bf16_to_f32((short)(w & 0xffffu)) + sat_bf16_to_f32((short)(w >> 16));
#else
    typedef __bf16 sat_b2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sat_b2, w), __builtin_bit_cast(sat_b2, 0x3f803f80u), acc, false);
#endif
}

// DOT2 (bf16 planes: NP == 1; round 5 — timed in profiles/r05_experiments/lean_ab/): P is packed to bf16 BEFORE the row sum, which then
// runs on sat_att_sum2 over the packed words — 16 instructions per 64-key tile instead of 32 adds; the normaliser is the sum of the
// ROUNDED probabilities (what P V multiplies): the output is normalised consistently, the LSE moves by <= 2^-9 / sqrt(keys) relative.
template <int NP, int NKB, bool MASK, bool FIRST, bool DOT2 = false>
SAT_DEVICE void sat_attn_fwd_tile(short (*k_lds)[SAT_ATT_T][SAT_ATT_ROW], short (*v_lds)[SAT_ATT_D][SAT_ATT_ROW], const bf16x8 (&qf)[4][NP],
                                  f32x16 (&oacc)[2], f32x16& negm, float& mb, float& l_run, int l31, int hi, int kperm, int nvalid) {
    f32x16 sacc[NKB];
    auto qk = [&]() {
        SAT_SETPRIO(1);
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (FIRST) {
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.0f;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 ka[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) ka[pl] = sat_att_frag_rm(k_lds[pl], kb * 32 + kperm, 16 * s + 8 * hi);
                // the first MFMA of the chain reads -mb straight from the negm block (no copy into the accumulator first)
                sacc[kb] = sat_att_mma<NP>(ka, qf[s], (!FIRST && s == 0) ? negm : sacc[kb]);
            }
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
                if (!MASK || key < nvalid) tmax = fmaxf(tmax, sacc[kb][r]);      // only the last tile holds padded keys (block-uniform)
            }
        return sat_att_halfmax(tmax);
    };
    static_assert(!DOT2 || NP == 1, "the packed row sum needs single-plane (bf16) probabilities");
    float ps0 = 0.0f, ps1 = 0.0f;
    bf16x8 pbs[DOT2 ? NKB : 1][2];      // DOT2: the tile's packed probabilities (the B operands of the P V MFMAs)
    auto expsum = [&]() {
        ps0 = 0.0f;
        ps1 = 0.0f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a = sat_exp2(sacc[kb][2 * j]), c = sat_exp2(sacc[kb][2 * j + 1]);
                if (MASK) {
                    const int key = kb * 32 + ((2 * j) & 7) + 8 * hi + 16 * ((2 * j) >> 3);
                    if (key >= nvalid) a = 0.0f;
                    if (key + 1 >= nvalid) c = 0.0f;
                }
                if (!DOT2) {
                    ps0 += a;
                    ps1 += c;
                }
                sacc[kb][2 * j] = a;
                sacc[kb][2 * j + 1] = c;
            }
        if (DOT2) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    bf16x8 one[NP];
                    sat_att_pack<NP>(sacc[kb], u, one);
                    pbs[DOT2 ? kb : 0][u] = one[0];
                    const u32x4 w = __builtin_bit_cast(u32x4, one[0]);
                    ps0 = sat_att_sum2(w[0], ps0);
                    ps1 = sat_att_sum2(w[1], ps1);
                    ps0 = sat_att_sum2(w[2], ps0);
                    ps1 = sat_att_sum2(w[3], ps1);
                }
        }
    };
    qk();
    if (FIRST) {
        const float tmax = rowmax();
        mb = tmax;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] -= tmax;
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -mb;
    }
    expsum();
    if (!FIRST) {
        if (sat_wave_any(!(ps0 + ps1 <= SAT_ATT_SUMLIM))) {      // stale running max in some row of the wave (rare): redo with the true one
            qk();
            const float d = fmaxf(rowmax(), 0.0f), alpha = sat_exp2(-d);
            mb += d;
            l_run *= alpha;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[kb][r] -= d;
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[r] = -mb;
            expsum();
        }
    }
    l_run += ps0 + ps1;
    SAT_SETPRIO(1);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf16x8 pb[NP];
            if (DOT2) pb[0] = pbs[DOT2 ? kb : 0][u];
            else sat_att_pack<NP>(sacc[kb], u, pb);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bf16x8 va[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) va[pl] = sat_att_frag_rm(v_lds[pl], t * 32 + l31, kb * 32 + 16 * u + 8 * hi);
                oacc[t] = sat_att_mma<NP>(va, pb, oacc[t]);
            }
        }
    SAT_SETPRIO(0);
}

template <typename T, int NP>
__global__ void __launch_bounds__(256)
#if !defined(SAT_HIPEMU)
__attribute__((amdgpu_waves_per_eu(3)))       // >= 3 workgroups per CU so that softmax VALU of one overlaps MFMAs of another
#endif
sat_attn_fwd_kernel(SatAttnParams p) {
    // K / V^T tiles double-buffered in LDS; tile k+1 travels through registers while tile k is consumed: one barrier per tile
    __shared__ __attribute__((aligned(16))) short k_lds2[2][NP][SAT_ATT_T][SAT_ATT_ROW];   // [buffer][plane][key][d]
    __shared__ __attribute__((aligned(16))) short v_lds2[2][NP][SAT_ATT_D][SAT_ATT_ROW];   // [buffer][plane][d][key]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int kperm = sat_att_kperm(l31);
    const int b = blockIdx.z, h = blockIdx.y;
    const int hk = h / (p.H / p.Hkv);
    const int qrow = blockIdx.x * 128 + wave * 32 + l31;     // < Nqp (grid covers Nqp in steps of 128 only when present)
    const bool q_in = qrow < p.Nqp;
    const bool q_ok = qrow < p.Nq;
    const bool w_ok = blockIdx.x * 128 + wave * 32 < p.Nq;   // wave-uniform: this wave owns a valid query
    const size_t qplane = ((size_t)b * p.H + h) * (size_t)p.Nqp * SAT_ATT_D;
    const size_t kplane = ((size_t)b * p.Hkv + hk) * (size_t)p.Nkp * SAT_ATT_D;
    const float sl2 = p.scale * 1.4426950408889634f;

    bf16x8 qf[4][NP];   // B operand of x = K (Q c)^T: lane (q = l31, hi) holds d = 16 s + 8 hi + e, pre-scaled by c = scale * log2(e)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        u32x4 w[NP];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            if (q_in) w[pl] = *reinterpret_cast<const u32x4*>(p.q_rm[pl] + qplane + (size_t)qrow * SAT_ATT_D + 16 * s + 8 * hi);
            else w[pl] = u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {       // two bf16 per word: element 2j in the low half
            float e0 = __builtin_bit_cast(float, w[0][j] << 16), e1 = __builtin_bit_cast(float, w[0][j] & 0xffff0000u);
            if (NP == 2) {
                e0 += __builtin_bit_cast(float, w[NP - 1][j] << 16);
                e1 += __builtin_bit_cast(float, w[NP - 1][j] & 0xffff0000u);
                uint32_t wh, wl;
                sat_split2_pk(e0 * sl2, e1 * sl2, &wh, &wl);
                w[0][j] = wh;
                w[NP - 1][j] = wl;
            } else {
                w[0][j] = sat_cvt2_pk(e0 * sl2, e1 * sl2);
            }
        }
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) qf[s][pl] = __builtin_bit_cast(bf16x8, w[pl]);
    }

    f32x16 oacc[2], negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        oacc[0][r] = 0.0f;
        oacc[1][r] = 0.0f;
        negm[r] = 0.0f;
    }
    float mb = 0.0f, l_run = 0.0f;      // running row max (exp2 domain) and row sum

    // a 64 x 64 bf16 tile = 512 16-byte pieces = 2 per thread and plane: LDS (row, part) and the byte offsets from the tile's uniform base
    int srow[2], spart[2];
    unsigned kob[2], vob[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = threadIdx.x + j * 256;
        srow[j] = c >> 3;
        spart[j] = c & 7;
        kob[j] = (unsigned)(srow[j] * SAT_ATT_D + spart[j] * 8) * 2u;
        vob[j] = ((unsigned)srow[j] * (unsigned)p.Nkp + (unsigned)spart[j] * 8u) * 2u;      // < 64 * Nkp * 2 bytes: Nkp < 2^24 (sat_attn_check)
    }
    // this (batch item, kv head)'s K and V^T planes: descriptors in scalar registers (the lo planes' only exist in the two-plane mode)
    const SatBuf kbuf0 = sat_buf_make(p.k_rm[0] + kplane), vbuf0 = sat_buf_make(p.v_tr[0] + kplane);
    const SatBuf kbuf1 = sat_buf_make(p.k_rm[NP - 1] + kplane), vbuf1 = sat_buf_make(p.v_tr[NP - 1] + kplane);
    bf16x8 kreg[NP][2], vreg[NP][2];
    auto tile_load = [&](int k0) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                kreg[pl][j] = sat_buf_load16(pl == 0 ? kbuf0 : kbuf1, kob[j], (unsigned)k0 * (SAT_ATT_D * 2));
                vreg[pl][j] = sat_buf_load16(pl == 0 ? vbuf0 : vbuf1, vob[j], (unsigned)k0 * 2u);
            }
    };
    auto tile_store = [&](int buf) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                *reinterpret_cast<bf16x8*>(&k_lds2[buf][pl][srow[j]][spart[j] * 8]) = kreg[pl][j];
                *reinterpret_cast<bf16x8*>(&v_lds2[buf][pl][srow[j]][spart[j] * 8]) = vreg[pl][j];
            }
    };
    tile_load(0);
    tile_store(0);
    if (SAT_ATT_T < p.Nk) tile_load(SAT_ATT_T);
    __syncthreads();
    int buf = 0, k0 = 0;
    if (p.Nk >= SAT_ATT_T) {      // the first full tile, peeled: it establishes the running max
        if (SAT_ATT_T < p.Nk) {
            tile_store(1);
            if (2 * SAT_ATT_T < p.Nk) tile_load(2 * SAT_ATT_T);
        }
        if (w_ok) sat_attn_fwd_tile<NP, 2, false, true, NP == 1>(k_lds2[0], v_lds2[0], qf, oacc, negm, mb, l_run, l31, hi, kperm, SAT_ATT_T);
        __syncthreads();
        k0 = SAT_ATT_T;
        buf = 1;
    }
    // full tiles: ONE straight-line body, so that the accumulators keep their registers around the loop; the ragged last tile is peeled
    for (; k0 + SAT_ATT_T <= p.Nk; k0 += SAT_ATT_T, buf ^= 1) {
        if (k0 + SAT_ATT_T < p.Nk) {
            tile_store(buf ^ 1);                                         // tile k+1: registers -> the other buffer
            if (k0 + 2 * SAT_ATT_T < p.Nk) tile_load(k0 + 2 * SAT_ATT_T);   // tile k+2 -> registers (lands during this tile's math)
        }
        if (w_ok) sat_attn_fwd_tile<NP, 2, false, false, NP == 1>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, SAT_ATT_T);
        __syncthreads();
    }
    if (k0 < p.Nk && w_ok) {
        const int rem = p.Nk - k0;
        if (k0 == 0) {            // fewer than 64 keys in all: the ragged tile is also the first
            if (rem > 32) sat_attn_fwd_tile<NP, 2, true, true, NP == 1>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
            else sat_attn_fwd_tile<NP, 1, true, true, NP == 1>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
        } else if (rem > 32) {
            sat_attn_fwd_tile<NP, 2, true, false, NP == 1>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
        } else {
            sat_attn_fwd_tile<NP, 1, true, false, NP == 1>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem);
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv_l = 1.0f / l_tot;
    if (q_ok) {
        // lane (q, hi) holds d = 32 t + 8 g + 4 hi + {0..3} in registers 4 g + {0..3}: 8-byte (bf16) / 16-byte (fp32) stores
        const long long obase = ((long long)b * p.Nq + qrow) * ((long long)p.H * SAT_ATT_D) + (long long)h * SAT_ATT_D;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {oacc[t][4 * g] * inv_l, oacc[t][4 * g + 1] * inv_l, oacc[t][4 * g + 2] * inv_l, oacc[t][4 * g + 3] * inv_l};
                const long long idx = obase + t * 32 + 8 * g + 4 * hi;
                if (sizeof(T) == 4) *(f32x4*)((float*)p.o + idx) = v;
                else *(u32x2*)((short*)p.o + idx) = u32x2{sat_cvt2_pk(v[0], v[1]), sat_cvt2_pk(v[2], v[3])};
            }
        // natural-log LSE of the scaled scores: (mb + log2 l) ln 2   (mb lives in the exp2 domain)
        if (p.lse && hi == 0) p.lse[((long long)b * p.H + h) * p.Nq + qrow] = (mb + log2f(l_tot)) * 0.6931471805599453f;
    }
}

// ---------------------------------------------------------------------------------------------
// forward, 64 queries per wave (round 6; bf16 planes only).  The 32-query kernel above is ISSUE-bound (profiles/r04_pmc_attn_summary.txt:
// ~135 issued instructions per 16 MFMAs); a wave that owns TWO 32-query blocks reads every K / V^T fragment once for both (the 16
// ds_read_b128 per 64-key tile, the staging loads / stores and the scalar work are shared: ~185 issue slots per 32 MFMAs instead of
// 270), at two waves per SIMD (~210 registers) instead of three.  The tile is walked one 32-KEY block at a time — QK^T (4 MFMAs per
// query block), exp2 / pack / row sum, P V (4 MFMAs per query block) — so only 2 x 16 score registers are live; the stale-max check
// (the deferred rescale of the kernel above) is per key block: 16 scores per lane, limit 16 x 2^SAT_ATT_DEFER.
// Used for long sequences whose grid still fills the chip (sat_attention_fwd: Nq >= 2048 and >= 2 workgroups of 256 queries per CU);
// measured gain: 2 % at N = 6145, none at N = 1025 — the kernel is bound by the softmax's VALU issue per score, which sharing
// fragments does not touch (the experiment's answer to round 5's hypothesis).
// ---------------------------------------------------------------------------------------------
#define SAT_ATT_SUMLIM_KB 256.0f

template <int NQB, int KB, bool MASK, bool FIRST>
SAT_DEVICE void sat_attn_fwd_kb(short (*k_lds)[SAT_ATT_ROW], short (*v_lds)[SAT_ATT_ROW], const bf16x8 (&qf)[NQB][4], f32x16 (&oacc)[NQB][2],
                                f32x16 (&negm)[NQB], float (&mb)[NQB], float (&l_run)[NQB], int l31, int hi, int kperm, int nvalid) {
    f32x16 sacc[NQB];
    const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    auto qk = [&]() {
        SAT_SETPRIO(1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 ka = sat_att_frag_rm(k_lds, KB * 32 + kperm, 16 * s + 8 * hi);      // ONE fragment read for both query blocks
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) sacc[qb] = sat_mfma_32x32x16_bf16(ka, qf[qb][s], s == 0 ? (FIRST ? zero : negm[qb]) : sacc[qb]);
        }
        SAT_SETPRIO(0);
    };
    auto rowmax = [&](int qb) {
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 7) + 8 * hi + 16 * (r >> 3);
            if (!MASK || key < nvalid) tmax = fmaxf(tmax, sacc[qb][r]);
        }
        return sat_att_halfmax(tmax);
    };
    bf16x8 pbs[NQB][2];
    float ps[NQB];
    auto expsum = [&](int qb) {
        float ps0 = 0.0f, ps1 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a = sat_exp2(sacc[qb][r]);
            if (MASK) {
                if ((r & 7) + 8 * hi + 16 * (r >> 3) >= nvalid) a = 0.0f;
            }
            sacc[qb][r] = a;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf16x8 one[1];
            sat_att_pack<1>(sacc[qb], u, one);
            pbs[qb][u] = one[0];
            const u32x4 w = __builtin_bit_cast(u32x4, one[0]);
            ps0 = sat_att_sum2(w[0], ps0);
            ps1 = sat_att_sum2(w[1], ps1);
            ps0 = sat_att_sum2(w[2], ps0);
            ps1 = sat_att_sum2(w[3], ps1);
        }
        ps[qb] = ps0 + ps1;
    };
    qk();
    if (FIRST) {
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            const float tmax = rowmax(qb);
            mb[qb] = tmax;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[qb][r] -= tmax;
                negm[qb][r] = -tmax;
            }
        }
    }
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) expsum(qb);
    if (!FIRST) {
        bool stale = false;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) stale = stale || !(ps[qb] <= SAT_ATT_SUMLIM_KB);
        if (sat_wave_any(stale)) {      // a row of the wave outgrew its running max (rare): redo this key block with the true one
            qk();
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                const float d = fmaxf(rowmax(qb), 0.0f), alpha = sat_exp2(-d);
                mb[qb] += d;
                l_run[qb] *= alpha;
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[qb][t][r] *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sacc[qb][r] -= d;
                    negm[qb][r] = -mb[qb];
                }
                expsum(qb);
            }
        }
    }
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) l_run[qb] += ps[qb];
    SAT_SETPRIO(1);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 va = sat_att_frag_rm(v_lds, t * 32 + l31, KB * 32 + 16 * u + 8 * hi);
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) oacc[qb][t] = sat_mfma_32x32x16_bf16(va, pbs[qb][u], oacc[qb][t]);
        }
    SAT_SETPRIO(0);
}

template <int NQB>
__global__ void __launch_bounds__(256)
#if !defined(SAT_HIPEMU)
__attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
sat_attn_fwd_q_kernel(SatAttnParams p) {
    __shared__ __attribute__((aligned(16))) short k_lds2[2][SAT_ATT_T][SAT_ATT_ROW];   // [buffer][key][d]
    __shared__ __attribute__((aligned(16))) short v_lds2[2][SAT_ATT_D][SAT_ATT_ROW];   // [buffer][d][key]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int kperm = sat_att_kperm(l31);
    const int b = blockIdx.z, h = blockIdx.y;
    const int hk = h / (p.H / p.Hkv);
    const int qrow0 = blockIdx.x * (128 * NQB) + wave * (32 * NQB);      // the wave's first query; its blocks: qrow0 + 32 qb + l31
    const bool w_ok = qrow0 < p.Nq;                                      // wave-uniform: this wave owns a valid query
    const size_t qplane = ((size_t)b * p.H + h) * (size_t)p.Nqp * SAT_ATT_D;
    const size_t kplane = ((size_t)b * p.Hkv + hk) * (size_t)p.Nkp * SAT_ATT_D;
    const float sl2 = p.scale * 1.4426950408889634f;

    bf16x8 qf[NQB][4];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const int qrow = qrow0 + qb * 32 + l31;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 w = u32x4{0u, 0u, 0u, 0u};
            if (qrow < p.Nqp) w = *reinterpret_cast<const u32x4*>(p.q_rm[0] + qplane + (size_t)qrow * SAT_ATT_D + 16 * s + 8 * hi);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e0 = __builtin_bit_cast(float, w[j] << 16), e1 = __builtin_bit_cast(float, w[j] & 0xffff0000u);
                w[j] = sat_cvt2_pk(e0 * sl2, e1 * sl2);
            }
            qf[qb][s] = __builtin_bit_cast(bf16x8, w);
        }
    }
    f32x16 oacc[NQB][2], negm[NQB];
    float mb[NQB], l_run[NQB];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        mb[qb] = 0.0f;
        l_run[qb] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            oacc[qb][0][r] = 0.0f;
            oacc[qb][1][r] = 0.0f;
            negm[qb][r] = 0.0f;
        }
    }

    int srow[2], spart[2];
    unsigned kob[2], vob[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = threadIdx.x + j * 256;
        srow[j] = c >> 3;
        spart[j] = c & 7;
        kob[j] = (unsigned)(srow[j] * SAT_ATT_D + spart[j] * 8) * 2u;
        vob[j] = ((unsigned)srow[j] * (unsigned)p.Nkp + (unsigned)spart[j] * 8u) * 2u;
    }
    const SatBuf kbuf = sat_buf_make(p.k_rm[0] + kplane), vbuf = sat_buf_make(p.v_tr[0] + kplane);
    bf16x8 kreg[2], vreg[2];
    auto tile_load = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            kreg[j] = sat_buf_load16(kbuf, kob[j], (unsigned)k0 * (SAT_ATT_D * 2));
            vreg[j] = sat_buf_load16(vbuf, vob[j], (unsigned)k0 * 2u);
        }
    };
    auto tile_store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<bf16x8*>(&k_lds2[buf][srow[j]][spart[j] * 8]) = kreg[j];
            *reinterpret_cast<bf16x8*>(&v_lds2[buf][srow[j]][spart[j] * 8]) = vreg[j];
        }
    };
    tile_load(0);
    tile_store(0);
    if (SAT_ATT_T < p.Nk) tile_load(SAT_ATT_T);
    __syncthreads();
    int buf = 0, k0 = 0;
    if (p.Nk >= SAT_ATT_T) {      // the first full tile, peeled: its first key block establishes the running max
        if (SAT_ATT_T < p.Nk) {
            tile_store(1);
            if (2 * SAT_ATT_T < p.Nk) tile_load(2 * SAT_ATT_T);
        }
        if (w_ok) {
            sat_attn_fwd_kb<NQB, 0, false, true>(k_lds2[0], v_lds2[0], qf, oacc, negm, mb, l_run, l31, hi, kperm, 32);
            sat_attn_fwd_kb<NQB, 1, false, false>(k_lds2[0], v_lds2[0], qf, oacc, negm, mb, l_run, l31, hi, kperm, 32);
        }
        __syncthreads();
        k0 = SAT_ATT_T;
        buf = 1;
    }
    for (; k0 + SAT_ATT_T <= p.Nk; k0 += SAT_ATT_T, buf ^= 1) {
        if (k0 + SAT_ATT_T < p.Nk) {
            tile_store(buf ^ 1);
            if (k0 + 2 * SAT_ATT_T < p.Nk) tile_load(k0 + 2 * SAT_ATT_T);
        }
        if (w_ok) {
            sat_attn_fwd_kb<NQB, 0, false, false>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, 32);
            sat_attn_fwd_kb<NQB, 1, false, false>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, 32);
        }
        __syncthreads();
    }
    if (k0 < p.Nk && w_ok) {      // the ragged last tile (also the first when Nk < 64)
        const int rem = p.Nk - k0;
        const int n0 = rem < 32 ? rem : 32;
        if (k0 == 0) sat_attn_fwd_kb<NQB, 0, true, true>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, n0);
        else sat_attn_fwd_kb<NQB, 0, true, false>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, n0);
        if (rem > 32) sat_attn_fwd_kb<NQB, 1, true, false>(k_lds2[buf], v_lds2[buf], qf, oacc, negm, mb, l_run, l31, hi, kperm, rem - 32);
    }
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
        const int qrow = qrow0 + qb * 32 + l31;
        const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32);
        const float inv_l = 1.0f / l_tot;
        if (qrow < p.Nq) {
            const long long obase = ((long long)b * p.Nq + qrow) * ((long long)p.H * SAT_ATT_D) + (long long)h * SAT_ATT_D;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const long long idx = obase + t * 32 + 8 * g + 4 * hi;
                    *(u32x2*)((short*)p.o + idx) = u32x2{sat_cvt2_pk(oacc[qb][t][4 * g] * inv_l, oacc[qb][t][4 * g + 1] * inv_l),
                                                         sat_cvt2_pk(oacc[qb][t][4 * g + 2] * inv_l, oacc[qb][t][4 * g + 3] * inv_l)};
                }
            if (p.lse && hi == 0) p.lse[((long long)b * p.H + h) * p.Nq + qrow] = (mb[qb] + log2f(l_tot)) * 0.6931471805599453f;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward, part 1: dQ.  Same roles as the forward (wave owns 32 queries, loop over key tiles).
//   S^T = K Q^T ; P^T = exp(S^T*scale - lse) ; dP^T = V dO^T ; dS^T = P^T (dP^T - D) scale ; dQ^T += K^T dS^T
// ---------------------------------------------------------------------------------------------
template <typename T, int NP>
__global__ void __launch_bounds__(256) sat_attn_bwd_dq_kernel(SatAttnParams p) {
    // K, V (row-major) and K^T tiles double-buffered; tile k+1 travels through registers while tile k is consumed
    __shared__ __attribute__((aligned(16))) short k_lds2[2][NP][SAT_ATT_T][SAT_ATT_ROW];    // [buffer][plane][key][d]
    __shared__ __attribute__((aligned(16))) short v_lds2[2][NP][SAT_ATT_T][SAT_ATT_ROW];    // [key][d]
    __shared__ __attribute__((aligned(16))) short kt_lds2[2][NP][SAT_ATT_D][SAT_ATT_ROW];   // [d][key]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int hk = h / (p.H / p.Hkv);
    const int qrow = blockIdx.x * 128 + wave * 32 + l31;
    const bool q_in = qrow < p.Nqp, q_ok = qrow < p.Nq;
    const size_t qplane = ((size_t)b * p.H + h) * (size_t)p.Nqp * SAT_ATT_D;
    const size_t kplane = ((size_t)b * p.Hkv + hk) * (size_t)p.Nkp * SAT_ATT_D;

    bf16x8 qf[4][NP], gf[4][NP];   // Q and dO fragments (d = 16 s + 8 hi + e)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            if (q_in) {
                qf[s][pl] = *reinterpret_cast<const bf16x8*>(p.q_rm[pl] + qplane + (size_t)qrow * SAT_ATT_D + 16 * s + 8 * hi);
                gf[s][pl] = *reinterpret_cast<const bf16x8*>(p.do_rm[pl] + qplane + (size_t)qrow * SAT_ATT_D + 16 * s + 8 * hi);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    qf[s][pl][e] = 0;
                    gf[s][pl][e] = 0;
                }
            }
        }
    const float l2e = 1.4426950408889634f;
    const float sl2 = p.scale * l2e;
    const float lse2 = q_ok ? p.lse[((long long)b * p.H + h) * p.Nq + qrow] * l2e : 0.0f;
    const float dsum = q_ok ? p.dsum[((long long)b * p.H + h) * p.Nq + qrow] : 0.0f;

    f32x16 dq[2];   // dQ^T: rows d, cols q
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[t][r] = 0.0f;

    bf16x8 rk[NP][2], rv[NP][2], rkt[NP][2];
    auto tile_load = [&](int k0) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = threadIdx.x + j * 256, r = c >> 3, part = c & 7;
                rk[pl][j] = *reinterpret_cast<const bf16x8*>(p.k_rm[pl] + kplane + (size_t)(k0 + r) * SAT_ATT_D + part * 8);
                rv[pl][j] = *reinterpret_cast<const bf16x8*>(p.v_rm[pl] + kplane + (size_t)(k0 + r) * SAT_ATT_D + part * 8);
                rkt[pl][j] = *reinterpret_cast<const bf16x8*>(p.k_tr[pl] + kplane + (size_t)r * p.Nkp + k0 + part * 8);
            }
    };
    auto tile_store = [&](int buf) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = threadIdx.x + j * 256, r = c >> 3, part = c & 7;
                *reinterpret_cast<bf16x8*>(&k_lds2[buf][pl][r][part * 8]) = rk[pl][j];
                *reinterpret_cast<bf16x8*>(&v_lds2[buf][pl][r][part * 8]) = rv[pl][j];
                *reinterpret_cast<bf16x8*>(&kt_lds2[buf][pl][r][part * 8]) = rkt[pl][j];
            }
    };
    tile_load(0);
    tile_store(0);
    if (SAT_ATT_T < p.Nk) tile_load(SAT_ATT_T);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < p.Nk; k0 += SAT_ATT_T, buf ^= 1) {
        short (*k_lds)[SAT_ATT_T][SAT_ATT_ROW] = k_lds2[buf];
        short (*v_lds)[SAT_ATT_T][SAT_ATT_ROW] = v_lds2[buf];
        short (*kt_lds)[SAT_ATT_D][SAT_ATT_ROW] = kt_lds2[buf];
        if (k0 + SAT_ATT_T < p.Nk) {
            tile_store(buf ^ 1);
            if (k0 + 2 * SAT_ATT_T < p.Nk) tile_load(k0 + 2 * SAT_ATT_T);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 sacc, pacc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sacc[r] = 0.0f;
                pacc[r] = 0.0f;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 ka[NP], va[NP];
#pragma unroll
                for (int pl = 0; pl < NP; ++pl) {
                    ka[pl] = sat_att_frag_rm(k_lds[pl], kb * 32 + l31, 16 * s + 8 * hi);
                    va[pl] = sat_att_frag_rm(v_lds[pl], kb * 32 + l31, 16 * s + 8 * hi);
                }
                sacc = sat_att_mma<NP>(ka, qf[s], sacc);
                pacc = sat_att_mma<NP>(va, gf[s], pacc);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float pv = (key < p.Nk && q_ok) ? sat_exp2(fmaf(sacc[r], sl2, -lse2)) : 0.0f;
                sacc[r] = pv * (pacc[r] - dsum) * p.scale;   // dS^T
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bf16x8 sb[NP];
                sat_att_pack<NP>(sacc, u, sb);
                const int kofs = kb * 32 + 16 * u + 4 * hi;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    bf16x8 ka[NP];
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) ka[pl] = sat_att_frag_acc(kt_lds[pl], t * 32 + l31, kofs);
                    dq[t] = sat_att_mma<NP>(ka, sb, dq[t]);
                }
            }
        }
        __syncthreads();
    }
    if (q_ok) {
        const long long obase = (((long long)b * p.H + h) * p.Nq + qrow) * SAT_ATT_D;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) SatOut<T>::put(p.dq, obase + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, dq[t][r]);
    }
}

// ---------------------------------------------------------------------------------------------
// The bf16 dQ kernel (round 5: round 4's "lean" arm is THE bf16 kernel — profiles/r05_experiments/lean_ab/: dQ + dK/dV per launch
// 272 -> 195 us (self, N = 1025, B = 4), 180 -> 124 us (cross); the two-plane kernel above serves the fp32 mode only).  The general kernel
// issues 605 instructions per 64-key tile and wave for 24 MFMAs: 162 v_accvgpr_read / write (the register allocator parks values in
// AGPRs: no occupancy attribute, 142 VGPRs), 96 for a per-element (key < Nk && q_ok) mask on EVERY tile, 160 of softmax / dS arithmetic
// (fma, exp, sub, 2 mul per score), 63 of address arithmetic.  Here:
//   * amdgpu_waves_per_eu(2, 2): the 55 KB of LDS allow two workgroups per CU anyway — 256 registers, nothing parked in AGPRs;
//   * no mask on full tiles: a column of dS^T only feeds the same query's column of dQ^T (never stored for q >= Nq), and keys >= Nk
//     multiply zero columns of K^T; the ragged last tile is peeled and masked;
//   * Q pre-scaled by scale * log2(e) (as the forward: the probabilities are recomputed from the operands that produced the LSE), -lse
//     (exp2 domain) and -D enter through the C operands of the S^T and dP^T MFMA chains: per score exp2, one multiply, half a convert;
//     the factor `scale` of dS is applied to dQ once at the end;
//   * K / V / K^T tile loads through buffer descriptors.
// ---------------------------------------------------------------------------------------------
template <bool MASK>
SAT_DEVICE void sat_attn_dq_bf16_tile(short (*k_lds)[SAT_ATT_ROW], short (*v_lds)[SAT_ATT_ROW], short (*kt_lds)[SAT_ATT_ROW],
                                      const bf16x8 (&qf)[4], const bf16x8 (&gf)[4], const f32x16& negl, const f32x16& negd,
                                      f32x16 (&dq)[2], int l31, int hi, int nvalid) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        f32x16 sacc, pacc;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 ka = sat_att_frag_rm(k_lds, kb * 32 + l31, 16 * s + 8 * hi);
            const bf16x8 va = sat_att_frag_rm(v_lds, kb * 32 + l31, 16 * s + 8 * hi);
            sacc = sat_mfma_32x32x16_bf16(ka, qf[s], s == 0 ? negl : sacc);      // x = K (Q c)^T - lse (exp2 domain)
            pacc = sat_mfma_32x32x16_bf16(va, gf[s], s == 0 ? negd : pacc);      // dP^T - D
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float pv = sat_exp2(sacc[r]);
            if (MASK) {
                const int key = kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (key >= nvalid) pv = 0.0f;
            }
            sacc[r] = pv * pacc[r];                                               // dS^T / scale
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf16x8 sb[1];
            sat_att_pack<1>(sacc, u, sb);
            const int kofs = kb * 32 + 16 * u + 4 * hi;
#pragma unroll
            for (int t = 0; t < 2; ++t) dq[t] = sat_mfma_32x32x16_bf16(sat_att_frag_acc(kt_lds, t * 32 + l31, kofs), sb[0], dq[t]);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
#if !defined(SAT_HIPEMU)
__attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
sat_attn_bwd_dq_bf16_kernel(SatAttnParams p) {
    __shared__ __attribute__((aligned(16))) short k_lds2[2][SAT_ATT_T][SAT_ATT_ROW];    // [buffer][key][d]
    __shared__ __attribute__((aligned(16))) short v_lds2[2][SAT_ATT_T][SAT_ATT_ROW];    // [buffer][key][d]
    __shared__ __attribute__((aligned(16))) short kt_lds2[2][SAT_ATT_D][SAT_ATT_ROW];   // [buffer][d][key]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int hk = h / (p.H / p.Hkv);
    const int qrow = blockIdx.x * 128 + wave * 32 + l31;
    const bool q_in = qrow < p.Nqp, q_ok = qrow < p.Nq;
    const size_t qplane = ((size_t)b * p.H + h) * (size_t)p.Nqp * SAT_ATT_D;
    const size_t kplane = ((size_t)b * p.Hkv + hk) * (size_t)p.Nkp * SAT_ATT_D;
    const float l2e = 1.4426950408889634f;
    const float sl2 = p.scale * l2e;

    bf16x8 qf[4], gf[4];   // Q (pre-scaled) and dO fragments (d = 16 s + 8 hi + e)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        u32x4 w = u32x4{0u, 0u, 0u, 0u}, g = u32x4{0u, 0u, 0u, 0u};
        if (q_in) {
            w = *reinterpret_cast<const u32x4*>(p.q_rm[0] + qplane + (size_t)qrow * SAT_ATT_D + 16 * s + 8 * hi);
            g = *reinterpret_cast<const u32x4*>(p.do_rm[0] + qplane + (size_t)qrow * SAT_ATT_D + 16 * s + 8 * hi);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float e0 = __builtin_bit_cast(float, w[j] << 16), e1 = __builtin_bit_cast(float, w[j] & 0xffff0000u);
            w[j] = sat_cvt2_pk(e0 * sl2, e1 * sl2);
        }
        qf[s] = __builtin_bit_cast(bf16x8, w);
        gf[s] = __builtin_bit_cast(bf16x8, g);
    }
    const float lse2 = q_ok ? p.lse[((long long)b * p.H + h) * p.Nq + qrow] * l2e : 0.0f;
    const float dsum = q_ok ? p.dsum[((long long)b * p.H + h) * p.Nq + qrow] : 0.0f;
    f32x16 negl, negd, dq[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        negl[r] = -lse2;
        negd[r] = -dsum;
        dq[0][r] = 0.0f;
        dq[1][r] = 0.0f;
    }

    int srow[2], spart[2];
    unsigned rmo[2], tro[2];      // byte offsets of this thread's two pieces in a row-major / a transposed tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = threadIdx.x + j * 256;
        srow[j] = c >> 3;
        spart[j] = c & 7;
        rmo[j] = (unsigned)(srow[j] * SAT_ATT_D + spart[j] * 8) * 2u;
        tro[j] = ((unsigned)srow[j] * (unsigned)p.Nkp + (unsigned)spart[j] * 8u) * 2u;
    }
    const SatBuf kbuf = sat_buf_make(p.k_rm[0] + kplane), vbuf = sat_buf_make(p.v_rm[0] + kplane), ktbuf = sat_buf_make(p.k_tr[0] + kplane);
    bf16x8 rk[2], rv[2], rkt[2];
    auto tile_load = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            rk[j] = sat_buf_load16(kbuf, rmo[j], (unsigned)k0 * (SAT_ATT_D * 2));
            rv[j] = sat_buf_load16(vbuf, rmo[j], (unsigned)k0 * (SAT_ATT_D * 2));
            rkt[j] = sat_buf_load16(ktbuf, tro[j], (unsigned)k0 * 2u);
        }
    };
    auto tile_store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<bf16x8*>(&k_lds2[buf][srow[j]][spart[j] * 8]) = rk[j];
            *reinterpret_cast<bf16x8*>(&v_lds2[buf][srow[j]][spart[j] * 8]) = rv[j];
            *reinterpret_cast<bf16x8*>(&kt_lds2[buf][srow[j]][spart[j] * 8]) = rkt[j];
        }
    };
    tile_load(0);
    tile_store(0);
    if (SAT_ATT_T < p.Nk) tile_load(SAT_ATT_T);
    __syncthreads();
    int buf = 0, k0 = 0;
    for (; k0 + SAT_ATT_T <= p.Nk; k0 += SAT_ATT_T, buf ^= 1) {      // full tiles: one straight-line body
        if (k0 + SAT_ATT_T < p.Nk) {
            tile_store(buf ^ 1);
            if (k0 + 2 * SAT_ATT_T < p.Nk) tile_load(k0 + 2 * SAT_ATT_T);
        }
        sat_attn_dq_bf16_tile<false>(k_lds2[buf], v_lds2[buf], kt_lds2[buf], qf, gf, negl, negd, dq, l31, hi, SAT_ATT_T);
        __syncthreads();
    }
    if (k0 < p.Nk) sat_attn_dq_bf16_tile<true>(k_lds2[buf], v_lds2[buf], kt_lds2[buf], qf, gf, negl, negd, dq, l31, hi, p.Nk - k0);
    if (q_ok) {
        const long long obase = (((long long)b * p.H + h) * p.Nq + qrow) * SAT_ATT_D;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) SatOut<T>::put(p.dq, obase + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, dq[t][r] * p.scale);
    }
}

// ---------------------------------------------------------------------------------------------
// backward, part 2: dK, dV.  Wave owns 32 KEYS (columns); loops over the query heads of its kv group
// and over 64-query tiles.
//   S = Q K^T (rows q, cols key) ; P = exp(S*scale - lse[q]) ; dV^T += dO^T P ; dP = dO V^T ;
//   dS = P (dP - D[q]) scale ; dK^T += Q^T dS
// ---------------------------------------------------------------------------------------------
template <typename T, int NP, int TQ>   // TQ = queries per tile (64, or 32 for the two-plane variant: LDS budget)
__global__ void __launch_bounds__(256)
#if !defined(SAT_HIPEMU)
__attribute__((amdgpu_waves_per_eu(2, 2)))   // two workgroups per CU (2 x 75 KB of LDS): one stages while the other computes
#endif
sat_attn_bwd_dkv_kernel(SatAttnParams p) {
    constexpr int TROW = TQ + 8;           // transposed-tile row (80 B or 144 B stride: conflict-free)
    // the four query-side tiles (Q, dO row-major; Q^T, dO^T) are double-buffered: tile i+1 travels through registers while
    // tile i is consumed — one barrier per tile
    __shared__ __attribute__((aligned(16))) short q_lds2[2][NP][TQ][SAT_ATT_ROW];    // [buffer][plane][q][d]
    __shared__ __attribute__((aligned(16))) short g_lds2[2][NP][TQ][SAT_ATT_ROW];    // dO [q][d]
    __shared__ __attribute__((aligned(16))) short qt_lds2[2][NP][SAT_ATT_D][TROW];   // [d][q]
    __shared__ __attribute__((aligned(16))) short gt_lds2[2][NP][SAT_ATT_D][TROW];   // dO^T [d][q]
    __shared__ float lse_lds2[2][TQ], ds_lds2[2][TQ];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, hk = blockIdx.y;
    const int group = p.H / p.Hkv;
    const int krow = blockIdx.x * 128 + wave * 32 + l31;
    const bool k_in = krow < p.Nkp, k_ok = krow < p.Nk;
    const size_t kplane = ((size_t)b * p.Hkv + hk) * (size_t)p.Nkp * SAT_ATT_D;

    bf16x8 kf[4][NP], vf[4][NP];   // K and V fragments of this wave's keys (d = 16 s + 8 hi + e)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            if (k_in) {
                kf[s][pl] = *reinterpret_cast<const bf16x8*>(p.k_rm[pl] + kplane + (size_t)krow * SAT_ATT_D + 16 * s + 8 * hi);
                vf[s][pl] = *reinterpret_cast<const bf16x8*>(p.v_rm[pl] + kplane + (size_t)krow * SAT_ATT_D + 16 * s + 8 * hi);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    kf[s][pl][e] = 0;
                    vf[s][pl][e] = 0;
                }
            }
        }
    const float l2e = 1.4426950408889634f;
    const float sl2 = p.scale * l2e;
    f32x16 dk[2], dv[2];   // dK^T, dV^T: rows d, cols key
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dk[t][r] = 0.0f;
            dv[t][r] = 0.0f;
        }

    constexpr int PQ = TQ * 8 / 256;            // 16-byte pieces per thread of a [TQ][64] tile
    constexpr int PT = 64 * (TQ / 8) / 256;     // ... of a [64][TQ] tile
    bf16x8 rq[NP][PQ], rg[NP][PQ], rqt[NP][PT], rgt[NP][PT];
    float r_lse = 0.0f, r_ds = 0.0f;
    const int nqt = (p.Nq + TQ - 1) / TQ, ntiles = group * nqt;
    auto tile_load = [&](int it) {
        const int hg = it / nqt, q0 = (it - hg * nqt) * TQ;
        const int h = hk * group + hg;
        const size_t qplane = ((size_t)b * p.H + h) * (size_t)p.Nqp * SAT_ATT_D;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int j = 0; j < PQ; ++j) {
                const int c = threadIdx.x + j * 256, r = c >> 3, part = c & 7;
                rq[pl][j] = *reinterpret_cast<const bf16x8*>(p.q_rm[pl] + qplane + (size_t)(q0 + r) * SAT_ATT_D + part * 8);
                rg[pl][j] = *reinterpret_cast<const bf16x8*>(p.do_rm[pl] + qplane + (size_t)(q0 + r) * SAT_ATT_D + part * 8);
            }
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                const int c = threadIdx.x + j * 256, r = c / (TQ / 8), part = c - r * (TQ / 8);
                rqt[pl][j] = *reinterpret_cast<const bf16x8*>(p.q_tr[pl] + qplane + (size_t)r * p.Nqp + q0 + part * 8);
                rgt[pl][j] = *reinterpret_cast<const bf16x8*>(p.do_tr[pl] + qplane + (size_t)r * p.Nqp + q0 + part * 8);
            }
        }
        if (threadIdx.x < TQ) {
            const int q = q0 + threadIdx.x;
            const bool ok = q < p.Nq;
            r_lse = ok ? p.lse[((long long)b * p.H + h) * p.Nq + q] * l2e : 0.0f;
            r_ds = ok ? p.dsum[((long long)b * p.H + h) * p.Nq + q] : 0.0f;
        }
    };
    auto tile_store = [&](int buf) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int j = 0; j < PQ; ++j) {
                const int c = threadIdx.x + j * 256, r = c >> 3, part = c & 7;
                *reinterpret_cast<bf16x8*>(&q_lds2[buf][pl][r][part * 8]) = rq[pl][j];
                *reinterpret_cast<bf16x8*>(&g_lds2[buf][pl][r][part * 8]) = rg[pl][j];
            }
#pragma unroll
            for (int j = 0; j < PT; ++j) {
                const int c = threadIdx.x + j * 256, r = c / (TQ / 8), part = c - r * (TQ / 8);
                *reinterpret_cast<bf16x8*>(&qt_lds2[buf][pl][r][part * 8]) = rqt[pl][j];
                *reinterpret_cast<bf16x8*>(&gt_lds2[buf][pl][r][part * 8]) = rgt[pl][j];
            }
        }
        if (threadIdx.x < TQ) {
            lse_lds2[buf][threadIdx.x] = r_lse;
            ds_lds2[buf][threadIdx.x] = r_ds;
        }
    };

    tile_load(0);
    tile_store(0);
    if (ntiles > 1) tile_load(1);
    __syncthreads();
    for (int it = 0, buf = 0; it < ntiles; ++it, buf ^= 1) {
        short (*q_lds)[TQ][SAT_ATT_ROW] = q_lds2[buf];
        short (*g_lds)[TQ][SAT_ATT_ROW] = g_lds2[buf];
        short (*qt_lds)[SAT_ATT_D][TROW] = qt_lds2[buf];
        short (*gt_lds)[SAT_ATT_D][TROW] = gt_lds2[buf];
        const float* lse_lds = lse_lds2[buf];
        const float* ds_lds = ds_lds2[buf];
        const int q0 = (it % nqt) * TQ;
        if (it + 1 < ntiles) {
            tile_store(buf ^ 1);                          // tile it+1: registers -> the other buffer
            if (it + 2 < ntiles) tile_load(it + 2);       // tile it+2 -> registers
        }
        {
            {
#pragma unroll
            for (int qb = 0; qb < TQ / 32; ++qb) {
                f32x16 sacc, pacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    sacc[r] = 0.0f;
                    pacc[r] = 0.0f;
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    bf16x8 qa[NP], ga[NP];
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) {
                        qa[pl] = sat_att_frag_rm(q_lds[pl], qb * 32 + l31, 16 * s + 8 * hi);
                        ga[pl] = sat_att_frag_rm(g_lds[pl], qb * 32 + l31, 16 * s + 8 * hi);
                    }
                    sacc = sat_att_mma<NP>(qa, kf[s], sacc);   // S[q][key]
                    pacc = sat_att_mma<NP>(ga, vf[s], pacc);   // dP[q][key]
                }
                f32x16 pr;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ql = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const bool ok = (q0 + ql) < p.Nq && k_ok;
                    const float pv = ok ? sat_exp2(fmaf(sacc[r], sl2, -lse_lds[ql])) : 0.0f;
                    pr[r] = pv;
                    sacc[r] = pv * (pacc[r] - ds_lds[ql]) * p.scale;   // dS[q][key]
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    bf16x8 pb[NP], sb[NP];
                    sat_att_pack<NP>(pr, u, pb);
                    sat_att_pack<NP>(sacc, u, sb);
                    const int qofs = qb * 32 + 16 * u + 4 * hi;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        bf16x8 ga[NP], qa[NP];
#pragma unroll
                        for (int pl = 0; pl < NP; ++pl) {
                            ga[pl] = sat_att_frag_acc(gt_lds[pl], t * 32 + l31, qofs);
                            qa[pl] = sat_att_frag_acc(qt_lds[pl], t * 32 + l31, qofs);
                        }
                        dv[t] = sat_att_mma<NP>(ga, pb, dv[t]);
                        dk[t] = sat_att_mma<NP>(qa, sb, dk[t]);
                    }
                }
            }
            }
        }
        __syncthreads();
    }
    if (k_ok) {
        const long long obase = (((long long)b * p.Hkv + hk) * p.Nk + krow) * SAT_ATT_D;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                SatOut<T>::put(p.dk, obase + d, dk[t][r]);
                SatOut<T>::put(p.dv, obase + d, dv[t][r]);
            }
    }
}

// ---------------------------------------------------------------------------------------------
// The bf16 dK / dV kernel (round 5: THE bf16 kernel, as the dQ kernel above; the two-plane kernel serves the fp32 mode).  The general
// kernel issues ~700 instructions per 64-query tile and wave for 32 MFMAs: 192 of them a BRANCHING per-element mask (v_cmp, s_and,
// s_and_saveexec, s_cbranch_execz, v_mov, s_or per score), 64 ds_read_b32 + as many waits for the per-row lse / D broadcasts, 5 VALU per
// score of softmax / dS arithmetic.  Here:
//   * no masks.  PRECONDITION (what sat_attn_prepare writes): rows [N, Np) of every operand plane are ZERO.  Then a query row
//     q >= Nq has S = 0, dP = 0, and — its staged -lse and -D being 0 — P = 1, dS = 0: it adds exactly 0 * 1 to dV^T (dO^T is zero
//     there) and Q^T * 0 to dK^T; a key column >= Nk is never stored;
//   * the staged -lse (exp2 domain) and -D of a 32-query block are read as FOUR 16-byte LDS reads each; -D enters the dP MFMA chain as its
//     C operand, -lse the one fma per score that scales the product in fp32 (pre-scaling K here was tried: with the forward's LSE built
//     from a pre-scaled Q the two roundings disagree by |s| 2^-9 in the exponent — 3 % on the spike cases — so the general kernel's
//     arithmetic is kept): per score fma, exp2, one multiply and two half-converts; `scale` of dS is applied to dK once at the end;
//   * Q / dO / Q^T / dO^T tile loads through buffer descriptors (one per plane for the whole kernel; head and tile enter as a scalar offset).
// ---------------------------------------------------------------------------------------------
// four 4-vectors -> one 16-register block (registers 4 g + e = v_g[e]) without element-wise copies
SAT_DEVICE f32x16 sat_att_cat4(f32x4 v0, f32x4 v1, f32x4 v2, f32x4 v3) {
#if defined(SAT_HIPEMU)
    f32x16 o;
    for (int e = 0; e < 4; ++e) {
        o[e] = v0[e];
        o[4 + e] = v1[e];
        o[8 + e] = v2[e];
        o[12 + e] = v3[e];
    }
    return o;
#else
    typedef float sat_f8v __attribute__((ext_vector_type(8)));
    const sat_f8v lo = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7), hi = __builtin_shufflevector(v2, v3, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
#endif
}

template <typename T>
__global__ void __launch_bounds__(256)
#if !defined(SAT_HIPEMU)
__attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
sat_attn_bwd_dkv_bf16_kernel(SatAttnParams p) {
    constexpr int TQ = 64, TROW = TQ + 8;
    __shared__ __attribute__((aligned(16))) short q_lds2[2][TQ][SAT_ATT_ROW];      // [buffer][q][d]
    __shared__ __attribute__((aligned(16))) short g_lds2[2][TQ][SAT_ATT_ROW];      // dO [q][d]
    __shared__ __attribute__((aligned(16))) short qt_lds2[2][SAT_ATT_D][TROW];     // [d][q]
    __shared__ __attribute__((aligned(16))) short gt_lds2[2][SAT_ATT_D][TROW];     // dO^T [d][q]
    __shared__ __attribute__((aligned(16))) float nl_lds2[2][TQ], nd_lds2[2][TQ];  // -lse * log2(e), -D of the tile's queries
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.z, hk = blockIdx.y;
    const int group = p.H / p.Hkv;
    const int krow = blockIdx.x * 128 + wave * 32 + l31;
    const bool k_in = krow < p.Nkp, k_ok = krow < p.Nk;
    const size_t kplane = ((size_t)b * p.Hkv + hk) * (size_t)p.Nkp * SAT_ATT_D;
    const float l2e = 1.4426950408889634f;
    const float sl2 = p.scale * l2e;

    bf16x8 kf[4], vf[4];   // K and V fragments of this wave's keys (d = 16 s + 8 hi + e)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        u32x4 w = u32x4{0u, 0u, 0u, 0u}, v = u32x4{0u, 0u, 0u, 0u};
        if (k_in) {
            w = *reinterpret_cast<const u32x4*>(p.k_rm[0] + kplane + (size_t)krow * SAT_ATT_D + 16 * s + 8 * hi);
            v = *reinterpret_cast<const u32x4*>(p.v_rm[0] + kplane + (size_t)krow * SAT_ATT_D + 16 * s + 8 * hi);
        }
        kf[s] = __builtin_bit_cast(bf16x8, w);
        vf[s] = __builtin_bit_cast(bf16x8, v);
    }
    f32x16 dk[2], dv[2];   // dK^T / scale, dV^T: rows d, cols key
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dk[t][r] = 0.0f;
            dv[t][r] = 0.0f;
        }

    // this thread's two 16-byte pieces of a row-major [64][64] tile and of a transposed [64][64] tile
    int rrow[2], rpart[2];
    unsigned rmo[2], tro[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = threadIdx.x + j * 256;
        rrow[j] = c >> 3;
        rpart[j] = c & 7;
        rmo[j] = (unsigned)(rrow[j] * SAT_ATT_D + rpart[j] * 8) * 2u;
        tro[j] = ((unsigned)rrow[j] * (unsigned)p.Nqp + (unsigned)rpart[j] * 8u) * 2u;
    }
    // one descriptor per plane, based at this (batch item, kv group)'s first query head; head hg and tile q0 enter as a scalar offset
    const size_t gplane = ((size_t)b * p.H + (size_t)hk * group) * (size_t)p.Nqp * SAT_ATT_D;
    const SatBuf qbuf = sat_buf_make(p.q_rm[0] + gplane), gbuf = sat_buf_make(p.do_rm[0] + gplane);
    const SatBuf qtbuf = sat_buf_make(p.q_tr[0] + gplane), gtbuf = sat_buf_make(p.do_tr[0] + gplane);
    bf16x8 rq[2], rg[2], rqt[2], rgt[2];
    float r_nl = 0.0f, r_nd = 0.0f;
    const int nqt = (p.Nq + TQ - 1) / TQ, ntiles = group * nqt;
    auto tile_load = [&](int it) {
        const int hg = it / nqt, q0 = (it - hg * nqt) * TQ;
        const unsigned hoff = (unsigned)hg * (unsigned)p.Nqp * (SAT_ATT_D * 2u);      // bytes: < group * Nqp * 128
        const unsigned so_rm = hoff + (unsigned)q0 * (SAT_ATT_D * 2u), so_tr = hoff + (unsigned)q0 * 2u;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            rq[j] = sat_buf_load16(qbuf, rmo[j], so_rm);
            rg[j] = sat_buf_load16(gbuf, rmo[j], so_rm);
            rqt[j] = sat_buf_load16(qtbuf, tro[j], so_tr);
            rgt[j] = sat_buf_load16(gtbuf, tro[j], so_tr);
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
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<bf16x8*>(&q_lds2[buf][rrow[j]][rpart[j] * 8]) = rq[j];
            *reinterpret_cast<bf16x8*>(&g_lds2[buf][rrow[j]][rpart[j] * 8]) = rg[j];
            *reinterpret_cast<bf16x8*>(&qt_lds2[buf][rrow[j]][rpart[j] * 8]) = rqt[j];
            *reinterpret_cast<bf16x8*>(&gt_lds2[buf][rrow[j]][rpart[j] * 8]) = rgt[j];
        }
        if (threadIdx.x < TQ) {
            nl_lds2[buf][threadIdx.x] = r_nl;
            nd_lds2[buf][threadIdx.x] = r_nd;
        }
    };

    tile_load(0);
    tile_store(0);
    if (ntiles > 1) tile_load(1);
    __syncthreads();
    for (int it = 0, buf = 0; it < ntiles; ++it, buf ^= 1) {
        short (*q_lds)[SAT_ATT_ROW] = q_lds2[buf];
        short (*g_lds)[SAT_ATT_ROW] = g_lds2[buf];
        short (*qt_lds)[TROW] = qt_lds2[buf];
        short (*gt_lds)[TROW] = gt_lds2[buf];
        const float* nl_lds = nl_lds2[buf];
        const float* nd_lds = nd_lds2[buf];
        if (it + 1 < ntiles) {
            tile_store(buf ^ 1);
            if (it + 2 < ntiles) tile_load(it + 2);
        }
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
                pr[r] = sat_exp2(fmaf(sacc[r], sl2, nl[r]));       // the product kernel's arithmetic: fp32 scaling of the unscaled product
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
        __syncthreads();
    }
    if (k_ok) {
        const long long obase = (((long long)b * p.Hkv + hk) * p.Nk + krow) * SAT_ATT_D;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                SatOut<T>::put(p.dk, obase + d, dk[t][r] * p.scale);
                SatOut<T>::put(p.dv, obase + d, dv[t][r]);
            }
    }
}

// D[b][h][q] = sum_d dO[b][q][h*64+d] * O[b][q][h*64+d]    (both (B, Nq, H*64), model dtype)
struct SatRowdotParams {
    const void* a;
    const void* b;
    float* out;
    int B, H, Nq;
};
template <typename T>
__global__ void __launch_bounds__(256) sat_attn_rowdot_kernel(SatRowdotParams p) {
    const long long total = (long long)p.B * p.H * p.Nq;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // one wave per (b, h, q)
    if (row >= total) return;
    const int lane = threadIdx.x & 63;
    const int q = (int)(row % p.Nq);
    const long long bh = row / p.Nq;
    const int h = (int)(bh % p.H), b = (int)(bh / p.H);
    const long long i = ((long long)b * p.Nq + q) * ((long long)p.H * SAT_ATT_D) + (long long)h * SAT_ATT_D + lane;
    const float s = sat_wave_sum(SatSrc<T>::at(p.a, i) * SatSrc<T>::at(p.b, i));
    if (lane == 0) p.out[row] = s;
}

// ---------------------------------------------------------------------------------------------
// host entry points
// ---------------------------------------------------------------------------------------------
static int sat_attn_check(int B, int H, int Hkv, int Nq, int Nk, int Nqp, int Nkp, int head_dim, int dtype, const char* who) {
    if (B <= 0 || H <= 0 || Hkv <= 0 || Nq <= 0 || Nk <= 0) { sat_set_error(who); return 1; }
    if (head_dim != SAT_ATT_D) { sat_set_error("attention: only head_dim == 64 (the Stable Audio DiT) is implemented"); return 1; }
    if (H % Hkv != 0) { sat_set_error("attention: H must be a multiple of Hkv"); return 1; }
    if (Nqp < Nq || Nkp < Nk || Nqp % SAT_ATT_T || Nkp % SAT_ATT_T) { sat_set_error("attention: padded lengths must be multiples of 64 and cover N"); return 1; }
    if (dtype < 0 || dtype > 3) { sat_set_error("attention: dtype must be 0 (f32, split planes) or 1 (bf16; forward only: 2 / 3 = bf16 with 32 / 64 queries per wave forced)"); return 1; }
    // tile loads go through buffer descriptors with 32-bit byte offsets: 64 rows of a transposed plane, and a kv group's query heads
    if (Nkp >= (1 << 24) || (long long)Nqp * (H / Hkv) >= (1 << 24)) { sat_set_error("attention: sequences of 2^24 tokens or more are not supported"); return 1; }
    return 0;
}

extern "C" int sat_attention_fwd(const short* q_hi, const short* q_lo, const short* k_hi, const short* k_lo,
                                 const short* vt_hi, const short* vt_lo, void* o, float* lse, int B, int H, int Hkv,
                                 int Nq, int Nk, int Nqp, int Nkp, int head_dim, float scale, int dtype, void* stream) {
    if (sat_attn_check(B, H, Hkv, Nq, Nk, Nqp, Nkp, head_dim, dtype, "sat_attention_fwd: empty shape")) return 1;
    SatAttnParams p{};
    p.q_rm[0] = q_hi; p.q_rm[1] = q_lo; p.k_rm[0] = k_hi; p.k_rm[1] = k_lo; p.v_tr[0] = vt_hi; p.v_tr[1] = vt_lo;
    p.o = o; p.lse = lse; p.B = B; p.H = H; p.Hkv = Hkv; p.Nq = Nq; p.Nk = Nk; p.Nqp = Nqp; p.Nkp = Nkp; p.scale = scale;
    dim3 grid(sat_cdiv(Nq, 128), H, B);
    if (dtype == 0) {
        SAT_LAUNCH((sat_attn_fwd_kernel<float, 2>), grid, dim3(256), stream, p);
    } else {
        // bf16: 64 queries per wave (sat_attn_fwd_q_kernel<2>) for LONG sequences whose 256-query workgroups still put >= 2 on every CU,
        // else 32 — measured (profiles/r06_experiments/attn_q64/): N = 6145 520 vs 530 us (0.357 vs 0.350 of the bf16 peak), N = 1025
        // 164.7 vs 157.3 us at B = 16 (the ragged 1025 = 4 x 256 + 1 costs a fifth workgroup per head and two waves per SIMD overlap the
        // softmax less than three).  dtype 2 / 3 force the 32- / 64-query kernel (A/B runs and the tests' second implementation)
        const long long wg64 = (long long)sat_cdiv(Nq, 256) * H * B;
        const bool q64 = dtype == 3 || (dtype == 1 && Nq >= 2048 && wg64 >= 2LL * sat_cu_count());
        if (q64) SAT_LAUNCH((sat_attn_fwd_q_kernel<2>), dim3(sat_cdiv(Nq, 256), H, B), dim3(256), stream, p);
        else SAT_LAUNCH((sat_attn_fwd_kernel<short, 1>), grid, dim3(256), stream, p);
    }
    return sat_check_launch("sat_attention_fwd");
}

extern "C" int sat_attention_rowdot(const void* dout, const void* out, float* dsum, int B, int H, int Nq, int dtype,
                                    void* stream) {
    if (B <= 0 || H <= 0 || Nq <= 0) { sat_set_error("sat_attention_rowdot: empty shape"); return 1; }
    SatRowdotParams p{dout, out, dsum, B, H, Nq};
    dim3 grid((unsigned)sat_cdivll((long long)B * H * Nq, 4));
    if (dtype == 0) SAT_LAUNCH(sat_attn_rowdot_kernel<float>, grid, dim3(256), stream, p);
    else SAT_LAUNCH(sat_attn_rowdot_kernel<short>, grid, dim3(256), stream, p);
    return sat_check_launch("sat_attention_rowdot");
}

// planes[16]: {q_rm, k_rm, v_rm, k_tr, q_tr, do_rm, do_tr, (unused)} x {hi, lo}
extern "C" int sat_attention_bwd(const short* const* planes, const float* lse, const float* dsum, void* dq, void* dk,
                                 void* dv, int B, int H, int Hkv, int Nq, int Nk, int Nqp, int Nkp, int head_dim,
                                 float scale, int dtype, void* stream) {
    if (sat_attn_check(B, H, Hkv, Nq, Nk, Nqp, Nkp, head_dim, dtype, "sat_attention_bwd: empty shape")) return 1;
    if (dtype > 1) { sat_set_error("sat_attention_bwd: dtype must be 0 (f32) or 1 (bf16)"); return 1; }
    SatAttnParams p{};
    for (int i = 0; i < 2; ++i) {
        p.q_rm[i] = planes[0 + i]; p.k_rm[i] = planes[2 + i]; p.v_rm[i] = planes[4 + i]; p.k_tr[i] = planes[6 + i];
        p.q_tr[i] = planes[8 + i]; p.do_rm[i] = planes[10 + i]; p.do_tr[i] = planes[12 + i];
    }
    p.lse = const_cast<float*>(lse); p.dsum = dsum; p.dq = dq; p.dk = dk; p.dv = dv;
    p.B = B; p.H = H; p.Hkv = Hkv; p.Nq = Nq; p.Nk = Nk; p.Nqp = Nqp; p.Nkp = Nkp; p.scale = scale;
    dim3 g1(sat_cdiv(Nq, 128), H, B), g2(sat_cdiv(Nk, 128), Hkv, B);
    if (dtype == 0) {      // fp32 mode: two bf16 planes per operand, three MFMAs per product
        SAT_LAUNCH((sat_attn_bwd_dq_kernel<float, 2>), g1, dim3(256), stream, p);
        SAT_LAUNCH((sat_attn_bwd_dkv_kernel<float, 2, 32>), g2, dim3(256), stream, p);
    } else {
        SAT_LAUNCH((sat_attn_bwd_dq_bf16_kernel<short>), g1, dim3(256), stream, p);
        SAT_LAUNCH((sat_attn_bwd_dkv_bf16_kernel<short>), g2, dim3(256), stream, p);
    }
    return sat_check_launch("sat_attention_bwd");
}

#include "attention_cross.h"
