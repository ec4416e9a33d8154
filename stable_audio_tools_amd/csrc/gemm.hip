// gemm.hip — the dense projections of the DiT on the bf16 matrix cores, with the elementwise work that follows them
// fused into the epilogue (reference: stable_audio_tools/models/transformer.py — to_qkv :362/:481, to_out :364/:534,
// to_q / to_kv :356-357, GLU proj + x*silu(gate) :263-275, FF linear_out :308, the gate / residual updates :684-712).
//
//   C[M, N] = epilogue( A[M, K] · B[N, K]^T )        A, B bf16, both K-contiguous ("NT"), fp32 accumulation
//
// which is nn.Linear's forward as it stands (A = activations, B = weight (out, in)); the data gradient uses the transposed
// weight copy and the weight gradient the transposed activations (sat_transpose_bf16), so one kernel serves all three.
// float32 models run the same kernel on bf16x3-split operands laid out along K ([hi|hi|lo] x [hi|lo|hi], sat_split_bf16x3):
// three MFMAs per product, ~2^-17 relative per product, fp32 accumulate — the arithmetic of the conv stack.
//
// Structure (one workgroup = BM x BN tile, waves as a WGM x WGN grid of 64 x 64 or 128 x 64 sub-tiles of 32x32x16 MFMAs):
//   * K-steps of 64.  Operand tiles go global -> LDS directly (global_load_lds_dwordx4: 1 KiB = 8 rows x 128 B per wave
//     instruction, no VGPR round trip).  The LDS image is row-major with 128-byte rows; the 16-byte slot s of row r holds
//     k-chunk s ^ ((r >> 1) & 7): a ds_read_b128 lane group (16 lanes, 16 different rows, same chunk) then touches 16
//     different 16-byte slots of the 256-byte bank row — conflict free.  LDS-DMA writes are lane-linear, so the swizzle is
//     applied to the per-lane SOURCE address (same 128-byte global segment, permuted among 8 lanes).
//   * two stage buffers; tile k+1 is in flight while tile k feeds the MFMAs; one barrier per K-step.
//   * epilogue: each wave transposes its accumulators through a private 8 KiB LDS window (32 rows x 64 fp32) so that a
//     lane owns 4 consecutive columns: bias / residual / gate loads and the stores are 8- or 16-byte accesses that cover
//     whole 128-byte row segments.
#include "sat_device.h"
#include <type_traits>


enum { SAT_EPI_STORE = 0, SAT_EPI_RES = 1, SAT_EPI_GATE_RES = 2, SAT_EPI_SWIGLU = 3, SAT_EPI_QKV = 4 };

struct SatGemmParams {
    const short* A;       // (M, K) bf16
    const short* B;       // (N, K) bf16
    void* C;              // (M, ldc) bf16 or fp32; split-K: slab z at C + z * M * ldc (fp32)
    const float* bias;    // (N) fp32 or null
    const void* res;      // (M, ldr) dtype of C: added after the gate
    const void* gate;     // (M / rows_per_gate, ldg) dtype of C: v * sigmoid(1 - gate)   (transformer.py:684, :699)
    void* pre;            // SWIGLU: optional (M, ldp) copy of the pre-activation [x | gate] for the backward
    const short* zeros;   // >= 16 bytes of zeros: source of the k-chunks past K
    const float* alpha;   // device scalar multiplied into the accumulators before the epilogue (fp8 de-quantisation), or null
    const float* row_alpha; // (M) per-row factor applied with it (fp8 activations quantised per row: sat_quant_fp8_rows), or null
    const float* col_alpha; // (N) per-COLUMN factor = per-output-channel de-quantisation scale of B's rows (fp8 weights quantised row by
                            // row, round 6: one scale per output channel instead of one per tensor), or null
    long long lda, ldb, ldc, ldr, ldg, ldp;
    int M, N, K;
    int rows_per_gate;
    int klen;             // K range of one blockIdx.y slice (multiple of 64; == padded K without split-K)
    int ntm, ntn;
    // QKV epilogue: rotary + attention planes (attention.hip layouts)
    const float* rope_cs; // (ntok, 16, 2) cos/sin
    short* q_rm;          // (nb, H, Np, 64)
    short* k_rm;
    short* v_tr;          // (nb, H, 64, Np)
    int ntok, npad, heads, rope_off, sec0;   // sec0: section of the first column block (0 q, 1 k, 2 v)
};

// ---- staging -------------------------------------------------------------------------------------------------------
// One 1-KiB piece (8 tile rows x 128 B) of an operand tile: lane -> (row, 16-byte slot); the slot holds k-chunk slot ^ ((row>>1)&7).
template <bool GLU>
SAT_DEVICE void sat_gemm_stage_piece(const short* base, long long ld, int row0, int nrows, int k0, int kend, char* lds, const short* zeros,
                                     int p, int lane, int glu_f, int glu_tile0) {
    const int r = p * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int gr;
    if constexpr (GLU) {   // tile row -> weight row: 64-row groups of [32 value rows | 32 gate rows]  (transformer.py:274)
        gr = ((r >> 5) & 1) * glu_f + glu_tile0 + (r >> 6) * 32 + (r & 31);
    } else {
        gr = row0 + r;
    }
    gr = gr < nrows ? gr : nrows - 1;
    const int k = k0 + c * 8;
    const short* src = (k < kend) ? base + (long long)gr * ld + k : zeros;
    sat_glds16(src, lds + p * 1024);
}

SAT_DEVICE bf16x8 sat_gemm_frag(const char* tile, int row, int kc) {
    return *(const bf16x8*)(tile + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
}
// two 16-byte fragments -> the 32-byte operand of one MX MFMA (v_mfma_scale_f32_32x32x64_f8f6f4)
SAT_DEVICE i32x8 sat_cat8(bf16x8 lo, bf16x8 hi) {
    const u32x4 a = __builtin_bit_cast(u32x4, lo), b = __builtin_bit_cast(u32x4, hi);
    return i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
}

// ---- epilogue helpers ----------------------------------------------------------------------------------------------
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <bool F32>
SAT_DEVICE f32x4 sat_load4(const void* base, long long idx) {
    if constexpr (F32) {
        return *(const f32x4*)((const float*)base + idx);
    } else {
        const u32x2 u = *(const u32x2*)((const short*)base + idx);
        f32x4 v;
        v[0] = __builtin_bit_cast(float, u[0] << 16);
        v[1] = __builtin_bit_cast(float, u[0] & 0xffff0000u);
        v[2] = __builtin_bit_cast(float, u[1] << 16);
        v[3] = __builtin_bit_cast(float, u[1] & 0xffff0000u);
        return v;
    }
}
template <bool F32>
SAT_DEVICE void sat_store4(void* base, long long idx, f32x4 v) {
    if constexpr (F32) {
        *(f32x4*)((float*)base + idx) = v;
    } else {
        u32x2 u;
        u[0] = sat_cvt2_pk(v[0], v[1]);
        u[1] = sat_cvt2_pk(v[2], v[3]);
        *(u32x2*)((short*)base + idx) = u;
    }
}
// sigmoid on the transcendental unit: v_exp_f32 (2^x, 1 ulp) + v_rcp_f32 (1 ulp) — 4 issues instead of libm's expf + an IEEE divide
// (~25): the SwiGLU / gate epilogues evaluate it once per output element (FF1 at M = 2050: 64 per thread and tile)
SAT_DEVICE float sat_gemm_sigmoid(float x) {
#if defined(SAT_HIPEMU)
    return 1.0f / (1.0f + expf(-x));
#else
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
#endif
}

// ---- epilogue of one 32-row x 64-column window of a wave's accumulators ---------------------------------------------
// The wave has put the window into its private LDS space `ep`: row-major [32][64] fp32 — or, for a window of v columns of the
// fused QKV epilogue (sat_gemm_window_is_v), transposed [64 d][33].  A lane then owns 4 consecutive columns: bias / residual /
// gate loads and the stores are 8- or 16-byte accesses that cover whole 128-byte row segments.
template <int EPI>
SAT_DEVICE bool sat_gemm_window_is_v(const SatGemmParams& p, int nwin) {
    if constexpr (EPI == SAT_EPI_QKV) return nwin < p.N && nwin / (p.heads * 64) + p.sec0 == 2;
    else return false;
}

template <int EPI, bool F32OUT>
SAT_DEVICE void sat_gemm_epilogue_window(const SatGemmParams& p, const float* ep, int mrow0, int nwin, int glu_col0, int glu_f, int lane) {
    int wb0 = 0, wt0 = 0;                              // QKV: (batch item, token) of the window's first row — ONE division per window
    if constexpr (EPI == SAT_EPI_QKV) {                   // (round 4: the per-lane m / ntok, m % ntok of every pass were the bulk of this epilogue)
        wb0 = mrow0 / p.ntok;
        wt0 = mrow0 - wb0 * p.ntok;
    }
    if constexpr (EPI == SAT_EPI_QKV) {
        // a 64-column window is one head of q, k or v (wave-uniform).  v goes out TRANSPOSED (nb, H, 64, Np): the window was
        // staged as [64 d][33] so that a lane reads 4 consecutive tokens of one head dim (odd stride: conflict free)
        if (sat_gemm_window_is_v<EPI>(p, nwin)) {
            const int h = (nwin % (p.heads * 64)) >> 6;
            // Row groups aligned in the TOKEN index (round 4): the window's first row is token wt0 of its batch item; with an odd token
            // count (1 + 1024) every batch item but the first starts its rows off the 8-byte grid, and groups aligned in the row index
            // made all of their stores 2 + 4 + 2 bytes.  Lane group j takes rows a + 4j .. a + 4j + 3 (a = rows to the first token that
            // is a multiple of 4): one aligned 8-byte store; the 4 left-over rows (a in front, 4 - a behind) go to the lanes j = 7 as
            // single elements.  A window that crosses into the next batch item changes phase there: the general cases below stay.
            const int va = (4 - (wt0 & 3)) & 3;
            const int vj = lane & 7;
            const bool vsplit = va != 0 && vj == 7;
            const int tq = vsplit ? 0 : va + 4 * vj;
#pragma unroll
            for (int pass = 0; pass < 8; ++pass) {
                const int d = pass * 8 + (lane >> 3);
                const int m = mrow0 + tq;                 // first of the lane's rows (4 consecutive rows unless vsplit; may straddle a batch item)
                short o[4];
                float of[4];
                int re[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    re[e] = vsplit ? (e < va ? e : 28 + e) : tq + e;
                    of[e] = ep[d * 33 + re[e]];
                }
                if (p.row_alpha) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) of[e] *= p.row_alpha[mrow0 + re[e] < p.M ? mrow0 + re[e] : p.M - 1];
                }
                if (p.col_alpha) {
                    const float ca = p.col_alpha[nwin + d];      // (the window is inside N: sat_gemm_window_is_v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) of[e] *= ca;
                }
                const uint32_t q01 = sat_cvt2_pk(of[0], of[1]), q23 = sat_cvt2_pk(of[2], of[3]);     // packed RNE converts
                o[0] = (short)(q01 & 0xffffu); o[1] = (short)(q01 >> 16); o[2] = (short)(q23 & 0xffffu); o[3] = (short)(q23 >> 16);
                int b = wb0, t = wt0 + tq;                // (batch item, token) of row m: no per-lane division (window-level wb0 / wt0)
                while (t >= p.ntok) { t -= p.ntok; ++b; }
                short* dst = p.v_tr + (((long long)b * p.heads + h) * 64 + d) * p.npad + t;
                if (!vsplit && m + 3 < p.M && t + 3 < p.ntok) {
                    // 4 tokens of one batch item: the widest aligned stores their phase allows
                    const uint32_t p01 = q01, p23 = q23;
                    const uint32_t p12 = (q01 >> 16) | (q23 << 16);
                    if ((t & 3) == 0) {
                        *(u32x2*)dst = u32x2{p01, p23};
                    } else if ((t & 1) == 0) {
                        *(uint32_t*)dst = p01;
                        *(uint32_t*)(dst + 2) = p23;
                    } else {
                        dst[0] = o[0];
                        *(uint32_t*)(dst + 1) = p12;
                        dst[3] = o[3];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int me = mrow0 + re[e];
                        if (me < p.M) {
                            int be = wb0, te = wt0 + re[e];
                            while (te >= p.ntok) { te -= p.ntok; ++be; }
                            p.v_tr[(((long long)be * p.heads + h) * 64 + d) * p.npad + te] = o[e];
                        }
                    }
                }
            }
            return;
        }
    }
    if constexpr (EPI == SAT_EPI_SWIGLU) {
        // window columns 0..31 = value, 32..63 = gate of output columns glu_col0 + (0..31)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int rr = pass * 8 + (lane >> 3), cc = (lane & 7) * 4;
            const int m = mrow0 + rr;
            const int n = glu_col0 + cc;
            f32x4 xv = *(const f32x4*)(ep + rr * 64 + cc);
            f32x4 gv = *(const f32x4*)(ep + rr * 64 + 32 + cc);
            if (m < p.M && n < glu_f) {
                if (p.row_alpha) {
                    const float ra = p.row_alpha[m];
                    xv *= ra;
                    gv *= ra;
                }
                if (p.col_alpha) {
                    xv *= *(const f32x4*)(p.col_alpha + n);
                    gv *= *(const f32x4*)(p.col_alpha + glu_f + n);
                }
                if (p.bias) {
                    xv += *(const f32x4*)(p.bias + n);
                    gv += *(const f32x4*)(p.bias + glu_f + n);
                }
                if (p.pre) {
                    sat_store4<F32OUT>(p.pre, (long long)m * p.ldp + n, xv);
                    sat_store4<F32OUT>(p.pre, (long long)m * p.ldp + glu_f + n, gv);
                }
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = xv[e] * gv[e] * sat_gemm_sigmoid(gv[e]);
                sat_store4<F32OUT>(p.C, (long long)m * p.ldc + n, o);
            }
        }
    } else {
        // (QKV: two passes in flight — with all eight unrolled the rotary's table loads, partner columns and row factors of every pass are
        // live at once: the 256 x 256 kernel spilled 113 registers, 239 -> 844 us at M = 12290)
        constexpr int PU = (EPI == SAT_EPI_QKV) ? 2 : 8;
#pragma unroll 1
        for (int p0 = 0; p0 < 8; p0 += PU)
#pragma unroll
        for (int pi = 0; pi < PU; ++pi) {
            const int pass = p0 + pi;
            const int rr = pass * 4 + (lane >> 4), cc = (lane & 15) * 4;
            const int m = mrow0 + rr;
            const int n = nwin + cc;
            f32x4 v = *(const f32x4*)(ep + rr * 64 + cc);
            if (m < p.M && n < p.N) {
                if (p.row_alpha) v *= p.row_alpha[m];
                if (p.col_alpha) v *= *(const f32x4*)(p.col_alpha + n);
                if (p.bias) v += *(const f32x4*)(p.bias + n);
                if constexpr (EPI == SAT_EPI_GATE_RES) {
                    const f32x4 g = sat_load4<F32OUT>(p.gate, (long long)(m / p.rows_per_gate) * p.ldg + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= sat_gemm_sigmoid(1.0f - g[e]);
                }
                if constexpr (EPI == SAT_EPI_RES || EPI == SAT_EPI_GATE_RES) v += sat_load4<F32OUT>(p.res, (long long)m * p.ldr + n);
                if constexpr (EPI == SAT_EPI_QKV) {
                    // fused to_qkv epilogue (transformer.py:481-507): split heads, partial rotary on q and k (first 32 dims
                    // of each 64-dim head, rotate_half pairs (d, d+16)), and emit the attention kernel's operand planes:
                    // q, k row-major (nb, H, Np, 64), v transposed (nb, H, 64, Np).  A 64-column window is exactly one head.
                    const int hd = p.heads * 64;
                    const int nsec = nwin / hd;                                            // (wave-uniform: a window is one head)
                    const int which = nsec + p.sec0, h = (nwin - nsec * hd) >> 6, d = n & 63;   // 0 q, 1 k, 2 v
                    int b = wb0, t = wt0 + rr;
                    while (t >= p.ntok) { t -= p.ntok; ++b; }
                    if (p.rope_cs && d < 32) {
                        // partner column d ^ 16 lives 4 lanes away in this row's 16-lane group
                        f32x4 o;
                        f32x4 pv = *(const f32x4*)(ep + rr * 64 + (cc ^ 16));
                        if (p.row_alpha) pv *= p.row_alpha[m];     // (the partner column comes straight from the window: same row factor)
                        if (p.col_alpha) pv *= *(const f32x4*)(p.col_alpha + (n ^ 16));      // ... its own column factors (nwin is a multiple of 64)
                        // (cos, sin) of the lane's four dims: 8 consecutive floats, 32-byte aligned ((d & 15) is a multiple of 4)
                        const float* cs = p.rope_cs + ((long long)(t + p.rope_off) * 16 + (d & 15)) * 2;
                        const f32x4 cs0 = *(const f32x4*)cs, cs1 = *(const f32x4*)(cs + 4);
                        const float cc_[4] = {cs0[0], cs0[2], cs1[0], cs1[2]}, ss_[4] = {cs0[1], cs0[3], cs1[1], cs1[3]};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float c = cc_[e], s = ss_[e];
                            o[e] = (d < 16) ? v[e] * c - pv[e] * s : v[e] * c + pv[e] * s;
                        }
                        v = o;
                    }
                    short* dst = (which == 0) ? p.q_rm : p.k_rm;
                    sat_store4<false>(dst, (((long long)b * p.heads + h) * p.npad + t) * 64 + d, v);
                } else {
                    sat_store4<F32OUT>(p.C, (long long)blockIdx.y * p.M * p.ldc + (long long)m * p.ldc + n, v);
                }
            }
        }
    }
}

template <int BM, int BN, int WGM, int WGN, int NSTAGE, int PIPE, int EPI, bool F32OUT, bool FP8 = false>
__global__ void __launch_bounds__(WGM * WGN * 64) sat_gemm_kernel(SatGemmParams p) {
    constexpr int NW = WGM * WGN;
    constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
    constexpr int ABYTES = BM * 128, BBYTES = BN * 128, STAGE = ABYTES + BBYTES;
    static_assert(TN == 2 || TN == 4, "epilogue windows are 64 columns wide (one or two per wave)");
    constexpr int WIN = NSTAGE * STAGE / NW;     // per-wave epilogue window
    constexpr int G = (BM + BN) / 8 / NW;        // LDS-DMA instructions per wave per tile
    static_assert(WIN >= 64 * 33 * 4, "epilogue window (32 x 64 fp32, or 64 x 33 transposed) must fit");
    __shared__ __attribute__((aligned(16))) char smem[NSTAGE * STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = SAT_UNIFORM((int)(threadIdx.x >> 6));
    const int wm = wave / WGN, wn = wave % WGN;
    int tm, tn;
    sat_xcd_tile((int)blockIdx.x, p.ntm, p.ntn, &tm, &tn);      // the m-tiles of one weight panel share an XCD's L2
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.y * p.klen;
    const int kend = (kbeg + p.klen < p.K) ? kbeg + p.klen : p.K;
    const int nk = (kend - kbeg + 63) >> 6;
    constexpr bool GLU = (EPI == SAT_EPI_SWIGLU);
    const int glu_f = p.N >> 1, glu_tile0 = tn * (BN / 2);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // LDS-DMA instructions IB .. IE-1 (of G per wave) of tile kt into slot buf: instruction i of wave w moves piece i*NW + w of the
    // stage image [A tile | B tile]; BM/8 is a multiple of NW, so "A or B" is a compile-time property of i
    auto stage_range = [&](int kt, int buf, auto ib, auto ie) {
        constexpr int IB = decltype(ib)::value, IE = decltype(ie)::value;
        static_assert((BM / 8) % NW == 0, "A pieces must split evenly over the waves");
        char* s = smem + buf * STAGE;
        const int k0 = kbeg + kt * 64;
#pragma unroll
        for (int i = IB; i < IE; ++i) {
            if (i * NW < BM / 8) sat_gemm_stage_piece<false>(p.A, p.lda, m0, p.M, k0, kend, s, p.zeros, i * NW + wave, lane, 0, 0);
            else sat_gemm_stage_piece<GLU>(p.B, p.ldb, n0, p.N, k0, kend, s + ABYTES, p.zeros, i * NW + wave - BM / 8, lane, glu_f, glu_tile0);
        }
    };
    auto stage = [&](int kt, int buf) { stage_range(kt, buf, std::integral_constant<int, 0>{}, std::integral_constant<int, G>{}); };
    auto frags = [&](const char* As, int ks, bf16x8 (&a)[TM], bf16x8 (&b)[TN]) {
        const int kc = ks * 2 + (lane >> 5);
        const char* Bs = As + ABYTES;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = sat_gemm_frag(As, wm * (TM * 32) + i * 32 + (lane & 31), kc);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = sat_gemm_frag(Bs, wn * (TN * 32) + j * 32 + (lane & 31), kc);
    };
    auto mfmas = [&](const bf16x8 (&a)[TM], const bf16x8 (&b)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = sat_mfma_32x32x16_bf16(a[i], b[j], acc[i][j]);
    };
    if constexpr (PIPE == 1) {
        // Software pipeline: fragments of k-substep s+1 are read from LDS while the MFMAs of substep s run, and the hand-over to
        // the next tile (counted wait on the LDS-DMA queue, barrier, refill of the slot just drained, first fragments of the new
        // tile) sits between substeps 2 and 3, under the MFMAs of substep 2 and in front of those of substep 3.
        // All NSTAGE slots are in use: tile kt+NSTAGE is issued at K-step kt into the slot tile kt occupied.
#pragma unroll
        for (int s = 0; s < NSTAGE; ++s)
            if (s < nk) stage(s, s);
        if (nk >= NSTAGE) { SAT_WAIT_VMCNT((NSTAGE - 1) * G); } else { SAT_WAIT_VMCNT(0); }
        SAT_RAW_BARRIER();
        bf16x8 a0[TM], b0[TN], a1[TM], b1[TN];
        frags(smem, 0, a0, b0);
        int rd = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const char* As = smem + rd * STAGE;
            const int nx = (rd + 1 == NSTAGE) ? 0 : rd + 1;
            frags(As, 1, a1, b1);
            mfmas(a0, b0);
            SAT_SCHED_FENCE();
            frags(As, 2, a0, b0);
            mfmas(a1, b1);
            SAT_SCHED_FENCE();
            frags(As, 3, a1, b1);
            mfmas(a0, b0);
            SAT_SCHED_FENCE();
            if (kt + 1 < nk) {
                SAT_WAIT_LGKM0();          // every read of this tile's slot has completed (and a1 / b1 are in registers)
                if (kt + NSTAGE - 1 < nk) { SAT_WAIT_VMCNT((NSTAGE - 2) * G); } else { SAT_WAIT_VMCNT(0); }
                SAT_RAW_BARRIER();         // tile kt+1 has landed for everybody; the slot of tile kt is drained by everybody
                SAT_SCHED_FENCE();
                if (kt + NSTAGE < nk) stage(kt + NSTAGE, rd);
                frags(smem + nx * STAGE, 0, a0, b0);
            }
            SAT_SCHED_FENCE();
            mfmas(a1, b1);
            rd = nx;
        }
        SAT_WAIT_LGKM0();
    } else {
    // ring of NSTAGE tiles: tiles kt .. kt+NSTAGE-2 are in flight when K-step kt starts
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nk) stage(s, s);
    int rd = 0, wr = NSTAGE - 1;
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's pieces of tile kt have landed (the NSTAGE-2 younger tiles may still be in flight; the tail drains) ...
        if (kt + NSTAGE - 2 < nk) { SAT_WAIT_VMCNT((NSTAGE - 2) * G); } else { SAT_WAIT_VMCNT(0); }
        SAT_RAW_BARRIER();     // ... everybody's have, and every wave is done reading the slot tile kt+NSTAGE-1 goes to
        if (kt + NSTAGE - 1 < nk) stage(kt + NSTAGE - 1, wr);
        wr = (wr + 1 == NSTAGE) ? 0 : wr + 1;
        const char* As = smem + rd * STAGE;
        rd = (rd + 1 == NSTAGE) ? 0 : rd + 1;
        if constexpr (FP8) {
            // fp8 (e4m3) operands: the stage image is the same 128-byte-row layout, now 128 k-elements per row; one MX MFMA
            // consumes 64 of them — lanes 0-31 the 32 bytes (two 16-byte chunks) at 64 ks, lanes 32-63 those at 64 ks + 32
            static_assert(!FP8 || PIPE == 0, "the fp8 variant uses the plain loop");
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int c0 = ks * 4 + (lane >> 5) * 2;
                const char* Bs = As + ABYTES;
                i32x8 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm * (TM * 32) + i * 32 + (lane & 31);
                    const u32x4 lo = __builtin_bit_cast(u32x4, sat_gemm_frag(As, row, c0)), hi4 = __builtin_bit_cast(u32x4, sat_gemm_frag(As, row, c0 + 1));
                    a[i] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int row = wn * (TN * 32) + j * 32 + (lane & 31);
                    const u32x4 lo = __builtin_bit_cast(u32x4, sat_gemm_frag(Bs, row, c0)), hi4 = __builtin_bit_cast(u32x4, sat_gemm_frag(Bs, row, c0 + 1));
                    b[j] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = sat_mfma_32x32x64_fp8(a[i], b[j], acc[i][j]);
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a[TM], b[TN];
            frags(As, ks, a, b);
            mfmas(a, b);
        }
        }
    }
    }
    SAT_RAW_BARRIER();         // stage buffers are free (no LDS-DMA is pending): each wave takes a private window for its epilogue

    if (p.alpha) {
        const float al = *p.alpha;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= al;
    }
    float* ep = (float*)(smem + wave * WIN);
    const int hi = lane >> 5, col = lane & 31;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int jh = 0; jh < TN / 2; ++jh) {          // 64-column windows of the wave's TN * 32 columns
            sat_wave_sync();
            const int mrow0 = m0 + wm * (TM * 32) + i * 32;
            const int nwin = n0 + wn * (TN * 32) + jh * 64;
            if (sat_gemm_window_is_v<EPI>(p, nwin)) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ep[(j * 32 + col) * 33 + (r & 3) + 8 * (r >> 2) + 4 * hi] = acc[i][2 * jh + j][r];
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + j * 32 + col] = acc[i][2 * jh + j][r];
            }
            sat_wave_sync();
            sat_gemm_epilogue_window<EPI, F32OUT>(p, ep, mrow0, nwin, glu_tile0 + wn * (TN * 16) + jh * 32, glu_f, lane);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 256 x 256 tile, 8 waves, BK = 64, two wave rows ONE BARRIER APART — the large-tile schedule for the projections whose tile
// count fills the chip (QKV, FF1 and their data gradients).  cdna_hip_programming.md §5 ("256² 8-phase template") is the
// structure, here with 32 x 32 x 16 MFMAs and four barrier intervals per K-step; the 128 x 128 kernel above stays for the
// few-tile shapes (profiles/EXPERIMENTS.md round 3 has the A/B numbers, incl. the 16 x 16 x 32 / eight-interval and the
// four-wave 128 x 128-per-wave variants that lost).
//
//   * waves as 2 (M) x 4 (N): wave (wr, wc) owns rows wr*128 .. +128 and columns wc*64 .. +64 of the tile (4 x 2 accumulator
//     tiles of v_mfma_f32_32x32x16_bf16, the full-rate shape: 32 cycles per instruction and SIMD).  A K-step is TWO phases;
//     phase P multiplies the wave's row half P (64 rows) by its 64 columns: 2 x 2 tiles x 4 k-sub-steps = 16 MFMAs = 512
//     matrix-pipe cycles.
//   * a phase = [read section: fragment reads, the LDS-DMA of two HALF-TILES (128 rows x 128 B each, 4 instructions per wave),
//     lgkmcnt(0)] s_barrier [16 MFMAs at raised priority] s_barrier.  wr = 1 takes an extra barrier before the loop and wr = 0
//     one after it: in every interval one wave of each SIMD is in its MFMA section and its partner in its read section, so
//     fragment reads and DMA issue sit beside the partner's matrix work.
//   * staging.  Two tile buffers of 64 KB ([A 256 x 128 B | B 256 x 128 B], rows swizzled as in the 128² kernel).  Phase 0 reads
//     the B fragments (8 ds_read_b128, kept for both phases) and its 8 A fragments and issues A of tile t+1; phase 1 reads its A
//     fragments and issues B of tile t+2, then waits vmcnt(4) (tile t+1 complete, B(t+2) in flight) in front of its middle
//     barrier: the first read of tile t+1 is two barriers later for the waiting wave and one barrier after the other wave row's
//     wait.  Fragment reads are retired BEFORE the middle barrier, so a region is overwritten by LDS-DMA at least one barrier
//     after the last read of it completed (B(t): read in phase 0 of K-step t, overwritten from phase 1; A(t-1): read in phase 1
//     of K-step t-1, overwritten from phase 0 of K-step t).
//   * DMA source addresses: a per-lane offset (row, swizzled k-chunk) computed once + a wave-uniform base advanced by 128 B per
//     K-step (one VALU add per instruction); the K tail (K % 64 != 0) takes the general path.
// Rows of an M-tail tile beyond M are neither read nor multiplied (a 2050-row activation costs its ninth row tile the DMA
// stream only).
// FP8 (round 4): the stage image is the same (128-byte rows = 128 fp8 k-values, K-step = 128 of them); a wave's four 16-byte fragment
// registers per row block are then the operands of TWO MX MFMAs (32 x 32 x 64, twice the bf16 rate): the matrix time per stage byte,
// the DMA and the LDS traffic per interval are those of the bf16 kernel.
template <int EPI, bool F32OUT, bool FP8 = false>
__global__ void __launch_bounds__(512) sat_gemm256_kernel(SatGemmParams p) {
    constexpr int BM = 256, BN = 256;
    constexpr int ABYTES = BM * 128, BBYTES = BN * 128, STAGE = ABYTES + BBYTES;
    constexpr int WIN = 2 * STAGE / 8;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = SAT_UNIFORM((int)(threadIdx.x >> 6));
    const int wr = wave >> 2, wc = wave & 3;
    int tm, tn;
    sat_xcd_tile((int)blockIdx.x, p.ntm, p.ntn, &tm, &tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.y * p.klen;
    const int kend = (kbeg + p.klen < p.K) ? kbeg + p.klen : p.K;
    const int nk = (kend - kbeg + 63) >> 6;
    constexpr bool GLU = (EPI == SAT_EPI_SWIGLU);
    const int glu_f = p.N >> 1, glu_tile0 = tn * (BN / 2);
    const int mv = p.M - m0 - wr * 128;          // valid rows of this wave's 128 (<= 0: none)

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-lane source offsets (bytes, relative to the operand's base pointer) of this wave's 4 A pieces and 4 B pieces per tile:
    // piece q of half h is tile rows (h*16 + wave + 8*q) * 8 .. +8; lane -> (row, 16-byte slot holding k-chunk slot ^ ((row>>1)&7))
    long long aoff[4], boff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int piece = (q >> 1) * 16 + wave + 8 * (q & 1);
        const int r = piece * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int ga = m0 + r;
        ga = ga < p.M ? ga : p.M - 1;
        aoff[q] = ((long long)ga * p.lda + c * 8) * 2;
        int gb;
        if constexpr (GLU) gb = ((r >> 5) & 1) * glu_f + glu_tile0 + (r >> 6) * 32 + (r & 31);
        else gb = n0 + r;
        gb = gb < p.N ? gb : p.N - 1;
        boff[q] = ((long long)gb * p.ldb + c * 8) * 2;
    }
    // half-tiles: AH = 0 / 1 -> A rows 0..127 / 128..255 (pieces q = 2*AH, 2*AH+1); same for B
    auto stage_a = [&](int kt) {
        char* s = smem + (kt & 1) * STAGE;
        const int k0 = kbeg + kt * 64;
        const bool full = k0 + 64 <= kend;
        if (full) {
            const char* base = (const char*)p.A + (long long)k0 * 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) sat_glds16(base + aoff[q], s + ((q >> 1) * 16 + wave + 8 * (q & 1)) * 1024);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                sat_gemm_stage_piece<false>(p.A, p.lda, m0, p.M, k0, kend, s, p.zeros, (q >> 1) * 16 + wave + 8 * (q & 1), lane, 0, 0);
        }
    };
    auto stage_b = [&](int kt) {
        char* s = smem + (kt & 1) * STAGE + ABYTES;
        const int k0 = kbeg + kt * 64;
        const bool full = k0 + 64 <= kend;
        if (full) {
            const char* base = (const char*)p.B + (long long)k0 * 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) sat_glds16(base + boff[q], s + ((q >> 1) * 16 + wave + 8 * (q & 1)) * 1024);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                sat_gemm_stage_piece<GLU>(p.B, p.ldb, n0, p.N, k0, kend, s, p.zeros, (q >> 1) * 16 + wave + 8 * (q & 1), lane, glu_f, glu_tile0);
        }
    };

    // prologue: tile 0 complete, tile 1's B in flight
    stage_b(0); stage_a(0);
    if (nk > 1) {
        stage_b(1);
        SAT_WAIT_VMCNT(4);
    } else {
        SAT_WAIT_VMCNT(0);
    }
    SAT_RAW_BARRIER();
    if (wr == 1) SAT_RAW_BARRIER();              // the second wave row runs one barrier behind the first

    bf16x8 bfr[2][4], afr[2][4];
    const int frow = lane & 31, fkc = lane >> 5;
    // 16-byte k-chunk of fragment register q: bf16 — k-sub-step q, half fkc; fp8 — MX MFMA u = q >> 1 takes the 32 bytes at 64 u + 32 fkc
    auto kchunk = [&](int q) { return FP8 ? (q >> 1) * 4 + fkc * 2 + (q & 1) : q * 2 + fkc; };
    auto phase = [&](int t, auto pc) {
        constexpr int P = decltype(pc)::value;
        const char* As = smem + (t & 1) * STAGE;
        const char* Bs = As + ABYTES;
        const bool on0 = P * 64 < mv, on1 = P * 64 + 32 < mv;
        // ---- read section ----
        if constexpr (P == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) bfr[j][ks] = sat_gemm_frag(Bs, wc * 64 + j * 32 + frow, kchunk(ks));
        }
        if (on0) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) afr[0][ks] = sat_gemm_frag(As, wr * 128 + P * 64 + frow, kchunk(ks));
        }
        if (on1) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) afr[1][ks] = sat_gemm_frag(As, wr * 128 + P * 64 + 32 + frow, kchunk(ks));
        }
        if constexpr (P == 0) { if (t + 1 < nk) stage_a(t + 1); }
        if constexpr (P == 1) {
            if (t + 2 < nk) { stage_b(t + 2); SAT_WAIT_VMCNT(4); }
            else { SAT_WAIT_VMCNT(0); }
        }
        SAT_WAIT_LGKM0();
        SAT_RAW_BARRIER();
        // ---- MFMA section ----
        SAT_SCHED_FENCE();
        if (on0) {
            SAT_SETPRIO(1);
            if constexpr (FP8) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const i32x8 b0 = sat_cat8(bfr[0][2 * u], bfr[0][2 * u + 1]), b1 = sat_cat8(bfr[1][2 * u], bfr[1][2 * u + 1]);
                    const i32x8 a0 = sat_cat8(afr[0][2 * u], afr[0][2 * u + 1]);
                    acc[P * 2][0] = sat_mfma_32x32x64_fp8(a0, b0, acc[P * 2][0]);
                    acc[P * 2][1] = sat_mfma_32x32x64_fp8(a0, b1, acc[P * 2][1]);
                    if (on1) {
                        const i32x8 a1 = sat_cat8(afr[1][2 * u], afr[1][2 * u + 1]);
                        acc[P * 2 + 1][0] = sat_mfma_32x32x64_fp8(a1, b0, acc[P * 2 + 1][0]);
                        acc[P * 2 + 1][1] = sat_mfma_32x32x64_fp8(a1, b1, acc[P * 2 + 1][1]);
                    }
                }
            } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[P * 2][j] = sat_mfma_32x32x16_bf16(afr[0][ks], bfr[j][ks], acc[P * 2][j]);
                if (on1) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[P * 2 + 1][j] = sat_mfma_32x32x16_bf16(afr[1][ks], bfr[j][ks], acc[P * 2 + 1][j]);
                }
            }
            }
            SAT_SETPRIO(0);
        }
        SAT_SCHED_FENCE();
        SAT_RAW_BARRIER();
    };
    bool full_done = false;
    if constexpr (!FP8) {
        // Full waves (all 128 rows valid) — round 5: THE K loop of the bf16 kernel (round 4's "lean" arm, timed in
        // profiles/r05_experiments/lean_ab/: never slower than the general loop, FF1 + SwiGLU 87.4 -> 83.7 us at M = 2050).  The general
        // loop below spends 318 instructions per K-step and wave on 32 MFMAs: per-phase `on0 / on1` conditions around the fragment reads
        // and around every k-sub-step's MFMAs.  For a wave whose 128 rows are all valid the same per-wave sequence (phase 0: B + A
        // fragments, request A of tile t + 1, barrier, MFMAs, barrier; phase 1: A fragments, request B of tile t + 2, counted wait,
        // barrier, MFMAs, barrier) runs without the row conditions (ONE loop body: three bodies split by "requests both / only A /
        // nothing" spill at the 256-register cap); a wave with an M tail takes the general loop (same barriers per K-step, so the wave
        // rows of a workgroup may differ).  fp8: this loop's instances spill ~20 registers at the 256 cap — the fp8 256 x 256 kernel
        // keeps the general loop (the eight-wave kernels' fp8 instances do take theirs).
        if (mv >= 128) {
            auto half = [&](int t, auto pc) __attribute__((always_inline)) {
                constexpr int P = decltype(pc)::value;
                const char* As = smem + (t & 1) * STAGE;
                if constexpr (P == 0) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) bfr[j][ks] = sat_gemm_frag(As + ABYTES, wc * 64 + j * 32 + frow, kchunk(ks));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) afr[i][ks] = sat_gemm_frag(As, wr * 128 + P * 64 + i * 32 + frow, kchunk(ks));
                if constexpr (P == 0) { if (t + 1 < nk) stage_a(t + 1); }
                if constexpr (P == 1) {
                    if (t + 2 < nk) { stage_b(t + 2); SAT_WAIT_VMCNT(4); }
                    else { SAT_WAIT_VMCNT(0); }
                }
                SAT_WAIT_LGKM0();
                SAT_RAW_BARRIER();
                SAT_SCHED_FENCE();
                SAT_SETPRIO(1);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[P * 2 + i][j] = sat_mfma_32x32x16_bf16(afr[i][ks], bfr[j][ks], acc[P * 2 + i][j]);
                SAT_SETPRIO(0);
                SAT_SCHED_FENCE();
                SAT_RAW_BARRIER();
            };
            for (int t = 0; t < nk; ++t) {
                half(t, std::integral_constant<int, 0>{});
                half(t, std::integral_constant<int, 1>{});
            }
            full_done = true;
        }
    }
    if (!full_done)
    for (int t = 0; t < nk; ++t) {
        phase(t, std::integral_constant<int, 0>{});
        phase(t, std::integral_constant<int, 1>{});
    }
    if (wr == 0) SAT_RAW_BARRIER();              // pairs with the second wave row's last barrier: every LDS read is done
    SAT_RAW_BARRIER();

    if (p.alpha) {
        const float al = *p.alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= al;
    }
    float* ep = (float*)(smem + wave * WIN);
    const int hi = lane >> 5, col = lane & 31;
    const int nwin = n0 + wc * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i * 32 >= mv) break;                 // (wave-uniform)
        sat_wave_sync();
        const int mrow0 = m0 + wr * 128 + i * 32;
        if (sat_gemm_window_is_v<EPI>(p, nwin)) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) ep[(j * 32 + col) * 33 + (r & 3) + 8 * (r >> 2) + 4 * hi] = acc[i][j][r];
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + j * 32 + col] = acc[i][j][r];
        }
        sat_wave_sync();
        sat_gemm_epilogue_window<EPI, F32OUT>(p, ep, mrow0, nwin, glu_tile0 + wc * 32, glu_f, lane);
    }
}

// ---- the same eight-wave schedule on smaller tiles: ONE barrier interval pair per K-step, ring of NST >= 3 stages ---------------------
// For the shapes the 256 x 256 tile quantises badly on 256 CUs (round 4; profiles/EXPERIMENTS.md):
//   * QKV at M = 2050 is 8 x 18 = 144 full 256 x 256 tiles: ONE round of the chip at 56 % of its CUs.  As 160 x 256 tiles it is
//     13 x 18 = 234 workgroups of 0.625 of the work each; FF2 (6 column tiles, 96 K-steps) becomes 78 tiles x split-K 3 = 234;
//   * the 1536 -> 1536 projections (to_out, the cross-attention to_q) are 17 x 12 = 204 tiles of 128 x 128: on the four-wave kernel
//     above that is one wave per SIMD, which cannot hide its own fragment reads; here the same tile is eight waves of 32 x 64.
// Template: BN = 64 WC columns (WC = 4 or 2 wave columns, one 64-column epilogue window per wave); the waves form two GROUPS
// (group = wave >> 2: the two waves of a SIMD are in different groups), WRG = 4 / WC wave rows per group; a wave of group g owns
// R_g rows (NB_g = R_g / 32 MFMA row blocks) x 64 columns; BM = WRG (R0 + R1).  Instances: 128 x 256 (R 64 / 64, 3 stages of 48 KB:
// round 3's tile 6), 160 x 256 (R 96 / 64, 3 stages of 52 KB), 128 x 128 (WC 2, R 32 / 32, 4 stages of 32 KB).
//   * K-step t of a wave = [read section: its fragments of tile t (all four k-sub-steps), the LDS-DMA request of its pieces of tile
//     t + NST - 1, counted wait for its pieces of tile t + 1, lgkmcnt(0)] s_barrier [MFMA section at raised priority] s_barrier.
//     Group 1 takes one barrier before the loop and group 0 one after it, so in every interval one wave of each SIMD multiplies
//     while its partner reads — the matrix pipe alternates between the groups, which therefore need not be the same size.
//   * hazards.  Group 0 reads tile t in interval 2t, group 1 in interval 2t + 1.  WAR: tile t + NST - 1 goes to the slot of tile
//     t - 1, whose last fragment read retired (lgkmcnt(0)) before the barrier that closes interval 2t - 1; it is requested in
//     intervals 2t (group 0) and 2t + 1 (group 1).  RAW: a wave's pieces of tile t + 1 are waited for in its read section of K-step t
//     (intervals 2t / 2t + 1), i.e. in front of a barrier every reader of tile t + 1 (intervals 2t + 2 / 2t + 3) has passed.  With
//     two stages the wait would have to sit in the interval of the request: NST >= 3.
//   * 1-KiB pieces p = wave + 8 q of the stage image [A tile | B tile]; (BM + BN) / 8 pieces: when that is 4 mod 8 the waves of
//     group 0 carry one piece more (their counted waits differ by that piece).
template <int BN, int R0, int R1, int NST, int EPI, bool F32OUT, bool FP8>
__global__ void __launch_bounds__(512) sat_gemm8_kernel(SatGemmParams p) {
    constexpr int WC = BN / 64, WRG = 4 / WC;
    constexpr int BM = WRG * (R0 + R1);
    constexpr int NB0 = R0 / 32, NB1 = R1 / 32, NB = NB0 > NB1 ? NB0 : NB1;
    constexpr int PA = BM / 8, PT = (BM + BN) / 8;              // 1-KiB pieces: A tile, whole stage
    constexpr int QHI = (PT + 7) / 8, QLO = PT / 8;             // pieces per wave: group 0 / group 1
    constexpr int ABYTES = BM * 128, STAGE = (BM + BN) * 128;
    constexpr int WIN = NST * STAGE / 8;
    constexpr int LOOK = NST - 1;
    static_assert(WC == 4 || WC == 2, "64-column epilogue windows, eight waves");
    static_assert(R0 % 32 == 0 && R1 % 32 == 0 && NB0 >= NB1 && NB <= 4, "wave rows in MFMA blocks; group 0 is the larger one");
    static_assert(PT % 8 == 0 || PT % 8 == 4, "pieces split over the waves by group");
    static_assert(NST >= 3 && NST <= 4 && NST * STAGE <= 160 * 1024, "ring of 3 or 4 stages in 160 KB");
    static_assert(WIN >= 64 * 33 * 4, "epilogue window must fit");
    __shared__ __attribute__((aligned(16))) char smem[NST * STAGE];
    const int lane = threadIdx.x & 63;
    const int wave = SAT_UNIFORM((int)(threadIdx.x >> 6));
    const int grp = wave >> 2, wrl = (wave & 3) / WC, wc = (wave & 3) % WC;
    const int nb = grp ? NB1 : NB0;                              // this wave's MFMA row blocks
    const int arow = grp ? WRG * R0 + wrl * R1 : wrl * R0;       // its first row of the tile
    int tm, tn;
    sat_xcd_tile((int)blockIdx.x, p.ntm, p.ntn, &tm, &tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.y * p.klen;
    const int kend = (kbeg + p.klen < p.K) ? kbeg + p.klen : p.K;
    const int nk = (kend - kbeg + 63) >> 6;
    constexpr bool GLU = (EPI == SAT_EPI_SWIGLU);
    const int glu_f = p.N >> 1, glu_tile0 = tn * (BN / 2);
    const int mv = p.M - m0 - arow;                              // valid rows of this wave's (<= 0: none)

    f32x16 acc[NB][2];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // per-lane source address (at k = 0) of this wave's pieces: lane -> (row, 16-byte slot holding k-chunk slot ^ ((row >> 1) & 7))
    const char* src[QHI];
#pragma unroll
    for (int q = 0; q < QHI; ++q) {
        const int piece = wave + 8 * q;
        if (piece < PA) {
            const int r = piece * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int ga = m0 + r;
            ga = ga < p.M ? ga : p.M - 1;
            src[q] = (const char*)p.A + ((long long)ga * p.lda + c * 8) * 2;
        } else {
            const int r = (piece - PA) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int gb;
            if constexpr (GLU) gb = ((r >> 5) & 1) * glu_f + glu_tile0 + (r >> 6) * 32 + (r & 31);
            else gb = n0 + r;
            gb = gb < p.N ? gb : p.N - 1;
            src[q] = (const char*)p.B + ((long long)gb * p.ldb + c * 8) * 2;
        }
    }
    auto stage = [&](int kt) __attribute__((always_inline)) {
        char* s = smem + (kt % NST) * STAGE;
        const int k0 = kbeg + kt * 64;
        if (k0 + 64 <= kend) {
#pragma unroll
            for (int q = 0; q < QHI; ++q)
                if (q < QLO || grp == 0) sat_glds16(src[q] + (long long)k0 * 2, s + (wave + 8 * q) * 1024);
        } else {
#pragma unroll
            for (int q = 0; q < QHI; ++q) {
                if (!(q < QLO || grp == 0)) continue;
                const int piece = wave + 8 * q;
                if (piece < PA) sat_gemm_stage_piece<false>(p.A, p.lda, m0, p.M, k0, kend, s, p.zeros, piece, lane, 0, 0);
                else sat_gemm_stage_piece<GLU>(p.B, p.ldb, n0, p.N, k0, kend, s + ABYTES, p.zeros, piece - PA, lane, glu_f, glu_tile0);
            }
        }
    };
    // counted wait: leave this wave's pieces of `tiles` tiles in flight
    auto wait_tiles = [&](auto tiles) __attribute__((always_inline)) {
        constexpr int T = decltype(tiles)::value;
        if constexpr (QHI == QLO) { SAT_WAIT_VMCNT(T * QHI); }
        else { if (grp == 0) { SAT_WAIT_VMCNT(T * QHI); } else { SAT_WAIT_VMCNT(T * QLO); } }
    };
    using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>; using T2 = std::integral_constant<int, 2>;

    // prologue: tiles 0 .. LOOK-1 requested, tile 0 complete
#pragma unroll
    for (int s = 0; s < LOOK; ++s)
        if (s < nk) stage(s);
    if (nk >= LOOK) { if constexpr (LOOK == 3) wait_tiles(T2{}); else wait_tiles(T1{}); }
    else if (LOOK == 3 && nk == 2) wait_tiles(T1{});
    else wait_tiles(T0{});
    SAT_RAW_BARRIER();
    if (grp == 1) SAT_RAW_BARRIER();             // the second group runs one barrier behind the first

    bf16x8 bfr[2][4], afr[NB][4];
    const int frow = lane & 31, fkc = lane >> 5;
    // 16-byte k-chunk of fragment register q: bf16 — k-sub-step q, half fkc; fp8 — MX MFMA u = q >> 1 takes the 32 bytes at 64 u + 32 fkc
    auto kchunk = [&](int q) { return FP8 ? (q >> 1) * 4 + fkc * 2 + (q & 1) : q * 2 + fkc; };
    {
        // THE K loop (round 5; round 4's "lean" arm, timed in profiles/r05_experiments/lean_ab/: 160 x 256 tile QKV 40.5 -> 35.3 us and
        // FF2 104.6 -> 86.9 us at M = 2050, 128 x 128 tile FF2 54 -> 48 us; the general loop it replaces is gone).  That loop issued 374
        // instructions per K-step for the 24 MFMAs of a 96-row wave: ring slots by `t % NST` (mul_hi sequences), fragment addresses
        // rebuilt per row block, a branch per (row block, k-sub-step) for `i < nb && i * 32 < mv`, group-dependent piece and wait counts
        // decided at run time, end-of-loop conditions in every step — at one instruction per four cycles and wave that bounded the read
        // intervals (2 x ~700 instructions x 4 = the ~2700 cycles per K-step the cost model had fitted; the matrix pipe needs 1280).
        // Here the sequence of operations per wave and K-step — fragment reads, request of tile t + LOOK, counted wait, lgkmcnt(0),
        // barrier, MFMAs, barrier: the hazard argument above — is specialised at compile time on (group, all row blocks active), the
        // staged steps are split from the <= LOOK steps that stage nothing, the ring slots rotate in scalar registers and every fragment
        // address is one of eight per-lane bases + slot + an immediate.
        int nact = (mv + 31) >> 5;
        nact = nact < 0 ? 0 : (nact > nb ? nb : nact);          // this wave's active row blocks (wave-uniform)
        const int sw = (frow >> 1) & 7;                         // the same swizzle for every row block of a lane (blocks start at multiples of 32 rows)
        int ao[4], bo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = (kchunk(q) ^ sw) << 4;
            ao[q] = (arow + frow) * 128 + o;
            bo[q] = ABYTES + (wc * 64 + frow) * 128 + o;
        }
        auto kloop = [&](auto gc, auto fullc) __attribute__((always_inline)) {
            constexpr int G = decltype(gc)::value;
            constexpr bool FULL = decltype(fullc)::value != 0;
            constexpr int NBG = G ? NB1 : NB0, QG = G ? QLO : QHI;
            int slot = 0, sslot = LOOK % NST, k0s = kbeg + LOOK * 64;
            auto kstep = [&](auto stagec, int tail_wait) __attribute__((always_inline)) {
                constexpr bool STAGE_IT = decltype(stagec)::value != 0;
                const char* As = smem + slot * STAGE;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bfr[j][q] = *(const bf16x8*)(As + bo[q] + j * 4096);
#pragma unroll
                for (int i = 0; i < NBG; ++i) {
                    if (FULL || i < nact) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) afr[i][q] = *(const bf16x8*)(As + ao[q] + i * 4096);
                    }
                }
                if constexpr (STAGE_IT) {
                    char* sd = smem + sslot * STAGE;
                    if (k0s + 64 <= kend) {
#pragma unroll
                        for (int q = 0; q < QG; ++q) sat_glds16(src[q] + (long long)k0s * 2, sd + (wave + 8 * q) * 1024);
                    } else {
#pragma unroll
                        for (int q = 0; q < QG; ++q) {
                            const int piece = wave + 8 * q;
                            if (piece < PA) sat_gemm_stage_piece<false>(p.A, p.lda, m0, p.M, k0s, kend, sd, p.zeros, piece, lane, 0, 0);
                            else sat_gemm_stage_piece<GLU>(p.B, p.ldb, n0, p.N, k0s, kend, sd + ABYTES, p.zeros, piece - PA, lane, glu_f, glu_tile0);
                        }
                    }
                    SAT_WAIT_VMCNT((LOOK - 1) * QG);            // tiles t + 2 .. t + LOOK stay in flight: tile t + 1 is complete
                } else {
                    if (tail_wait) { SAT_WAIT_VMCNT(QG); } else { SAT_WAIT_VMCNT(0); }
                }
                SAT_WAIT_LGKM0();
                SAT_RAW_BARRIER();
                SAT_SCHED_FENCE();
                SAT_SETPRIO(1);
                if constexpr (FP8) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const i32x8 b0 = sat_cat8(bfr[0][2 * u], bfr[0][2 * u + 1]), b1 = sat_cat8(bfr[1][2 * u], bfr[1][2 * u + 1]);
#pragma unroll
                        for (int i = 0; i < NBG; ++i) {
                            if (FULL || i < nact) {
                                const i32x8 a = sat_cat8(afr[i][2 * u], afr[i][2 * u + 1]);
                                acc[i][0] = sat_mfma_32x32x64_fp8(a, b0, acc[i][0]);
                                acc[i][1] = sat_mfma_32x32x64_fp8(a, b1, acc[i][1]);
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                        for (int i = 0; i < NBG; ++i) {
                            if (FULL || i < nact) {
#pragma unroll
                                for (int j = 0; j < 2; ++j) acc[i][j] = sat_mfma_32x32x16_bf16(afr[i][ks], bfr[j][ks], acc[i][j]);
                            }
                        }
                    }
                }
                SAT_SETPRIO(0);
                SAT_SCHED_FENCE();
                SAT_RAW_BARRIER();
                slot = slot + 1 == NST ? 0 : slot + 1;
                sslot = sslot + 1 == NST ? 0 : sslot + 1;
                k0s += 64;
            };
            int t = 0;
            for (; t + LOOK < nk; ++t) kstep(std::integral_constant<int, 1>{}, 0);                 // steps that request tile t + LOOK
            for (; t < nk; ++t) kstep(std::integral_constant<int, 0>{}, (LOOK == 3 && t + 2 < nk) ? 1 : 0);   // the last <= LOOK steps
        };
        using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
        if (grp == 0) { if (nact == NB0) kloop(C0{}, C1{}); else kloop(C0{}, C0{}); }
        else { if (nact == NB1) kloop(C1{}, C1{}); else kloop(C1{}, C0{}); }
    }
    if (grp == 0) SAT_RAW_BARRIER();             // pairs with the second group's last barrier: every LDS read is done
    SAT_RAW_BARRIER();

    if (p.alpha) {
        const float al = *p.alpha;
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= al;
    }
    float* ep = (float*)(smem + wave * WIN);
    const int hi = lane >> 5, col = lane & 31;
    const int nwin = n0 + wc * 64;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        if (i >= nb || i * 32 >= mv) break;      // (wave-uniform)
        sat_wave_sync();
        const int mrow0 = m0 + arow + i * 32;
        if (sat_gemm_window_is_v<EPI>(p, nwin)) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) ep[(j * 32 + col) * 33 + (r & 3) + 8 * (r >> 2) + 4 * hi] = acc[i][j][r];
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * hi) * 64 + j * 32 + col] = acc[i][j][r];
        }
        sat_wave_sync();
        sat_gemm_epilogue_window<EPI, F32OUT>(p, ep, mrow0, nwin, glu_tile0 + wc * 32, glu_f, lane);
    }
}

template <int BN, int R0, int R1, int NST>
static int sat_gemm8_launch(SatGemmParams& p, int epi, int f32out, int splits, void* stream, bool fp8 = false) {
    constexpr int BM = (4 / (BN / 64)) * (R0 + R1);
    p.ntm = sat_cdiv(p.M, BM);
    p.ntn = sat_cdiv(epi == SAT_EPI_SWIGLU ? p.N / 2 : p.N, epi == SAT_EPI_SWIGLU ? BN / 2 : BN);
    dim3 grid(p.ntm * p.ntn, splits), block(512);
#define SAT_GEMM8_CASE(E, F)                                                                         \
    if (epi == E && f32out == (F ? 1 : 0)) {                                                         \
        if (fp8) { SAT_LAUNCH((sat_gemm8_kernel<BN, R0, R1, NST, E, F, true>), grid, block, stream, p); }   \
        else { SAT_LAUNCH((sat_gemm8_kernel<BN, R0, R1, NST, E, F, false>), grid, block, stream, p); }      \
        return sat_check_launch("sat_gemm (eight-wave ring)");                                       \
    }
    SAT_GEMM8_CASE(SAT_EPI_STORE, false)
    SAT_GEMM8_CASE(SAT_EPI_STORE, true)
    SAT_GEMM8_CASE(SAT_EPI_RES, false)
    SAT_GEMM8_CASE(SAT_EPI_RES, true)
    SAT_GEMM8_CASE(SAT_EPI_GATE_RES, false)
    SAT_GEMM8_CASE(SAT_EPI_GATE_RES, true)
    SAT_GEMM8_CASE(SAT_EPI_SWIGLU, false)
    SAT_GEMM8_CASE(SAT_EPI_SWIGLU, true)
    SAT_GEMM8_CASE(SAT_EPI_QKV, false)
#undef SAT_GEMM8_CASE
    sat_set_error("sat_gemm: unsupported epilogue / output type");
    return 1;
}

static int sat_gemm256_launch(SatGemmParams& p, int epi, int f32out, int splits, void* stream, bool fp8 = false) {
    p.ntm = sat_cdiv(p.M, 256);
    p.ntn = sat_cdiv(epi == SAT_EPI_SWIGLU ? p.N / 2 : p.N, epi == SAT_EPI_SWIGLU ? 128 : 256);
    dim3 grid(p.ntm * p.ntn, splits), block(512);
#define SAT_GEMM256_CASE(E, F)                                                                       \
    if (epi == E && f32out == (F ? 1 : 0)) {                                                         \
        if (fp8) { SAT_LAUNCH((sat_gemm256_kernel<E, F, true>), grid, block, stream, p); }           \
        else { SAT_LAUNCH((sat_gemm256_kernel<E, F>), grid, block, stream, p); }                     \
        return sat_check_launch("sat_gemm (256x256)");                                               \
    }
    SAT_GEMM256_CASE(SAT_EPI_STORE, false)
    SAT_GEMM256_CASE(SAT_EPI_STORE, true)
    SAT_GEMM256_CASE(SAT_EPI_RES, false)
    SAT_GEMM256_CASE(SAT_EPI_RES, true)
    SAT_GEMM256_CASE(SAT_EPI_GATE_RES, false)
    SAT_GEMM256_CASE(SAT_EPI_GATE_RES, true)
    SAT_GEMM256_CASE(SAT_EPI_SWIGLU, false)
    SAT_GEMM256_CASE(SAT_EPI_SWIGLU, true)
    SAT_GEMM256_CASE(SAT_EPI_QKV, false)
#undef SAT_GEMM256_CASE
    sat_set_error("sat_gemm: unsupported epilogue / output type");
    return 1;
}

static short* g_sat_zero_page = nullptr;   // set by the caller through sat_gemm_bf16's `zeros` argument (caller-owned)

template <int BM, int BN, int WGM, int WGN, int NSTAGE, int PIPE, bool FP8 = false>
static int sat_gemm_launch(SatGemmParams& p, int epi, int f32out, int splits, void* stream) {
    p.ntm = sat_cdiv(p.M, BM);
    p.ntn = sat_cdiv(epi == SAT_EPI_SWIGLU ? p.N / 2 : p.N, epi == SAT_EPI_SWIGLU ? BN / 2 : BN);
    dim3 grid(p.ntm * p.ntn, splits), block(WGM * WGN * 64);
#define SAT_GEMM_CASE(E, F)                                                                          \
    if (epi == E && f32out == (F ? 1 : 0)) {                                                         \
        SAT_LAUNCH((sat_gemm_kernel<BM, BN, WGM, WGN, NSTAGE, PIPE, E, F, FP8>), grid, block, stream, p);               \
        return sat_check_launch("sat_gemm_bf16");                                                    \
    }
    SAT_GEMM_CASE(SAT_EPI_STORE, false)
    SAT_GEMM_CASE(SAT_EPI_STORE, true)
    SAT_GEMM_CASE(SAT_EPI_RES, false)
    SAT_GEMM_CASE(SAT_EPI_RES, true)
    SAT_GEMM_CASE(SAT_EPI_GATE_RES, false)
    SAT_GEMM_CASE(SAT_EPI_GATE_RES, true)
    SAT_GEMM_CASE(SAT_EPI_SWIGLU, false)
    SAT_GEMM_CASE(SAT_EPI_SWIGLU, true)
    SAT_GEMM_CASE(SAT_EPI_QKV, false)
#undef SAT_GEMM_CASE
    sat_set_error("sat_gemm_bf16: unsupported epilogue / output type");
    return 1;
}

// The shipped workgroup tiles (round 5: the experiments 1, 2, 3, 5, 6 of rounds 2-4 are gone — profiles/EXPERIMENTS.md keeps their numbers):
//   0 = 128 x 128 / 4 waves / 2-slot ring / software-pipelined, two workgroups per CU (sat_gemm_kernel): few-tile shapes, split-K;
//   4 = 256 x 256 / 8 waves / two wave rows one barrier apart, 4 intervals per K-step (sat_gemm256_kernel);
//   sat_gemm8_kernel (eight waves in two groups one barrier apart, one interval pair per K-step, 3- or 4-stage ring):
//   7 = 160 x 256, 8 = 128 x 128.
static int sat_gemm_dispatch(SatGemmParams& p, int epi, int f32out, int splits, int tile, void* stream) {
    if (tile == 4) return sat_gemm256_launch(p, epi, f32out, splits, stream);
    if (tile == 7) return sat_gemm8_launch<256, 96, 64, 3>(p, epi, f32out, splits, stream);
    if (tile == 8) return sat_gemm8_launch<128, 32, 32, 4>(p, epi, f32out, splits, stream);
    if (tile != 0) { sat_set_error("sat_gemm: tile must be 0, 4, 7 or 8"); return 1; }
    return sat_gemm_launch<128, 128, 2, 2, 2, 1>(p, epi, f32out, splits, stream);
}
// fp8 operands: 0 = 128x128 / 4 waves / plain loop (round 3), 4 = 256x256, 7 = 160x256, 8 = 128x128 eight-wave ring
static int sat_gemm_dispatch_fp8(SatGemmParams& p, int epi, int f32out, int tile, void* stream) {
    if (tile == 4) return sat_gemm256_launch(p, epi, f32out, 1, stream, true);
    if (tile == 7) return sat_gemm8_launch<256, 96, 64, 3>(p, epi, f32out, 1, stream, true);
    if (tile == 8) return sat_gemm8_launch<128, 32, 32, 4>(p, epi, f32out, 1, stream, true);
    if (tile != 0) { sat_set_error("sat_gemm_fp8: tile must be 0, 4, 7 or 8"); return 1; }
    return sat_gemm_launch<128, 128, 2, 2, 2, 0, true>(p, epi, f32out, 1, stream);
}

// C = epilogue(A · B^T).  A (M, K), B (N, K) bf16 with row strides lda / ldb (elements, multiples of 8; K % 8 == 0).
// epilogue: 0 store [+bias]; 1 + res; 2 * sigmoid(1 - gate[m / rows_per_gate]) + res; 3 SwiGLU over B = [value rows | gate rows]
// (N = 2F, C is (M, F), optional pre-activation copy `pre` (M, 2F)).  out_f32: C / res / gate / pre are fp32 instead of bf16.
// splits > 1 (epilogue 0, fp32 out, no bias): K is cut into `splits` ranges, slab z is written at C + z * M * ldc — reduce with
// sat_reduce_splits.  zeros: >= 16 bytes of device zeros.  tile: 0 = 128 x 128 (4 waves), 1 = 256 x 128 (8 waves).
extern "C" int sat_gemm_bf16(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, const float* bias,
                             const void* res, long long ldr, const void* gate, long long ldg, int rows_per_gate, void* pre,
                             long long ldp, const void* zeros, int M, int N, int K, int epilogue, int out_f32, int splits, int tile,
                             void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) { sat_set_error("sat_gemm_bf16: empty shape"); return 1; }
    if ((K & 7) || (lda & 7) || (ldb & 7) || (N & 7) || (ldc & 3)) { sat_set_error("sat_gemm_bf16: K, N, lda, ldb must be multiples of 8"); return 1; }
    if (!zeros) { sat_set_error("sat_gemm_bf16: zeros page required"); return 1; }
    if (epilogue < 0 || epilogue > 3) { sat_set_error("sat_gemm_bf16: bad epilogue"); return 1; }
    if ((epilogue == SAT_EPI_RES || epilogue == SAT_EPI_GATE_RES) && !res) { sat_set_error("sat_gemm_bf16: residual missing"); return 1; }
    if (epilogue == SAT_EPI_GATE_RES && (!gate || rows_per_gate <= 0)) { sat_set_error("sat_gemm_bf16: gate missing"); return 1; }
    if (epilogue == SAT_EPI_SWIGLU && (N & 15)) { sat_set_error("sat_gemm_bf16: SwiGLU needs N = 2F with F % 8 == 0"); return 1; }
    if (splits < 1) splits = 1;
    if (splits > 1 && (epilogue != SAT_EPI_STORE || !out_f32 || bias)) { sat_set_error("sat_gemm_bf16: split-K is plain fp32 output only"); return 1; }
    SatGemmParams p{};
    p.A = (const short*)A; p.B = (const short*)B; p.C = C; p.bias = bias; p.res = res; p.gate = gate; p.pre = pre;
    p.zeros = (const short*)zeros;
    p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldr = ldr; p.ldg = ldg; p.ldp = ldp;
    p.M = M; p.N = N; p.K = K; p.rows_per_gate = rows_per_gate > 0 ? rows_per_gate : 1;
    const int ksteps = sat_cdiv(K, 64);
    p.klen = sat_cdiv(ksteps, splits) * 64;
    return sat_gemm_dispatch(p, epilogue, out_f32, splits, tile, stream);
}

// Attention input projections with the head split, the partial rotary and the attention kernel's operand layout fused into
// the epilogue (transformer.py:469-507).  B holds `nsec` consecutive blocks of heads*64 rows; block i is section sec0 + i
// (0 = q, 1 = k, 2 = v): to_qkv -> (sec0 0, nsec 3), to_q -> (0, 1), to_kv -> (1, 2).  q and k go to row-major planes
// (nb, heads, Np, 64), v to the transposed plane (nb, heads, 64, Np), all bf16.  rope_cs (>= ntok + rope_off, 16, 2) fp32
// cos/sin (sat_rope_tables) rotates the first 32 dims of every q / k head; NULL = no rotary (cross-attention, :689).
// Plane rows / columns >= ntok are NOT written: the caller zero-fills the planes once.
extern "C" int sat_gemm_qkv_bf16(const void* A, long long lda, const void* B, long long ldb, const float* rope_cs, int rope_off,
                                 void* q_rm, void* k_rm, void* v_tr, const void* zeros, int nb, int ntok, int npad, int heads,
                                 int K, int sec0, int nsec, int tile, void* stream) {
    if (nb <= 0 || ntok <= 0 || heads <= 0 || K <= 0 || npad < ntok) { sat_set_error("sat_gemm_qkv_bf16: bad shape"); return 1; }
    if ((K & 7) || (lda & 7) || (ldb & 7)) { sat_set_error("sat_gemm_qkv_bf16: K, lda, ldb must be multiples of 8"); return 1; }
    if (sec0 < 0 || nsec < 1 || sec0 + nsec > 3) { sat_set_error("sat_gemm_qkv_bf16: bad section range"); return 1; }
    if ((sec0 == 0 && !q_rm) || (sec0 <= 1 && sec0 + nsec > 1 && !k_rm) || (sec0 + nsec > 2 && !v_tr)) { sat_set_error("sat_gemm_qkv_bf16: missing plane"); return 1; }
    SatGemmParams p{};
    p.A = (const short*)A; p.B = (const short*)B; p.zeros = (const short*)zeros;
    p.lda = lda; p.ldb = ldb;
    p.M = nb * ntok; p.N = nsec * heads * 64; p.K = K; p.rows_per_gate = 1;
    p.klen = sat_cdiv(K, 64) * 64;
    p.rope_cs = rope_cs; p.rope_off = rope_off; p.q_rm = (short*)q_rm; p.k_rm = (short*)k_rm; p.v_tr = (short*)v_tr;
    p.ntok = ntok; p.npad = npad; p.heads = heads; p.sec0 = sec0;
    return sat_gemm_dispatch(p, SAT_EPI_QKV, 0, 1, tile, stream);
}

// Second half of a split-K projection: out[m][n] = sum_z slabs[z][m][n] (+ bias[n]) (+ res[m][n]), written in the output dtype.
// For the few-tile / long-K projections (FF2: 6144 -> 1536 at M = 2050 is 204 tiles of 96 K-steps on 256 CUs) cutting K in two
// doubles the workgroups; the slabs (fp32, 12.6 MB each) stay in the Infinity Cache between the two launches.
struct SatSplitEpiParams {
    const float* slabs;
    const float* bias;
    const void* res;
    void* out;
    long long ldr, ldo;
    int M, N, S, f32;
};
__global__ void __launch_bounds__(256) sat_splitk_epilogue_kernel(SatSplitEpiParams p) {
    const int nq = p.N >> 2;
    const long long total = (long long)p.M * nq;
    const long long slab = (long long)p.M * p.N;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / nq), n = (int)(i % nq) * 4;
        f32x4 v = *(const f32x4*)(p.slabs + (long long)m * p.N + n);
        for (int z = 1; z < p.S; ++z) v += *(const f32x4*)(p.slabs + z * slab + (long long)m * p.N + n);
        if (p.bias) v += *(const f32x4*)(p.bias + n);
        if (p.res) v += p.f32 ? sat_load4<true>(p.res, (long long)m * p.ldr + n) : sat_load4<false>(p.res, (long long)m * p.ldr + n);
        if (p.f32) sat_store4<true>(p.out, (long long)m * p.ldo + n, v);
        else sat_store4<false>(p.out, (long long)m * p.ldo + n, v);
    }
}
extern "C" int sat_splitk_epilogue(const float* slabs, int S, const float* bias, const void* res, long long ldr, void* out, long long ldo,
                                   int M, int N, int out_f32, void* stream) {
    if (M <= 0 || N <= 0 || S < 1 || (N & 3)) { sat_set_error("sat_splitk_epilogue: bad shape (N % 4 == 0)"); return 1; }
    SatSplitEpiParams p{slabs, bias, res, out, ldr, ldo, M, N, S, out_f32};
    const long long total = (long long)M * (N >> 2);
    SAT_LAUNCH(sat_splitk_epilogue_kernel, dim3((unsigned)(sat_cdivll(total, 256) < 4096 ? sat_cdivll(total, 256) : 4096)), dim3(256), stream, p);
    return sat_check_launch("sat_splitk_epilogue");
}

// fp8 (OCP e4m3) variants of the two entry points above for the forward projections of the long-context configuration
// (BASELINE.json configs[4]): A (M, K) and B (N, K) are fp8 bytes, K a multiple of 16, lda / ldb in elements (multiples of 16);
// products run on v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (twice the bf16 MFMA rate, half the operand bytes);
// alpha: device scalar = (de-quantisation scale of A) x (of B), multiplied into the fp32 accumulators before the epilogue; row_alpha
// (M floats, or NULL): a per-ROW factor applied with it — A quantised row by row (sat_quant_fp8_rows; alpha is then B's scale alone);
// col_alpha (N floats, 16-byte aligned, or NULL): a per-COLUMN factor — B's rows (output channels) quantised one by one with
// sat_quant_fp8_rows (round 6: the weight's scale is per output channel; alpha may then be NULL).
// Everything else (epilogues, output types, tile: 0 / 4 / 7 / 8) as sat_gemm_bf16 / sat_gemm_qkv_bf16; no split-K.
extern "C" int sat_gemm_fp8(const void* A, long long lda, const void* B, long long ldb, void* C, long long ldc, const float* bias,
                            const void* res, long long ldr, const void* gate, long long ldg, int rows_per_gate, void* pre,
                            long long ldp, const void* zeros, const float* alpha, const float* row_alpha, const float* col_alpha, int M, int N,
                            int K, int epilogue, int out_f32, int tile, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) { sat_set_error("sat_gemm_fp8: empty shape"); return 1; }
    if ((K & 15) || (lda & 15) || (ldb & 15) || (N & 7) || (ldc & 3)) { sat_set_error("sat_gemm_fp8: K, lda, ldb must be multiples of 16, N of 8"); return 1; }
    if (!zeros || (!alpha && !col_alpha)) { sat_set_error("sat_gemm_fp8: zeros page and alpha (or col_alpha) required"); return 1; }
    if (col_alpha && ((uintptr_t)col_alpha & 15)) { sat_set_error("sat_gemm_fp8: col_alpha must be 16-byte aligned"); return 1; }
    if (epilogue < 0 || epilogue > 3) { sat_set_error("sat_gemm_fp8: bad epilogue"); return 1; }
    if ((epilogue == SAT_EPI_RES || epilogue == SAT_EPI_GATE_RES) && !res) { sat_set_error("sat_gemm_fp8: residual missing"); return 1; }
    if (epilogue == SAT_EPI_GATE_RES && (!gate || rows_per_gate <= 0)) { sat_set_error("sat_gemm_fp8: gate missing"); return 1; }
    if (epilogue == SAT_EPI_SWIGLU && (N & 15)) { sat_set_error("sat_gemm_fp8: SwiGLU needs N = 2F with F % 8 == 0"); return 1; }
    SatGemmParams p{};
    // two fp8 elements = one "short" of the staging code: the kernel sees (M, K/2) x (N, K/2) 16-bit matrices
    p.A = (const short*)A; p.B = (const short*)B; p.C = C; p.bias = bias; p.res = res; p.gate = gate; p.pre = pre;
    p.zeros = (const short*)zeros; p.alpha = alpha; p.row_alpha = row_alpha; p.col_alpha = col_alpha;
    p.lda = lda / 2; p.ldb = ldb / 2; p.ldc = ldc; p.ldr = ldr; p.ldg = ldg; p.ldp = ldp;
    p.M = M; p.N = N; p.K = K / 2; p.rows_per_gate = rows_per_gate > 0 ? rows_per_gate : 1;
    p.klen = sat_cdiv(p.K, 64) * 64;
    return sat_gemm_dispatch_fp8(p, epilogue, out_f32, tile, stream);
}
extern "C" int sat_gemm_qkv_fp8(const void* A, long long lda, const void* B, long long ldb, const float* rope_cs, int rope_off,
                                void* q_rm, void* k_rm, void* v_tr, const void* zeros, const float* alpha, const float* row_alpha,
                                const float* col_alpha, int nb, int ntok, int npad, int heads, int K, int sec0, int nsec, int tile, void* stream) {
    if (nb <= 0 || ntok <= 0 || heads <= 0 || K <= 0 || npad < ntok) { sat_set_error("sat_gemm_qkv_fp8: bad shape"); return 1; }
    if ((K & 15) || (lda & 15) || (ldb & 15)) { sat_set_error("sat_gemm_qkv_fp8: K, lda, ldb must be multiples of 16"); return 1; }
    if (sec0 < 0 || nsec < 1 || sec0 + nsec > 3 || (!alpha && !col_alpha)) { sat_set_error("sat_gemm_qkv_fp8: bad section range / alpha"); return 1; }
    if (col_alpha && ((uintptr_t)col_alpha & 15)) { sat_set_error("sat_gemm_qkv_fp8: col_alpha must be 16-byte aligned"); return 1; }
    SatGemmParams p{};
    p.A = (const short*)A; p.B = (const short*)B; p.zeros = (const short*)zeros; p.alpha = alpha; p.row_alpha = row_alpha; p.col_alpha = col_alpha;
    p.lda = lda / 2; p.ldb = ldb / 2;
    p.M = nb * ntok; p.N = nsec * heads * 64; p.K = K / 2; p.rows_per_gate = 1;
    p.klen = sat_cdiv(p.K, 64) * 64;
    p.rope_cs = rope_cs; p.rope_off = rope_off; p.q_rm = (short*)q_rm; p.k_rm = (short*)k_rm; p.v_tr = (short*)v_tr;
    p.ntok = ntok; p.npad = npad; p.heads = heads; p.sec0 = sec0;
    return sat_gemm_dispatch_fp8(p, SAT_EPI_QKV, 0, tile, stream);
}

// Quantise to fp8 e4m3: dst[r][c] = sat_448(src[r][c] * qscale[0]) (round to nearest even); src fp32 or bf16 (R, C), row strides in
// elements.  qscale is a DEVICE scalar (448 / amax, computed by the caller without a host round trip).
struct SatQuantParams {
    const void* src;
    uint8_t* dst;
    const float* qscale;
    long long lds_, ldd;
    int R, Cc, src_f32;
};
__global__ void __launch_bounds__(256) sat_quant_fp8_kernel(SatQuantParams p) {
    const float qs = *p.qscale;
    const int cch = p.Cc >> 2;
    const long long total = (long long)p.R * cch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / cch), c = (int)(i % cch) * 4;
        float x[4];
        if (p.src_f32) {
            const float* s = (const float*)p.src + (long long)r * p.lds_ + c;
            for (int e = 0; e < 4; ++e) x[e] = s[e];
        } else {
            const short* s = (const short*)p.src + (long long)r * p.lds_ + c;
            for (int e = 0; e < 4; ++e) x[e] = sat_bf16_to_f32(s[e]);
        }
        *(uint32_t*)(p.dst + (long long)r * p.ldd + c) = sat_f32x4_to_fp8(x[0] * qs, x[1] * qs, x[2] * qs, x[3] * qs);
    }
}
extern "C" int sat_quant_fp8(const void* src, long long lds, void* dst, long long ldd, const float* qscale, int R, int C, int src_f32,
                             void* stream) {
    if (R <= 0 || C <= 0 || (C & 3) || (ldd & 3) || !qscale) { sat_set_error("sat_quant_fp8: C and ldd must be multiples of 4"); return 1; }
    SatQuantParams p{src, (uint8_t*)dst, qscale, lds, ldd, R, C, src_f32};
    const long long total = (long long)R * (C >> 2);
    SAT_LAUNCH(sat_quant_fp8_kernel, dim3((unsigned)(sat_cdivll(total, 256) < 8192 ? sat_cdivll(total, 256) : 8192)), dim3(256), stream, p);
    return sat_check_launch("sat_quant_fp8");
}

// Per-ROW dynamic quantisation in ONE pass (round 4): dst[r][:] = sat_448(src[r][:] * 448 / max|src[r][:]|), scale[r] = max|src[r][:]| / 448
// (clamped at 1e-12 / 448) — the per-token scale the GEMM's epilogue multiplies back (row_alpha).  One wave per row: the row lives in
// registers (<= 16 chunks of 8 elements per lane: C <= 8192) between the max reduction and the conversion, so the activation is read
// once; the per-tensor path read it twice (sat_absmax_scale + sat_quant_fp8: 35.7 + 14.7 us per projection input in the N = 6145
// sampler, 19 % of its step — profiles/EXPERIMENTS.md) and a token's outliers no longer set every other token's step size.
struct SatQuantRowsParams {
    const void* src;
    uint8_t* dst;
    float* scale;
    long long lds_, ldd;
    int R, Cc, src_f32;
};
template <bool F32>
__global__ void __launch_bounds__(256) sat_quant_fp8_rows_kernel(SatQuantRowsParams p) {
    constexpr int MAXCH = 16;
    constexpr int W = F32 ? 8 : 4;                         // dwords a lane keeps per 8-element chunk (bf16 stays packed)
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.R) return;                                  // (whole wave)
    const int nch = p.Cc >> 3;                             // 8-element chunks of the row
    uint32_t raw[MAXCH][W];
    auto value = [&](int i, int e) -> float {
        if constexpr (F32) return __builtin_bit_cast(float, raw[i][e]);
        else return __builtin_bit_cast(float, (e & 1) ? (raw[i][e >> 1] & 0xffff0000u) : (raw[i][e >> 1] << 16));
    };
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = i * 64 + lane;
        if (i * 64 < nch) {                                // (wave-uniform)
#pragma unroll
            for (int e = 0; e < W; ++e) raw[i][e] = 0u;
            if (c < nch) {
                if constexpr (F32) {
                    const u32x4 a = *(const u32x4*)((const float*)p.src + (long long)r * p.lds_ + c * 8);
                    const u32x4 b = *(const u32x4*)((const float*)p.src + (long long)r * p.lds_ + c * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { raw[i][e] = a[e]; raw[i][4 + e] = b[e]; }
                } else {
                    const u32x4 u = *(const u32x4*)((const short*)p.src + (long long)r * p.lds_ + c * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) raw[i][e] = u[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(value(i, e)));
        }
    }
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
    const float am = fmaxf(m, 1e-12f);
    const float qs = 448.0f / am;
    if (lane == 0) p.scale[r] = am / 448.0f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int c = i * 64 + lane;
        if (i * 64 < nch && c < nch) {
            typedef uint32_t u2 __attribute__((ext_vector_type(2)));
            const u2 w = {sat_f32x4_to_fp8(value(i, 0) * qs, value(i, 1) * qs, value(i, 2) * qs, value(i, 3) * qs),
                          sat_f32x4_to_fp8(value(i, 4) * qs, value(i, 5) * qs, value(i, 6) * qs, value(i, 7) * qs)};
            *(u2*)(p.dst + (long long)r * p.ldd + c * 8) = w;
        }
    }
}
extern "C" int sat_quant_fp8_rows(const void* src, long long lds, void* dst, long long ldd, float* scale, int R, int C, int src_f32,
                                  void* stream) {
    if (R <= 0 || C <= 0 || (C & 7) || C > 8192 || (ldd & 7) || (lds & 7) || !scale) {
        sat_set_error("sat_quant_fp8_rows: C, lds, ldd multiples of 8, C <= 8192");
        return 1;
    }
    SatQuantRowsParams p{src, (uint8_t*)dst, scale, lds, ldd, R, C, src_f32};
    if (src_f32) SAT_LAUNCH(sat_quant_fp8_rows_kernel<true>, dim3((unsigned)sat_cdiv(R, 4)), dim3(256), stream, p);
    else SAT_LAUNCH(sat_quant_fp8_rows_kernel<false>, dim3((unsigned)sat_cdiv(R, 4)), dim3(256), stream, p);
    return sat_check_launch("sat_quant_fp8_rows");
}

// Per-tensor dynamic scale of the fp8 quantisation in ONE launch: every block reduces max|x| over its share, the LAST block to
// arrive (device-scope ticket; agent-scope release / acquire around it, cdna_hip_programming.md §6 Guideline 16) reduces the partial
// maxima and writes scales[0] = 448 / amax (the quantisation scale sat_quant_fp8 reads) and scales[1] = amax / 448 (the
// de-quantisation scale for the GEMM's alpha), amax clamped at 1e-12.  work: caller-owned, >= 1 + gridDim floats; work[0] is the
// ticket counter — zero before the first call, reset by the kernel.
struct SatAbsmaxParams {
    const void* src;
    float* work;
    float* scales;
    long long lds_;
    int R, Cc, src_f32;
};
__global__ void __launch_bounds__(256) sat_absmax_scale_kernel(SatAbsmaxParams p) {
    __shared__ float red[4];
    __shared__ int last;
    const int cch = p.Cc >> 2;
    const long long total = (long long)p.R * cch;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / cch), c = (int)(i % cch) * 4;
        if (p.src_f32) {
            const f32x4 v = *(const f32x4*)((const float*)p.src + (long long)r * p.lds_ + c);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        } else {
            const f32x4 v = sat_load4<false>(p.src, (long long)r * p.lds_ + c);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
    }
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float bm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        p.work[1 + blockIdx.x] = bm;
#if !defined(SAT_HIPEMU)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        const unsigned ticket = atomicAdd((unsigned*)p.work, 1u);
        last = (ticket == gridDim.x - 1);
#if !defined(SAT_HIPEMU)
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
    }
    __syncthreads();
    if (!last) return;
    float g = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) g = fmaxf(g, p.work[1 + i]);
    for (int k = 32; k >= 1; k >>= 1) g = fmaxf(g, __shfl_xor(g, k));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = g;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float am = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), 1e-12f);
        p.scales[0] = 448.0f / am;
        p.scales[1] = am / 448.0f;
        *(unsigned*)p.work = 0u;
    }
}
extern "C" int sat_absmax_scale_blocks(int R, int C) {
    const long long total = (long long)R * (C >> 2);
    const long long b = sat_cdivll(total, 256 * 4);
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
// src (R, C) fp32|bf16 (row stride lds elements, C % 4 == 0) -> scales[0] = 448 / max|src|, scales[1] = max|src| / 448.
// work: >= 1 + sat_absmax_scale_blocks(R, C) floats, work[0] == 0 on entry (the kernel leaves it 0).
extern "C" int sat_absmax_scale(const void* src, long long lds, float* work, float* scales, int R, int C, int src_f32, void* stream) {
    if (R <= 0 || C <= 0 || (C & 3) || !work || !scales) { sat_set_error("sat_absmax_scale: bad arguments (C % 4 == 0)"); return 1; }
    SatAbsmaxParams p{src, work, scales, lds, R, C, src_f32};
    SAT_LAUNCH(sat_absmax_scale_kernel, dim3((unsigned)sat_absmax_scale_blocks(R, C)), dim3(256), stream, p);
    return sat_check_launch("sat_absmax_scale");
}

// ---------------------------------------------------------------------------------------------------------------------
// Operand preparation: cast / transpose / bf16x3 split, 16-byte accesses on both sides.
//   dst[c][r] (C, ldd) <- src[r][c] (R, lds)   (mode bit 0: transpose), source fp32 or bf16, destination bf16;
//   rows r in [R, Rpad) of a transposed destination are zero-filled (reduction-dim padding of the weight-gradient GEMM).
// ---------------------------------------------------------------------------------------------------------------------
struct SatCastParams {
    const void* src;
    short* dst;
    long long lds_, ldd;
    int R, Cc, Rpad, src_f32, transpose;
    short* dst2;            // transpose launches only: optional second destination (R, ldd2) — the un-transposed cast of the same tile
    long long ldd2;
};
// 8 consecutive source elements -> 8 bf16 (16 bytes); `vec` = the 16-byte path is legal for this launch
SAT_DEVICE u32x4 sat_cast_load8(const SatCastParams& p, int r, int c, bool vec) {
    u32x4 o = {0u, 0u, 0u, 0u};
    if (r >= p.R || c >= p.Cc) return o;
    if (vec && c + 8 <= p.Cc) {
        if (p.src_f32) {
            const f32x4* s = (const f32x4*)((const float*)p.src + (long long)r * p.lds_ + c);
            const f32x4 a = s[0], b = s[1];
            o[0] = sat_cvt2_pk(a[0], a[1]); o[1] = sat_cvt2_pk(a[2], a[3]);
            o[2] = sat_cvt2_pk(b[0], b[1]); o[3] = sat_cvt2_pk(b[2], b[3]);
        } else {
            o = *(const u32x4*)((const short*)p.src + (long long)r * p.lds_ + c);
        }
        return o;
    }
    for (int e = 0; e < 8; ++e) {
        uint32_t h = 0;
        if (c + e < p.Cc)
            h = (uint16_t)(p.src_f32 ? sat_f32_to_bf16(((const float*)p.src)[(long long)r * p.lds_ + c + e]) : ((const short*)p.src)[(long long)r * p.lds_ + c + e]);
        o[e >> 1] |= h << (16 * (e & 1));
    }
    return o;
}
// transpose of one 64 x 64 tile (tile coordinates bx = column tile, by = row tile) through LDS; rows of the LDS tile are 66 shorts apart
// (a column read walks 33 banks per row)
SAT_DEVICE void sat_cast_transpose_tile(const SatCastParams& p, short (*tile)[66], int bx, int by) {
    const bool vec = (p.transpose & 2) != 0;
    const int r0 = by * 64, c0 = bx * 64;
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {
        const int rr = i >> 3, cc = (i & 7) * 8;
        const u32x4 v = sat_cast_load8(p, r0 + rr, c0 + cc, vec);
        uint32_t* t = (uint32_t*)&tile[rr][cc];
        t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
        if (p.dst2 && r0 + rr < p.R && c0 + cc < p.Cc)      // (dual launches are vec launches with C % 8 == 0: checked on the host)
            *(u32x4*)(p.dst2 + (long long)(r0 + rr) * p.ldd2 + c0 + cc) = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 8; i += 256) {
        const int cc = i >> 3, rr = (i & 7) * 8;
        const int c = c0 + cc, r = r0 + rr;
        if (c >= p.Cc || r >= p.Rpad) continue;
        u32x4 o;
        for (int e = 0; e < 4; ++e)
            o[e] = (uint32_t)(uint16_t)tile[rr + 2 * e][cc] | ((uint32_t)(uint16_t)tile[rr + 2 * e + 1][cc] << 16);
        short* d = p.dst + (long long)c * p.ldd + r;
        if (vec && r + 8 <= p.Rpad) {
            *(u32x4*)d = o;
        } else {
            for (int e = 0; e < 8 && r + e < p.Rpad; ++e) d[e] = (short)(o[e >> 1] >> (16 * (e & 1)));
        }
    }
}
__global__ void __launch_bounds__(256) sat_cast_kernel(SatCastParams p) {
    // 16-byte accesses need 16-byte aligned rows on both sides (checked on the host: vec flag rides in `transpose` bit 1)
    const bool vec = (p.transpose & 2) != 0;
    if (!(p.transpose & 1)) {
        const int cchunks = (p.Cc + 7) >> 3;
        const long long total = (long long)p.R * cchunks;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const int r = (int)(i / cchunks), c = (int)(i % cchunks) * 8;
            const u32x4 v = sat_cast_load8(p, r, c, vec);
            short* d = p.dst + (long long)r * p.ldd + c;
            if (vec && c + 8 <= p.Cc) {
                *(u32x4*)d = v;
            } else {
                for (int e = 0; e < 8 && c + e < p.Cc; ++e) d[e] = (short)(v[e >> 1] >> (16 * (e & 1)));
            }
        }
        return;
    }
    __shared__ short tile[64][66];
    sat_cast_transpose_tile(p, tile, blockIdx.x, blockIdx.y);
}
// two independent transposing casts in ONE launch (blockIdx.z picks the tensor; tiles outside a tensor's range exit): the two operands of a
// weight-gradient GEMM, dz^T and x^T, are prepared together (round 6)
struct SatCastPairParams { SatCastParams a, b; };
__global__ void __launch_bounds__(256) sat_cast_pair_kernel(SatCastPairParams pp) {
    __shared__ short tile[64][66];
    const SatCastParams& p = blockIdx.z ? pp.b : pp.a;
    if ((int)blockIdx.x * 64 >= p.Cc || (int)blockIdx.y * 64 >= p.Rpad) return;      // block-uniform
    sat_cast_transpose_tile(p, tile, blockIdx.x, blockIdx.y);
}
// src (R, C) fp32|bf16 row stride lds -> dst bf16: (R, ldd) copy/cast, or transposed (C, ldd) with rows R..Rpad-1 zero.
extern "C" int sat_cast_bf16(const void* src, long long lds, void* dst, long long ldd, int R, int C, int Rpad, int src_f32,
                             int transpose, void* stream) {
    if (R <= 0 || C <= 0) { sat_set_error("sat_cast_bf16: empty shape"); return 1; }
    if (Rpad < R) Rpad = R;
    // 16-byte path: every row start 16-byte aligned on both sides
    const long long salign = src_f32 ? 4 : 8;
    const bool vec = ((unsigned long long)src % 16 == 0) && ((unsigned long long)dst % 16 == 0) && (lds % salign == 0) && (ldd % 8 == 0);
    SatCastParams p{src, (short*)dst, lds, ldd, R, C, Rpad, src_f32, (transpose ? 1 : 0) | (vec ? 2 : 0), nullptr, 0};
    if (transpose) {
        SAT_LAUNCH(sat_cast_kernel, dim3(sat_cdiv(C, 64), sat_cdiv(Rpad, 64)), dim3(256), stream, p);
    } else {
        const long long total = (long long)R * ((C + 7) / 8);
        SAT_LAUNCH(sat_cast_kernel, dim3((unsigned)(sat_cdivll(total, 256) < 8192 ? sat_cdivll(total, 256) : 8192)), dim3(256), stream, p);
    }
    return sat_check_launch("sat_cast_bf16");
}

// dst_a (Ca, Rpad_a) <- src_a (Ra, Ca)^T and dst_b (Cb, Rpad_b) <- src_b (Rb, Cb)^T in one launch; each as sat_cast_bf16(transpose = 1).
extern "C" int sat_cast_bf16_tpair(const void* src_a, long long lds_a, void* dst_a, long long ldd_a, int Ra, int Ca, int Rpad_a, int a_f32,
                                   const void* src_b, long long lds_b, void* dst_b, long long ldd_b, int Rb, int Cb, int Rpad_b, int b_f32,
                                   void* stream) {
    if (Ra <= 0 || Ca <= 0 || Rb <= 0 || Cb <= 0 || !src_a || !dst_a || !src_b || !dst_b) { sat_set_error("sat_cast_bf16_tpair: bad arguments"); return 1; }
    if (Rpad_a < Ra) Rpad_a = Ra;
    if (Rpad_b < Rb) Rpad_b = Rb;
    const bool vec_a = ((unsigned long long)src_a % 16 == 0) && ((unsigned long long)dst_a % 16 == 0) && (lds_a % (a_f32 ? 4 : 8) == 0) && (ldd_a % 8 == 0);
    const bool vec_b = ((unsigned long long)src_b % 16 == 0) && ((unsigned long long)dst_b % 16 == 0) && (lds_b % (b_f32 ? 4 : 8) == 0) && (ldd_b % 8 == 0);
    SatCastPairParams pp{SatCastParams{src_a, (short*)dst_a, lds_a, ldd_a, Ra, Ca, Rpad_a, a_f32, 1 | (vec_a ? 2 : 0), nullptr, 0},
                         SatCastParams{src_b, (short*)dst_b, lds_b, ldd_b, Rb, Cb, Rpad_b, b_f32, 1 | (vec_b ? 2 : 0), nullptr, 0}};
    const int gx = sat_cdiv(Ca > Cb ? Ca : Cb, 64), gy = sat_cdiv(Rpad_a > Rpad_b ? Rpad_a : Rpad_b, 64);
    SAT_LAUNCH(sat_cast_pair_kernel, dim3(gx, gy, 2), dim3(256), stream, pp);
    return sat_check_launch("sat_cast_bf16_tpair");
}

// One pass over a weight for BOTH bf16 copies a training step needs (round 6): dst (R, ldd) = cast(src) for the forward GEMM, dst_t
// (C, ldd_t) = its transpose (columns R..Rpad-1 zero) for the data-gradient GEMM — the source tile is read once.
extern "C" int sat_cast_bf16_dual(const void* src, long long lds, void* dst, long long ldd, void* dst_t, long long ldd_t, int R, int C,
                                  int Rpad, int src_f32, void* stream) {
    if (R <= 0 || C <= 0 || !src || !dst || !dst_t) { sat_set_error("sat_cast_bf16_dual: bad arguments"); return 1; }
    if (Rpad < R) Rpad = R;
    const long long salign = src_f32 ? 4 : 8;
    const bool vec = ((unsigned long long)src % 16 == 0) && ((unsigned long long)dst % 16 == 0) && ((unsigned long long)dst_t % 16 == 0) &&
                     (lds % salign == 0) && (ldd % 8 == 0) && (ldd_t % 8 == 0) && (C % 8 == 0);
    if (!vec) { sat_set_error("sat_cast_bf16_dual: needs 16-byte aligned rows on all three tensors and C % 8 == 0"); return 1; }
    SatCastParams p{src, (short*)dst_t, lds, ldd_t, R, C, Rpad, src_f32, 3, (short*)dst, ldd};
    SAT_LAUNCH(sat_cast_kernel, dim3(sat_cdiv(C, 64), sat_cdiv(Rpad, 64)), dim3(256), stream, p);
    return sat_check_launch("sat_cast_bf16_dual");
}

// fp32 (R, C) -> bf16 (R, 3C) split planes along K for the fp32-accurate GEMM: side 0 (activations / A) = [hi | hi | lo],
// side 1 (weights / B) = [hi | lo | hi], so that A' · B'^T = hi·hi + hi·lo + lo·hi.
struct SatSplitParams {
    const float* src;
    short* dst;
    long long lds_, ldd;
    int R, Cc, side;
};
__global__ void __launch_bounds__(256) sat_split_kernel(SatSplitParams p) {
    const long long total = (long long)p.R * (p.Cc >> 1);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / (p.Cc >> 1)), c = (int)(i % (p.Cc >> 1)) * 2;
        const float a = p.src[(long long)r * p.lds_ + c], b = p.src[(long long)r * p.lds_ + c + 1];
        uint32_t hi, lo;
        sat_split2_pk(a, b, &hi, &lo);
        uint32_t* d = (uint32_t*)(p.dst + (long long)r * p.ldd + c);
        d[0] = hi;
        d[p.Cc >> 1] = p.side ? lo : hi;
        d[p.Cc] = p.side ? hi : lo;
    }
}
extern "C" int sat_split_bf16x3(const float* src, long long lds, void* dst, long long ldd, int R, int C, int side, void* stream) {
    if (R <= 0 || C <= 0 || (C & 1)) { sat_set_error("sat_split_bf16x3: C must be even"); return 1; }
    SatSplitParams p{src, (short*)dst, lds, ldd, R, C, side};
    const long long total = (long long)R * (C >> 1);
    SAT_LAUNCH(sat_split_kernel, dim3((unsigned)(sat_cdivll(total, 256) < 4096 ? sat_cdivll(total, 256) : 4096)), dim3(256), stream, p);
    return sat_check_launch("sat_split_bf16x3");
}
