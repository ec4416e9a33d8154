// ru_k1_bwd.hip — the backward of a ResidualUnit's 1x1 convolution in ONE pass over its two operands (round 5).
//
// Reference graph (stable_audio_tools/models/autoencoders.py:58-83):   y = x + conv1(snake2(h)),  h = conv7(snake1(x)).
// With dy = dL/dy the autograd of `conv1(snake2(h))` needs
//   dW2[co][ci]  = sum_{b,t} dy[co][t] * snake2(h)[ci][t]                      (weight gradient, K = 1)
//   db2[co]      = sum_{b,t} dy[co][t]
//   dh[ci][t]    = (sum_co W2[co][ci] dy[co][t]) * dsnake2(h[ci][t])            (data gradient)
//   dlog-alpha2 / dlog-beta2 [ci] = sum (W2^T dy) * d snake2 / d log-alpha|beta
//   db1[ci]      = sum_{b,t} dh[ci][t]                                         (bias gradient of the k7 conv that produced h)
// Until round 5 these were four launches — sat_wgrad_small_bf16x3_kernel<1> (reads dy, h), sat_conv1d_bf16x3_kernel<4, 4> as the data
// gradient (reads dy, h; writes dh + its planes), sat_rowsum over dh — each streaming the same two (B, C, T) tensors from HBM: 32 bytes
// per element of the unit where 16 suffice (read dy, read h, write dh, write dh's planes), on kernels that are HBM-bound at the C = 128
// levels (3.5 - 3.8 TB/s; profiles/r04_vae_train_kernel_stats.csv: 23 ms of the 148-ms generator step).  Here one workgroup walks a
// contiguous range of 32-step time tiles with ALL of the unit's dy channels and 128 of its h channels:
//   * the tile of dy is converted once (bf16 hi / lo split, three MFMAs per product: fp32-class accuracy as everywhere on this path) into
//     a natural [co][t] image (A operand of the weight gradient: 8 consecutive t per lane) AND a transposed [t][co] image (B operand of
//     the data gradient: 8 consecutive co per lane); the tile of h becomes snake2(h) planes [ci][t] (B operand of the weight gradient);
//   * data gradient (K = C over co) and weight gradient (K = 32 steps, accumulated in registers over the workgroup's whole range) run
//     on v_mfma_f32_32x32x16_bf16; W2^T fragments come from L2 (a [ci][co] plane pair prepared once per step by sat_ru_k1_pack);
//   * the epilogue runs in the STAGING layout (a thread owns four consecutive steps of a channel, its h values still in registers):
//     dsnake2, the three per-channel sums (kept in registers across tiles: a thread's channels never change), 16-byte stores of dh, and
//     — through a transposed LDS image — dh's activation planes for the k7 data-gradient that consumes it next (conv1d_planes.h layout).
// Work per tile (C = 128): 65.5 KB of HBM traffic, 48 MFMAs per wave; two workgroups per CU (58 KB of LDS each).  Served: C = 128 (the
// two widest levels of encoder and decoder: 12 of the 30 units, 78 % of the stack's activation bytes); other widths keep the separate
// kernels (at C = 256 a wave's W2^T fragments no longer fit the register file beside its 128 accumulator registers).
#include "conv_common.h"

typedef uint32_t sat_u32x2 __attribute__((ext_vector_type(2)));

#define SAT_RK_TT 32                 // time steps per tile
#define SAT_RK_ROWN (SAT_RK_TT + 8)  // shorts per row of a natural [channel][t] image (80 B: conflict-free 16-byte fragment reads)
#define SAT_RK_EROW (SAT_RK_TT + 4)  // floats per row of the epilogue tile
#define SAT_RK_LEAD 32               // zero rows before t = 0 in an activation plane (SAT_K7P_LEAD of conv1d_planes.h)

// four 4-dword vectors <-> one 16-register block (dwords 4 g + e = v_g[e]): pure register naming, no copies
SAT_DEVICE f32x16 sat_rk_cat4(f32x4 v0, f32x4 v1, f32x4 v2, f32x4 v3) {
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
SAT_DEVICE bf16x8 sat_rk_frag(const f32x16& v, int g) {      // g is a compile-time constant after unrolling
    const f32x4 q = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
    return __builtin_bit_cast(bf16x8, q);
}

struct SatRuK1BwdParams {
    const float* dy;      // (B, C, T)
    const float* h;       // (B, C, T)
    const short* wt_hi;   // W2^T planes [ci][co]
    const short* wt_lo;
    const float* alpha2;  // (C) log-alpha / log-beta of snake2
    const float* beta2;
    float* dh;            // (B, C, T)
    short* em_hi;         // dh as activation planes [B][C/8][em_rows][8] (row 32 + t), or null
    short* em_lo;
    float* dw_partial;    // [nsplit][C (co)][C (ci)]
    float* part;          // [4][C][nsplit]: d log-alpha2, d log-beta2, sum dh, sum dy
    int B, C, T, em_rows, nsplit, tiles_per_split, ntiles;
};

// Eight waves, ONE workgroup per CU.  Staging, epilogue and emission are shared by all 512 threads (a thread owns four consecutive steps
// of channels (tid >> 3) and (tid >> 3) + 64); the matrix work is split by ROLE: waves 0-3 run the data gradient (wave w: ci rows
// 32 w .. + 32 of the tile, K = 128 over co; its W2^T fragments — 32 rows x 128 co, hi + lo = 64 registers — stay resident: fetched per
// k-step from L2 they serialised the loop behind sixteen dependent round trips per tile, the first timing of this kernel), waves 4-7
// the weight gradient (a 64 x 64 block of dW2 each, accumulated over the workgroup's whole range: 64 registers).  Waves w and w + 4
// share a SIMD, so every SIMD carries one wave of each role: 24 MFMAs per wave and tile on both.
// The tiles of dy and h travel global -> LDS by LDS-DMA (32 one-KiB pieces per tile, four per wave) into a two-slot ring of raw fp32
// images, one whole tile ahead, behind COUNTED waits and bare barriers: gfx950's vmcnt is one in-order counter for loads and stores, so
// a register prefetch issued a tile ahead is waited for together with the stores between (measured: that version ran at 3.3 TB/s).
template <int C>
__global__ void __launch_bounds__(512) sat_ru_k1_bwd_kernel(SatRuK1BwdParams p) {
    static_assert(C == 128, "one 128-channel tile of h per workgroup, W2^T fragments in registers");
    constexpr int TT = SAT_RK_TT, ROWN = SAT_RK_ROWN, ROWT = C + 8, EROW = SAT_RK_EROW;
    constexpr int NJ = C / 64;                   // float4 slots of a (C x 32) tile per thread (C rows x 8 slots / 512 threads)
    constexpr int DYT = 2 * TT * ROWT;           // shorts: transposed dy image, hi + lo   (also the transposed dh image of the emission)
    constexpr int DYN = 2 * C * ROWN;            // natural dy image
    constexpr int AN = 2 * C * ROWN;             // natural snake2(h) image   (also the fp32 epilogue tile)
    constexpr int RAW = 2 * C * TT;              // floats: one raw tile = dy rows 0 .. C-1, h rows C .. 2C-1, 32 steps (128 B) each
    static_assert(C * EROW * 4 <= AN * 2, "the epilogue tile aliases the snake2(h) image");
    static_assert(TT * 4 == 128, "a raw row is eight 16-byte DMA lanes");
    // ONE shared object (a second one makes hipcc drain the DMA queue before every LDS read: cdna_hip_programming.md section 5)
    __shared__ __attribute__((aligned(16))) char smem[(DYT + DYN + AN) * 2 + 2 * RAW * 4];
    short* dyT = reinterpret_cast<short*>(smem);  // [plane][t][ROWT]
    short* dyN = dyT + DYT;                      // [plane][co][ROWN]
    short* aN = dyN + DYN;                       // [plane][ci][ROWN]
    float* etile = reinterpret_cast<float*>(aN); // [ci][EROW]
    short* dhT = dyT;                            // [plane][t][ROWT]: dh transposed, for the emission
    float* raw = reinterpret_cast<float*>(smem + (DYT + DYN + AN) * 2);      // [slot][2C][32]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = SAT_UNIFORM((int)(threadIdx.x >> 6));
    const int l31 = lane & 31, hi = lane >> 5;
    const int split = blockIdx.x;
    const int tiles_per_b = p.T / TT;
    const int tile_begin = split * p.tiles_per_split;
    int tile_end = tile_begin + p.tiles_per_split;
    if (tile_end > p.ntiles) tile_end = p.ntiles;
    const bool dgrad_wave = wave < 4;            // wave-uniform role
    const int wq = wave & 3, wm = wq >> 1, wn = wq & 1;

    // this thread's slots of a tile: channels (tid >> 3) + 64 j, steps 4 (tid & 7) .. + 3
    const int srow = tid >> 3, st4 = (tid & 7) * 4;
    // Transposed images [t][channel]: the 16-byte channel groups of row t are stored at group ^ ((t >> 3) & 3).  A wave's 2-byte
    // transposed stores go to rows 0, 4, .. 28 of the same channel: at a row stride of 272 bytes those are two banks without the
    // swizzle (4-way conflicts on every one of the 32 stores per thread and tile — the first versions of this kernel spent a third of
    // their time there) and eight with it; the 16-byte reads (fragments, plane emission) apply the same XOR.
    const int swz_w = ((st4 >> 3) & 3) << 3;      // writer: rows st4 .. st4 + 3 share (t >> 3)
    const int swz_f = ((l31 >> 3) & 3) << 3;      // fragment reader: row l31
    float sa[NJ], sb[NJ], sib[NJ];               // snake2 constants of the thread's h channels
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int ci = srow + 64 * j;
        sa[j] = expf(p.alpha2[ci]);
        sb[j] = expf(p.beta2[ci]);
        sib[j] = 1.0f / (sb[j] + 1e-9f);
    }
    float sum_da[NJ], sum_db[NJ], sum_dh[NJ], sum_dy[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) sum_da[j] = sum_db[j] = sum_dh[j] = sum_dy[j] = 0.0f;

    // 64 registers with two meanings (the roles never mix inside a wave, and the register allocator would otherwise keep BOTH the
    // accumulators and the fragments live in every wave — that spilled 48 registers into the tile loop):
    //   weight-gradient waves: big[2 mb + nb] = the dW2 accumulator of rows (co) 64 wm + 32 mb, columns (ci) 64 wn + 32 nb;
    //   data-gradient waves:   W2^T fragments, rows 32 wq + l31, k = co: k-step s of the hi plane = dwords 4 (s & 3) .. + 3 of
    //                          big[s >> 2], of the lo plane of big[2 + (s >> 2)].
    f32x16 big[4];
    if (dgrad_wave) {
        const short* wrow_hi = p.wt_hi + (size_t)(32 * wq + l31) * C + 8 * hi;
        const short* wrow_lo = p.wt_lo + (size_t)(32 * wq + l31) * C + 8 * hi;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            big[q] = sat_rk_cat4(*reinterpret_cast<const f32x4*>(wrow_hi + 64 * q), *reinterpret_cast<const f32x4*>(wrow_hi + 64 * q + 16),
                                 *reinterpret_cast<const f32x4*>(wrow_hi + 64 * q + 32), *reinterpret_cast<const f32x4*>(wrow_hi + 64 * q + 48));
            big[2 + q] = sat_rk_cat4(*reinterpret_cast<const f32x4*>(wrow_lo + 64 * q), *reinterpret_cast<const f32x4*>(wrow_lo + 64 * q + 16),
                                     *reinterpret_cast<const f32x4*>(wrow_lo + 64 * q + 32), *reinterpret_cast<const f32x4*>(wrow_lo + 64 * q + 48));
        }
        // (the fragments are in registers before the first LDS-DMA goes out: no ordinary load is pending while one is in flight)
#if !defined(SAT_HIPEMU)
        asm volatile("" : "+v"(big[0]), "+v"(big[1]), "+v"(big[2]), "+v"(big[3]));
#endif
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) big[i][r] = 0.0f;
    }

    // LDS-DMA of a tile: piece = wave + 8 q covers raw rows 8 piece .. + 7 (q < 2: dy channels, q >= 2: h channels); lane -> (row, 16-byte slot)
    const size_t lane_off = (size_t)(lane >> 3) * p.T + (lane & 7) * 4;
    auto dma_tile = [&](int tile) __attribute__((always_inline)) {
        const int b = tile / tiles_per_b, t0 = (tile - b * tiles_per_b) * TT;
        char* slot = reinterpret_cast<char*>(raw) + (tile & 1) * (RAW * 4);
        const size_t base = (size_t)b * C * p.T + t0 + lane_off;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int piece = wave + 8 * q;                       // rows 8 piece ..: channels 8 (piece & 15) .. of dy (q < 2) or h
            const float* src = (q < 2 ? p.dy : p.h) + base + (size_t)(8 * (piece & 15)) * p.T;
            sat_glds16(src, slot + piece * 1024);
        }
    };
    if (tile_begin < tile_end) dma_tile(tile_begin);

    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const int b = tile / tiles_per_b, t0 = (tile - b * tiles_per_b) * TT;
        const float* rt = raw + (tile & 1) * RAW;
        // the next tile's pieces go out first (their slot was last read in the previous tile's epilogue, a barrier ago); then this wave's
        // pieces of THIS tile are waited for: everything it issued since — the previous tile's stores and the four new pieces — is younger
        const bool more = tile + 1 < tile_end;
        if (more) { dma_tile(tile + 1); SAT_WAIT_VMCNT(4); } else { SAT_WAIT_VMCNT(0); }
        SAT_RAW_BARRIER();
        // ---- P1: split dy (natural + transposed images), snake2(h) (natural image) ----
        f32x4 hv[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int ch = srow + 64 * j;
            const f32x4 dyv = *reinterpret_cast<const f32x4*>(rt + ch * TT + st4);
            hv[j] = *reinterpret_cast<const f32x4*>(rt + (C + ch) * TT + st4);
            uint32_t h0, l0, h1, l1;
            sat_split2_pk(dyv[0], dyv[1], &h0, &l0);
            sat_split2_pk(dyv[2], dyv[3], &h1, &l1);
            *reinterpret_cast<sat_u32x2*>(dyN + ch * ROWN + st4) = sat_u32x2{h0, h1};
            *reinterpret_cast<sat_u32x2*>(dyN + C * ROWN + ch * ROWN + st4) = sat_u32x2{l0, l1};
            const int cs = ch ^ swz_w;
            dyT[(st4 + 0) * ROWT + cs] = (short)(h0 & 0xffffu);
            dyT[(st4 + 1) * ROWT + cs] = (short)(h0 >> 16);
            dyT[(st4 + 2) * ROWT + cs] = (short)(h1 & 0xffffu);
            dyT[(st4 + 3) * ROWT + cs] = (short)(h1 >> 16);
            dyT[TT * ROWT + (st4 + 0) * ROWT + cs] = (short)(l0 & 0xffffu);
            dyT[TT * ROWT + (st4 + 1) * ROWT + cs] = (short)(l0 >> 16);
            dyT[TT * ROWT + (st4 + 2) * ROWT + cs] = (short)(l1 & 0xffffu);
            dyT[TT * ROWT + (st4 + 3) * ROWT + cs] = (short)(l1 >> 16);
            sum_dy[j] += (dyv[0] + dyv[1]) + (dyv[2] + dyv[3]);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = sat_snake(hv[j][e], sa[j], sib[j]);
            sat_split2_pk(v[0], v[1], &h0, &l0);
            sat_split2_pk(v[2], v[3], &h1, &l1);
            *reinterpret_cast<sat_u32x2*>(aN + ch * ROWN + st4) = sat_u32x2{h0, h1};
            *reinterpret_cast<sat_u32x2*>(aN + C * ROWN + ch * ROWN + st4) = sat_u32x2{l0, l1};
        }
        SAT_WAIT_LGKM0();
        SAT_RAW_BARRIER();

        // ---- P2: the matrix work, by role ----
        if (dgrad_wave) {
            f32x16 accd;
#pragma unroll
            for (int r = 0; r < 16; ++r) accd[r] = 0.0f;
            const short* brow = dyT + l31 * ROWT;
#pragma unroll
            for (int s = 0; s < C / 16; ++s) {
                const int col = (16 * s + 8 * hi) ^ swz_f;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(brow + col);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(brow + TT * ROWT + col);
                const bf16x8 wh = sat_rk_frag(big[s >> 2], s & 3), wl = sat_rk_frag(big[2 + (s >> 2)], s & 3);
                accd = sat_mfma_32x32x16_bf16(wh, bh, accd);
                accd = sat_mfma_32x32x16_bf16(wh, bl, accd);
                accd = sat_mfma_32x32x16_bf16(wl, bh, accd);
            }
            SAT_WAIT_LGKM0();
            SAT_RAW_BARRIER();                    // every fragment read of the tile is done (both roles): the images may be overwritten
            // ---- P3: W2^T dy from the accumulator layout into the epilogue tile [ci][t] ----
#pragma unroll
            for (int r = 0; r < 16; ++r) etile[(32 * wq + (r & 3) + 8 * (r >> 2) + 4 * hi) * EROW + l31] = accd[r];
        } else {
#pragma unroll
            for (int s = 0; s < TT / 16; ++s) {
                bf16x8 bh[2], bl[2];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const short* br = aN + (wn * 64 + 32 * nb + l31) * ROWN + 16 * s + 8 * hi;
                    bh[nb] = *reinterpret_cast<const bf16x8*>(br);
                    bl[nb] = *reinterpret_cast<const bf16x8*>(br + C * ROWN);
                }
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const short* ar = dyN + (wm * 64 + 32 * mb + l31) * ROWN + 16 * s + 8 * hi;
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ar);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(ar + C * ROWN);
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        big[2 * mb + nb] = sat_mfma_32x32x16_bf16(ah, bh[nb], big[2 * mb + nb]);
                        big[2 * mb + nb] = sat_mfma_32x32x16_bf16(ah, bl[nb], big[2 * mb + nb]);
                        big[2 * mb + nb] = sat_mfma_32x32x16_bf16(al, bh[nb], big[2 * mb + nb]);
                    }
                }
            }
            SAT_WAIT_LGKM0();
            SAT_RAW_BARRIER();                    // (pairs with the data-gradient waves' barrier)
        }
        SAT_WAIT_LGKM0();
        SAT_RAW_BARRIER();

        // ---- P4: dsnake2, the per-channel sums, dh (16-byte stores) and its transposed bf16 image ----
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int ch = srow + 64 * j;
            const f32x4 pre = *reinterpret_cast<const f32x4*>(etile + ch * EROW + st4);
            f32x4 out;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const SatSnakeGrad g = sat_snake_grad(hv[j][e], sa[j], sb[j]);
                sum_da[j] += pre[e] * g.dla;
                sum_db[j] += pre[e] * g.dlb;
                out[e] = pre[e] * g.dx;
                sum_dh[j] += out[e];
            }
            *reinterpret_cast<f32x4*>(p.dh + ((size_t)b * C + ch) * p.T + t0 + st4) = out;
            if (p.em_hi) {
                uint32_t h0, l0, h1, l1;
                sat_split2_pk(out[0], out[1], &h0, &l0);
                sat_split2_pk(out[2], out[3], &h1, &l1);
                const int cs = ch ^ swz_w;
                dhT[(st4 + 0) * ROWT + cs] = (short)(h0 & 0xffffu);
                dhT[(st4 + 1) * ROWT + cs] = (short)(h0 >> 16);
                dhT[(st4 + 2) * ROWT + cs] = (short)(h1 & 0xffffu);
                dhT[(st4 + 3) * ROWT + cs] = (short)(h1 >> 16);
                dhT[TT * ROWT + (st4 + 0) * ROWT + cs] = (short)(l0 & 0xffffu);
                dhT[TT * ROWT + (st4 + 1) * ROWT + cs] = (short)(l0 >> 16);
                dhT[TT * ROWT + (st4 + 2) * ROWT + cs] = (short)(l1 & 0xffffu);
                dhT[TT * ROWT + (st4 + 3) * ROWT + cs] = (short)(l1 >> 16);
            }
        }
        SAT_WAIT_LGKM0();
        SAT_RAW_BARRIER();

        // ---- P5: plane emission: rows of 8 channels at one step, 16 bytes each (consecutive threads = consecutive steps) ----
        if (p.em_hi) {
            const int t = tid & 31, g = tid >> 5;                              // g: 8-channel group (0..15)
            const size_t o = (((size_t)b * (C / 8) + g) * p.em_rows + SAT_RK_LEAD + t0 + t) * 8;
            const int gs = (8 * g) ^ (((t >> 3) & 3) << 3);
            *reinterpret_cast<u32x4*>(p.em_hi + o) = *reinterpret_cast<const u32x4*>(dhT + t * ROWT + gs);
            *reinterpret_cast<u32x4*>(p.em_lo + o) = *reinterpret_cast<const u32x4*>(dhT + TT * ROWT + t * ROWT + gs);
            SAT_WAIT_LGKM0();
            SAT_RAW_BARRIER();                    // the transposed image aliases the next tile's transposed dy image
        }
    }

    // ---- the workgroup's dW2 slab and its per-channel sums ----
    if (!dgrad_wave) {
        float* slab = p.dw_partial + (size_t)split * C * C;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = wm * 64 + 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const int ci = wn * 64 + 32 * nb + l31;
                    slab[(size_t)co * C + ci] = big[2 * mb + nb][r];
                }
    }
    const size_t ps = (size_t)C * p.nsplit;      // one of the four partial planes [C][nsplit]
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float va = sum_da[j], vb = sum_db[j], vh = sum_dh[j], vy = sum_dy[j];
#pragma unroll
        for (int m = 1; m <= 4; m <<= 1) {       // the eight lanes that share a channel are consecutive
            va += __shfl_xor(va, m);
            vb += __shfl_xor(vb, m);
            vh += __shfl_xor(vh, m);
            vy += __shfl_xor(vy, m);
        }
        if ((tid & 7) == 0) {
            const size_t o = (size_t)(srow + 64 * j) * p.nsplit + split;
            p.part[o] = va;
            p.part[ps + o] = vb;
            p.part[2 * ps + o] = vh;
            p.part[3 * ps + o] = vy;
        }
    }
}

static int sat_ru_k1_plan(int B, int C, int T, int* nsplit, int* tiles_per_split, int* ntiles) {
    if (B <= 0 || T <= 0 || C != 128 || (T % SAT_RK_TT) != 0) return 1;      // (the kernel template also instantiates for C = 256: its
                                                                                // W2^T fragments do not fit the register file — not served)
    const long long nt = (long long)B * (T / SAT_RK_TT);
    if (nt > 0x7fffffff) return 1;
    int want = sat_cu_count();                   // one eight-wave workgroup per CU
    if (want > nt) want = (int)nt;
    const int per = (int)((nt + want - 1) / want);
    *tiles_per_split = per;
    *nsplit = (int)((nt + per - 1) / per);
    *ntiles = (int)nt;
    return 0;
}

// number of dW2 slabs / columns of the partial-sum planes for this shape; -1: the shape is not served (the caller keeps the separate
// weight-gradient / data-gradient / row-sum launches)
extern "C" int sat_ru_k1_bwd_nsplit(int B, int C, int T) {
    int ns, per, nt;
    if (sat_ru_k1_plan(B, C, T, &ns, &per, &nt)) return -1;
    return ns;
}

// See the header of this file.  wt_hi / wt_lo: sat_ru_k1_pack(W2).  alpha2 / beta2: snake2's log parameters (C).  dh (B, C, T) out.
// em_hi / em_lo (optional): dh as activation planes [B][C/8][em_rows][8], row 32 + t (rows around the sequence stay as the caller
// zeroed them).  dw_partial [nsplit][C][C] (sum with sat_reduce_splits: torch layout (Cout, Cin, 1)); part [4][C][nsplit] (sum the last
// axis: d log-alpha2, d log-beta2, bias gradient of the k7 conv = sum dh, bias gradient of the k1 conv = sum dy).
extern "C" int sat_ru_k1_bwd(const float* dy, const float* h, const short* wt_hi, const short* wt_lo, const float* alpha2,
                             const float* beta2, float* dh, short* em_hi, short* em_lo, int em_rows, float* dw_partial, float* part,
                             int B, int C, int T, void* stream) {
    SatRuK1BwdParams p{dy, h, wt_hi, wt_lo, alpha2, beta2, dh, em_hi, em_lo, dw_partial, part, B, C, T, em_rows, 0, 0, 0};
    if (sat_ru_k1_plan(B, C, T, &p.nsplit, &p.tiles_per_split, &p.ntiles)) {
        sat_set_error("sat_ru_k1_bwd: needs C == 128 and T % 32 == 0 (sat_ru_k1_bwd_nsplit)");
        return 1;
    }
    if (!dy || !h || !wt_hi || !wt_lo || !alpha2 || !beta2 || !dh || !dw_partial || !part) { sat_set_error("sat_ru_k1_bwd: missing buffer"); return 1; }
    if ((em_hi == nullptr) != (em_lo == nullptr) || (em_hi && em_rows < SAT_RK_LEAD + T)) { sat_set_error("sat_ru_k1_bwd: bad emission planes"); return 1; }
    if ((((uintptr_t)dy | (uintptr_t)h | (uintptr_t)dh | (uintptr_t)em_hi | (uintptr_t)em_lo | (uintptr_t)wt_hi | (uintptr_t)wt_lo) & 15) != 0) {
        sat_set_error("sat_ru_k1_bwd: buffers must be 16-byte aligned");
        return 1;
    }
    SAT_LAUNCH(sat_ru_k1_bwd_kernel<128>, dim3(p.nsplit), dim3(512), stream, p);
    return sat_check_launch("sat_ru_k1_bwd");
}

// W2 (C, C) fp32 [co][ci]  ->  W2^T as bf16 hi / lo planes [ci][co]
struct SatRuK1PackParams {
    const float* w;
    short* hi;
    short* lo;
    int C;
};
__global__ void __launch_bounds__(256) sat_ru_k1_pack_kernel(SatRuK1PackParams p) {
    const int i = blockIdx.x * 256 + threadIdx.x;      // output element ci * C + co
    if (i >= p.C * p.C) return;
    const int ci = i / p.C, co = i - ci * p.C;
    const float v = p.w[(size_t)co * p.C + ci];
    const short hb = sat_f32_to_bf16(v);
    p.hi[i] = hb;
    p.lo[i] = sat_f32_to_bf16(v - sat_bf16_to_f32(hb));
}
extern "C" int sat_ru_k1_pack(const float* w, short* hi, short* lo, int C, void* stream) {
    if (C <= 0 || !w || !hi || !lo) { sat_set_error("sat_ru_k1_pack: bad arguments"); return 1; }
    SatRuK1PackParams p{w, hi, lo, C};
    SAT_LAUNCH(sat_ru_k1_pack_kernel, dim3(sat_cdiv(C * C, 256)), dim3(256), stream, p);
    return sat_check_launch("sat_ru_k1_pack");
}
