// conv1d_bf16x3_k7q.h — the k = 5..7 stride-1 (dilated) convolutions of the ResidualUnits (autoencoders.py:58-83) and their
// data-gradients from pre-split activation planes, third generation.  Included by conv1d_bf16x3.hip.
//
// conv1d_bf16x3_k7p.h runs its eight waves in lock-step (fragment reads and MFMAs of both waves of a SIMD at the same time: matrix
// pipe 47 % busy at C = 128) and spends one k-slot in eight on a zero tap (8 tap groups for 7 taps).  Here:
//   * a K-chunk is 16 input channels x the K taps: MFMA k-step tau = tap tau, k-slots 0-7 = channels 0-7 of the chunk (lanes 0-31),
//     8-15 = channels 8-15 (lanes 32-63) — 7 k-steps for 7 taps, no padding.  Weights are packed for it by sat_pack_weights_k7q as
//     [chunk][tap][8-channel group][co][8]: the A fragment of (tap, group) is 16 bytes per lane at consecutive co rows (conflict
//     free, no swizzle), and a stage's weight slab is a lane-linear copy (56 one-KiB LDS-DMA pieces of the 128-row channel tile).
//     Activations: the planes of conv1d_bf16x3_k7p.h ([B][Cin/8][rows][8]); a stage holds [plane][group][320 rows][8] (20 pieces).
//   * the two wave rows (wr = wave >> 2: 64 of the 128 output channels each; wc = wave & 3: 64 of the 256 time steps) run ONE
//     BARRIER APART, as in gemm.hip's 256 x 256 kernel: a chunk is two phases (taps 0-3: 48 MFMAs per wave, taps 4..K-1: 36), each
//     [read section: the phase's fragment reads (8 ds_read_b128 per tap) + half of the NEXT chunk's LDS-DMA, lgkmcnt(0)] s_barrier
//     [MFMAs at raised priority] s_barrier; in every interval one wave of each SIMD multiplies while its partner reads / issues DMA.
//     Two stages of 76 KiB; the wait for chunk c+1 (vmcnt(0)) sits in front of the middle barrier of chunk c's second phase: the
//     first read of chunk c+1 is two barriers later for the waiting wave and one barrier after the other wave row's wait; a stage
//     is overwritten from the phase after the barrier that retired its last reads.
//   * epilogue: as conv1d_bf16x3_k7.h (bias, dsnake + its per-channel sums, residual, tanh; 16-byte accesses through an LDS
//     transposition that reuses the drained stage memory).
//   * round 6 (VARIANT 3, the shipped K loop): the next chunk's LDS-DMA is issued INSIDE the MFMA sections, not beside the fragment reads
//     (the item above describes VARIANT 1, kept as the A / B arm: SatConvBfLaunch::dma_in_mfma, ops.k7q_dma_in_mfma) — see INMFMA below.
#pragma once

#define SAT_K7Q_TAPS 7
#define SAT_K7Q_WBYTES (2 * SAT_K7Q_TAPS * 2 * 2048)        // [plane][tap][group][128 co][16 B]
#define SAT_K7Q_ABYTES (2 * 2 * SAT_K7_AROWS * 16)           // [plane][group][320 rows][16 B]
#define SAT_K7Q_STAGE (SAT_K7Q_WBYTES + SAT_K7Q_ABYTES)
#define SAT_K7Q_WPIECES (2 * SAT_K7Q_TAPS * 2 * 2)           // 56
#define SAT_K7Q_APIECES (2 * 2 * (SAT_K7_AROWS / 64))        // 20
#define SAT_K7Q_PIECES (SAT_K7Q_WPIECES + SAT_K7Q_APIECES)   // 76

// weight preparation for this kernel: torch weight w[D0][D1][K] (fp32) -> hi / lo bf16 planes [chunk c16][tap][group g][m (padded to
// 128)][e] holding W'[m][v][tap], v = (c16 * 2 + g) * 8 + e (0 where m / v are out of range):
//   mode 0 (conv, w = [out][in][K]): W' = w[m][v][tap];   mode 1 (data-gradient of a stride-1 conv): W'[m][v][tap] = w[v][m][K-1-tap]
struct SatPackQParams {
    const float* w;
    short* hi;
    short* lo;
    int D0, D1, K, mode, m_v, v_v, out_pad;
    long long total;
};
__global__ void __launch_bounds__(256) sat_pack_k7q_kernel(SatPackQParams p) {
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= p.total) return;
    const int e = (int)(o & 7);
    long long q = o >> 3;
    const int m = (int)(q % p.out_pad);
    q /= p.out_pad;
    const int g = (int)(q & 1);
    q >>= 1;
    const int tap = (int)(q % p.K), chunk = (int)(q / p.K);
    const int v = (chunk * 2 + g) * 8 + e;
    float val = 0.0f;
    if (m < p.m_v && v < p.v_v) {
        if (p.mode == 0) val = p.w[((size_t)m * p.D1 + v) * p.K + tap];
        else val = p.w[((size_t)v * p.D1 + m) * p.K + (p.K - 1 - tap)];
    }
    short h, l;
    sat_split2(val, &h, &l);
    p.hi[o] = h;
    p.lo[o] = l;
}
static bool sat_pack_q_geometry(int D0, int D1, int K, int mode, SatPackQParams* p) {
    if (mode < 0 || mode > 1 || D0 <= 0 || D1 <= 0 || !(K == 1 || (K >= 5 && K <= SAT_K7Q_TAPS))) return false;
    p->D0 = D0; p->D1 = D1; p->K = K; p->mode = mode;
    if (mode == 0) { p->m_v = D0; p->v_v = D1; } else { p->m_v = D1; p->v_v = D0; }
    p->out_pad = sat_cdiv(p->m_v, SAT_K7_CO) * SAT_K7_CO;
    p->total = (long long)sat_cdiv(p->v_v, 16) * K * 2 * p->out_pad * 8;
    return true;
}
extern "C" long long sat_pack_weights_k7q_size(int D0, int D1, int K, int mode) {
    SatPackQParams p{};
    return sat_pack_q_geometry(D0, D1, K, mode, &p) ? p.total : -1;
}
extern "C" int sat_pack_weights_k7q(const float* w, short* hi, short* lo, int D0, int D1, int K, int mode, void* stream) {
    SatPackQParams p{};
    if (!sat_pack_q_geometry(D0, D1, K, mode, &p)) { sat_set_error("sat_pack_weights_k7q: needs 5 <= K <= 7 (or K == 1: the fused unit's 1x1 conv), mode 0|1"); return 1; }
    p.w = w; p.hi = hi; p.lo = lo;
    SAT_LAUNCH(sat_pack_k7q_kernel, dim3((unsigned)sat_cdivll(p.total, 256)), dim3(256), stream, p);
    return sat_check_launch("sat_pack_weights_k7q");
}

// FUSED (template): the whole ResidualUnit forward, y = x + conv1(snake2(conv7(snake1(x)))) (autoencoders.py:58-83), for C <= 128
// (one channel tile holds every channel of the intermediate h): after the k7 K loop the workgroup owns h[:, t0 .. t0 + 256) in its
// accumulators — it stores h (fp32, kept for the backward), writes snake2(h + bias1) as bf16 hi / lo planes into the drained stage
// memory ([plane][16 channel groups][256 rows][8]: straight from the accumulator layout, 8-byte ds_writes, no transposition), runs the
// 1x1 conv as a second GEMM on them (K = 128: 8 MFMA k-steps of 16 channels, 96 MFMAs per wave; its weight fragments come from L2
// straight into registers) and finishes with the ordinary epilogue on the second accumulators (bias2, residual x, plane emission for
// the next unit).  The k1 launch, its read of h and its separate activation pass disappear.
typedef uint32_t u32x2_q __attribute__((ext_vector_type(2)));
// PERSIST (round 6): one workgroup per CU walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...  Every chunk of a tile lives in the stage of
// the OPPOSITE parity (chunk 0 in stage 1) — the half of the LDS the epilogue's transposition windows (offset 0, 70 KB) do not touch: the
// NEXT tile's chunk 0 is requested (76 LDS-DMA pieces, nothing in registers) right after the K loop and lands under the epilogue,
// instead of a fresh workgroup's prologue waiting for it with an idle CU (~2-3 us of the ~51 us a C = 128 tile takes).
template <int VARIANT = 1, bool FUSED = false, bool PERSIST = false>
__global__ void __launch_bounds__(SAT_K7_NT) sat_conv1d_bf16x3_k7q_kernel(SatConvBfLaunch a) {
    static_assert(!(FUSED && PERSIST), "the fused unit is not persistent");
    // VARIANT 1 (shipped): the next chunk's LDS-DMA goes out longest latency first — phase 0 issues its activation pieces (HBM) and
    // taps 0, 1, phase 1 taps 2..6 (L2-resident weights) — with COUNTED waits: phase 1 leaves taps 4-6 in flight (vmcnt(3)), they are
    // retired by the next chunk's phase 0 (which leaves its own new pieces in flight: 5 for waves 0-3, 4 for waves 4-7) one phase
    // before they are read.  VARIANT 0: weights first, vmcnt(0) once per chunk (3-4 % slower at C = 128 / 256, profiles/EXPERIMENTS.md).
    constexpr bool REORDER = (VARIANT & 1) != 0;
    // VARIANT 3 (round 6): the next chunk's LDS-DMA is issued INSIDE the MFMA sections — a piece costs 100-185 cycles of issue in a read section that
    // also carries 32 ds_read_b128 and ~60 between MFMAs (MI355X_MICROARCH.md), and an ablation put the DMA at 21-24 % of the launch
    // (profiles/r06_experiments/k7q_ablation/).  Phase 0 issues [activations, taps 0-3] of chunk c+1 after its taps' MFMAs, phase 1 [taps 4-6];
    // each read section then waits vmcnt(0) for what the PREVIOUS MFMA section issued: phase 1's wait retires [activations, taps 0-3] of c+1 (first
    // read two barriers later by the first wave row, three by the second), phase 0's wait retires [taps 4-6] of its own chunk (first read
    // after the phase's closing barrier).  Both wave rows' waits precede, by a barrier, every read of what they publish.
    constexpr bool INMFMA = VARIANT == 3;
    constexpr int CO_T = SAT_K7_CO, T_T = SAT_K7_T, NT = SAT_K7_NT, AROWS = SAT_K7_AROWS;
    constexpr int TW = T_T / 64;                          // waves along time
    const SatConvParams& p = a.p;
    // ONE LDS object (a second one makes hipcc drain the LDS-DMA queue in front of every fragment read: cdna_hip_programming.md §5)
    constexpr int RED_OFF = 2 * SAT_K7Q_STAGE, EP_OFF = RED_OFF + 2 * TW * CO_T * 4;
    __shared__ __attribute__((aligned(1024))) char lds[EP_OFF + 6 * CO_T * 4];
    float (*red_lds)[TW][CO_T] = reinterpret_cast<float (*)[TW][CO_T]>(lds + RED_OFF);
    float (*ep_lds)[CO_T] = reinterpret_cast<float (*)[CO_T]>(lds + EP_OFF);

    // (PERSIST re-derives the lane- / wave-dependent values per tile from laundered copies: as loop invariants of the tile loop they
    // — and everything the epilogue computes from them — would stay live across the K loop and the epilogue alike: 171 spilled registers)
    int tid = threadIdx.x;
    int lane = tid & 63;
    int wave = SAT_UNIFORM(tid >> 6);
    int l31 = lane & 31, hi = lane >> 5;
    int wr = wave / TW;
    int co_w = wr * 64, t_w = (wave % TW) * 64;
    const int K = p.K, dil = p.dil;
    const int nchunks = (a.cin_v + 15) / 16;
    const int co_tiles = a.cout_pad / CO_T, t_tiles = (a.nq + T_T - 1) / T_T;
    const int total_tiles = co_tiles * t_tiles * p.B;
    int tile_id = (int)blockIdx.x;
    int co_tile, win;
    sat_xcd_tile(tile_id, co_tiles, t_tiles * p.B, &co_tile, &win);
    int b = win / t_tiles, t_tile = win - b * t_tiles;
    int co0 = co_tile * CO_T, t0 = t_tile * T_T;
    int row_in0 = SAT_K7P_LEAD + t0 - p.pad;              // plane row of the window's first input step (>= 0: pad <= LEAD)
    constexpr int SPAR = PERSIST ? 1 : 0;                 // chunk c is staged in stage (c + SPAR) & 1

    auto load_consts = [&]() {
    if (tid < CO_T) {
        const int m = co0 + tid;
        const bool ok = m < a.cout_v;
        ep_lds[0][tid] = (ok && p.bias) ? p.bias[m] : 0.0f;
        ep_lds[1][tid] = (ok && p.x2) ? expf(p.alpha2[m]) : 1.0f;
        ep_lds[2][tid] = (ok && p.x2) ? expf(p.beta2[m]) : 1.0f;
        if constexpr (FUSED) {                            // (never a data-gradient: rows 1 / 2 carry the second SnakeBeta's constants)
            ep_lds[1][tid] = ok ? a.ru_a2[m] : 0.0f;
            ep_lds[2][tid] = ok ? a.ru_ib2[m] : 0.0f;
            ep_lds[5][tid] = (ok && a.ru_bias2) ? a.ru_bias2[m] : 0.0f;
        }
        ep_lds[3][tid] = (ok && a.em_a) ? a.em_a[m] : 0.0f;
        ep_lds[4][tid] = (ok && a.em_a) ? a.em_ib[m] : 0.0f;
    }
    };
    load_consts();

    // ---- LDS-DMA of a chunk: 56 weight pieces (per tap: 8 = plane x group x 64-row half, ONE per wave) + 20 activation pieces
    //      ((plane, group) x 5 x 64 rows: waves 0-7 twice, waves 0-3 a third).  Every source address is a WAVE-UNIFORM base
    //      (scalar registers) + lane * 16 bytes: no per-piece address registers to keep alive (a spilled address would be reloaded
    //      with a vmcnt(0) that drains the DMA queue) ----
    unsigned lane16 = (unsigned)lane * 16u;
    const int w_pl = wave >> 2, w_g = (wave >> 1) & 1, w_half = wave & 1;
    const char* w_src0 = (const char*)(w_pl ? a.w_lo : a.w_hi) + ((size_t)w_g * a.cout_pad + co0 + w_half * 64) * 16;
    const size_t w_tap_stride = (size_t)a.cout_pad * 32;                      // bytes between taps: [tap][2 groups][cout_pad][16 B]
    const int w_dst0 = w_pl * (SAT_K7Q_TAPS * 4096) + (w_g * 2 + w_half) * 1024;
    auto issue_w = [&](int c, int st, int tap) {                             // this wave's piece of tap `tap`
        if (tap < K) sat_glds16(w_src0 + ((size_t)c * K + tap) * w_tap_stride + lane16, lds + st * SAT_K7Q_STAGE + w_dst0 + tap * 4096);
    };
    auto issue_a = [&](int c, int st, int i) {                               // activation piece r = 8 i + wave (r < 20)
        const int r = 8 * i + wave;
        if (r < SAT_K7Q_APIECES) {
            const int pl = r / 10, g = (r / 5) & 1, sub = r % 5;
            int c8 = c * 2 + g;
            c8 = c8 < a.xp_c8 ? c8 : a.xp_c8 - 1;         // past the end: any finite rows (their weights are zero)
            const char* src = (const char*)(pl ? a.xp_lo : a.xp_hi) + (((size_t)b * a.xp_c8 + c8) * a.xp_rows + row_in0 + sub * 64) * 16;
            sat_glds16(src + lane16, lds + st * SAT_K7Q_STAGE + SAT_K7Q_WBYTES + r * 1024);
        }
    };

    f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    };
    zero_acc();

    struct Frags { bf16x8 wa[2][2], xa[2][2]; };          // [mi | ni][plane]
    Frags fr[4];
    auto load_frags = [&](Frags& f, const char* sb, int tap) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const char* wb = sb + ((pl * SAT_K7Q_TAPS + tap) * 2 + hi) * 2048;
            f.wa[0][pl] = *reinterpret_cast<const bf16x8*>(wb + (co_w + l31) * 16);
            f.wa[1][pl] = *reinterpret_cast<const bf16x8*>(wb + (co_w + 32 + l31) * 16);
            const char* ab = sb + SAT_K7Q_WBYTES + (pl * 2 + hi) * (AROWS * 16);
            f.xa[0][pl] = *reinterpret_cast<const bf16x8*>(ab + (t_w + l31 + tap * dil) * 16);
            f.xa[1][pl] = *reinterpret_cast<const bf16x8*>(ab + (t_w + 32 + l31 + tap * dil) * 16);
        }
    };
    auto mfma_frags = [&](const Frags& f) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = sat_mfma_32x32x16_bf16(f.wa[mi][0], f.xa[ni][0], acc[mi][ni]);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = sat_mfma_32x32x16_bf16(f.wa[mi][0], f.xa[ni][1], acc[mi][ni]);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = sat_mfma_32x32x16_bf16(f.wa[mi][1], f.xa[ni][0], acc[mi][ni]);
    };

    bool wave_on_co = (co0 + co_w) < a.cout_v;            // Cout <= 64 (one co half empty): that wave row multiplies nothing
#if !defined(SAT_HIPEMU)
    // De-phase the CUs: the first wave of workgroups starts everywhere at once and every tile takes the same time, so all 256 CUs
    // would reach their epilogues — 128 KiB of stores (+ two loads of the same size in a data-gradient) each — in the same few
    // microseconds, and the HBM burst is not overlapped with anybody's K loop.  A one-off start delay of 0 .. 7 x ~4 us keeps the
    // CUs' epilogues apart for the rest of the launch.
    if (a.stagger && blockIdx.x < 256) {
        const int steps = (int)((blockIdx.x >> 3) & 7) * a.stagger;
        for (int i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    // prologue: chunk 0 complete in its stage
    auto issue_chunk0 = [&]() {
#pragma unroll
        for (int tap = 0; tap < SAT_K7Q_TAPS; ++tap) issue_w(0, SPAR, tap);
        issue_a(0, SPAR, 0); issue_a(0, SPAR, 1); issue_a(0, SPAR, 2);
    };
    issue_chunk0();
  for (;;) {                                               // (one pass unless PERSIST)
    SAT_WAIT_VMCNT(0);
    SAT_RAW_BARRIER();
    if (wr == 1) SAT_RAW_BARRIER();                        // the second wave row runs one barrier behind the first

    for (int c = 0; c < nchunks; ++c) {
        const char* sb = lds + ((c + SPAR) & 1) * SAT_K7Q_STAGE;
        const bool more = c + 1 < nchunks;
        // ---- phase 0: taps 0..3 ----
#pragma unroll
        for (int u = 0; u < 4; ++u) load_frags(fr[u], sb, u);
        if constexpr (INMFMA) {
            SAT_WAIT_VMCNT(0);                             // taps 4 .. K-1 of THIS chunk (issued in the previous chunk's second MFMA section)
        } else if constexpr (REORDER) {
            if (more) {
                issue_a(c + 1, (c + 1 + SPAR) & 1, 0); issue_a(c + 1, (c + 1 + SPAR) & 1, 1); issue_a(c + 1, (c + 1 + SPAR) & 1, 2);
                issue_w(c + 1, (c + 1 + SPAR) & 1, 0); issue_w(c + 1, (c + 1 + SPAR) & 1, 1);
                if (K != SAT_K7Q_TAPS) { SAT_WAIT_VMCNT(0); }
                else if (wave < 4) { SAT_WAIT_VMCNT(5); }
                else { SAT_WAIT_VMCNT(4); }
            } else {
                SAT_WAIT_VMCNT(0);
            }
        } else {
            if (more) {
#pragma unroll
                for (int tap = 0; tap < 5; ++tap) issue_w(c + 1, (c + 1 + SPAR) & 1, tap);
            }
        }
        SAT_WAIT_LGKM0();
        SAT_RAW_BARRIER();
        SAT_SCHED_FENCE();
        SAT_SETPRIO(1);
        if constexpr (INMFMA) {
            const int sn = (c + 1 + SPAR) & 1;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (wave_on_co) mfma_frags(fr[u]);
                SAT_SCHED_FENCE();
                if (more) {                                // (block-uniform) longest latency first: the activation pieces come from HBM
                    // (early in the section: behind taps 0 and 1 — spread over all four taps the launch was 0.4 % slower, `k7q_dma_in_mfma/`)
                    if (u == 0) { issue_a(c + 1, sn, 0); issue_a(c + 1, sn, 1); issue_a(c + 1, sn, 2); issue_w(c + 1, sn, 0); }
                    if (u == 1) { issue_w(c + 1, sn, 1); issue_w(c + 1, sn, 2); issue_w(c + 1, sn, 3); }
                }
                SAT_SCHED_FENCE();
            }
        } else
        if (wave_on_co) {
#pragma unroll
            for (int u = 0; u < 4; ++u) mfma_frags(fr[u]);
        }
        SAT_SETPRIO(0);
        SAT_SCHED_FENCE();
        SAT_RAW_BARRIER();
        // ---- phase 1: taps 4..K-1 ----
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (4 + u < K) load_frags(fr[u], sb, 4 + u);
        if constexpr (INMFMA) {
            SAT_WAIT_VMCNT(0);                             // activations and taps 0-3 of chunk c + 1 (issued in this chunk's first MFMA section)
        } else if constexpr (REORDER) {
            if (more) {
#pragma unroll
                for (int tap = 2; tap < SAT_K7Q_TAPS; ++tap) issue_w(c + 1, (c + 1 + SPAR) & 1, tap);
                if (K == SAT_K7Q_TAPS) { SAT_WAIT_VMCNT(3); } else { SAT_WAIT_VMCNT(0); }
            }
        } else if (more) {
            issue_w(c + 1, (c + 1 + SPAR) & 1, 5); issue_w(c + 1, (c + 1 + SPAR) & 1, 6);
            issue_a(c + 1, (c + 1 + SPAR) & 1, 0); issue_a(c + 1, (c + 1 + SPAR) & 1, 1); issue_a(c + 1, (c + 1 + SPAR) & 1, 2);
            SAT_WAIT_VMCNT(0);                             // chunk c+1 has landed (this wave's pieces; the barriers publish the others')
        }
        SAT_WAIT_LGKM0();
        SAT_RAW_BARRIER();
        SAT_SCHED_FENCE();
        SAT_SETPRIO(1);
        if constexpr (INMFMA) {
            const int sn = (c + 1 + SPAR) & 1;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (wave_on_co && 4 + u < K) mfma_frags(fr[u]);
                SAT_SCHED_FENCE();
                if (more) issue_w(c + 1, sn, 4 + u);       // (taps >= K: nothing)
                SAT_SCHED_FENCE();
            }
        } else
        if (wave_on_co) {
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (4 + u < K) mfma_frags(fr[u]);
        }
        SAT_SETPRIO(0);
        SAT_SCHED_FENCE();
        SAT_RAW_BARRIER();
    }
    if (wr == 0) SAT_RAW_BARRIER();                        // pairs with the second wave row's last barrier
    __syncthreads();                                       // every wave is done with the stages: their memory serves the epilogue
    // this tile's coordinates for the epilogue; PERSIST: the next tile's become current for the DMA lambdas
    const int e_b = b, e_t_tile = t_tile, e_co0 = co0, e_t0 = t0;
    bool have_next = false;
    if constexpr (PERSIST) {
        tile_id += (int)gridDim.x;
        have_next = tile_id < total_tiles;                 // block-uniform
        if (have_next) {
            const int prev_co0 = co0;
            sat_xcd_tile(tile_id, co_tiles, t_tiles * p.B, &co_tile, &win);
            b = win / t_tiles;
            t_tile = win - b * t_tiles;
            co0 = co_tile * CO_T;
            t0 = t_tile * T_T;
            row_in0 = SAT_K7P_LEAD + t0 - p.pad;
            w_src0 += (long long)(co0 - prev_co0) * 16;
            issue_chunk0();                                // into stage 1: lands under the epilogue below (which uses offsets < 70 KB)
        }
    }

    // the epilogue of one accumulator set: bias (ep_lds row 0), dsnake + its sums (data-gradients), residual, tanh, stores, plane emission
    // PERSIST: the epilogue reads its launch arguments through a pointer to the kernarg segment that is laundered once per tile — loads
    // through it cannot be hoisted out of the tile loop, where ~40 scalar registers of epilogue-only arguments would stay live across
    // the K loop (the first build of this variant spilled 112 SGPRs and, through their v_writelane homes, 171 VGPRs)
#if defined(SAT_HIPEMU)
    const SatConvBfLaunch& ea = a;
#else
    typedef const __attribute__((address_space(4))) SatConvBfLaunch* sat_kargs_t;
    sat_kargs_t eap = (sat_kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    if constexpr (PERSIST) asm volatile("" : "+s"(eap));
    const auto& ea = *eap;
#endif
    const auto& ep = ea.p;
    auto epilogue = [&](float* y_out, const float* res_in, bool emit_on) __attribute__((always_inline)) {
    const int b = e_b, t_tile = e_t_tile, co0 = e_co0, t0 = e_t0;      // (shadow the DMA lambdas' — PERSIST: already the next tile's)
    short* em_hi_ = emit_on ? ea.em_hi : nullptr;
    short* em_lo_ = emit_on ? ea.em_lo : nullptr;
    // ------------------------------------ epilogue (as the generic kernel) ------------------------------------
    const bool bwd = (ep.x2 != nullptr);
    const bool wave_on = (co0 + co_w) < ea.cout_v;
    const bool mi1_on = (co0 + co_w + 32) < ea.cout_v;
    if (bwd) {
        __syncthreads();
        for (int i = tid; i < 2 * TW * CO_T; i += NT) (&red_lds[0][0][0])[i] = 0.0f;
        __syncthreads();
    }
    const bool vec4 = (ep.Tout & 3) == 0 && (((uintptr_t)y_out | (uintptr_t)ep.x2 | (uintptr_t)res_in) & 15) == 0;
    if (vec4) {
        // 16-byte epilogue: each wave transposes its accumulators through LDS (the stage memory is free now) so that
        // a lane owns 4 consecutive time steps of a row; the x2 / res loads of a 32-row half are all issued before use.
        if (!bwd) __syncthreads();                          // (bwd already synchronised above)
        float (*tile)[68] = reinterpret_cast<float (*)[68]>(lds) + wave * 32;       // 32 x 68 floats per wave, in the (drained) stage memory
        const int lr = lane >> 4, t4 = (lane & 15) * 4;    // this lane's row within a group of 4, its 4 time steps
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const bool half_on = wave_on && (mi == 0 || mi1_on);
            if (mi == 1) __syncthreads();                   // every wave is done reading its first half
            if (half_on) {
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tile[(r & 3) + 8 * (r >> 2) + 4 * hi][ni * 32 + l31] = acc[mi][ni][r];
            }
            __syncthreads();                                // (a wave only reads its own tile: this orders its own lanes)
            if (half_on) {
                const int tg = t0 + t_w + t4;
                f32x4 xv[8], rv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int col = co_w + mi * 32 + j * 4 + lr;
                    const int co = co0 + col;
                    const bool ok = co < ea.cout_v && tg < ep.Tout;
                    const size_t o = ((size_t)b * ep.Cout + (ok ? co : 0)) * ep.Tout + (ok ? tg : 0);
                    xv[j] = (bwd && ok) ? *reinterpret_cast<const f32x4*>(ep.x2 + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                    rv[j] = (res_in && ok) ? *reinterpret_cast<const f32x4*>(res_in + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int row = j * 4 + lr;
                    const int col = co_w + mi * 32 + row;
                    const int co = co0 + col;
                    const bool ok = co < ea.cout_v && tg < ep.Tout;
                    const f32x4 av = *reinterpret_cast<const f32x4*>(&tile[row][t4]);
                    const float bias = ep_lds[0][col];
                    const float a2 = ep_lds[1][col], b2 = ep_lds[2][col];
                    float pda = 0.f, pdb = 0.f;
                    f32x4 ov;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = av[e] + bias;
                        if (bwd) {
                            const SatSnakeGrad g = sat_snake_grad(xv[j][e], a2, b2);
                            pda += v * g.dla;
                            pdb += v * g.dlb;
                            v *= g.dx;
                        }
                        v += rv[j][e];
                        if (ep.tanh_out) v = tanhf(v);
                        ov[e] = v;
                    }
                    if (ok) *reinterpret_cast<f32x4*>(y_out + ((size_t)b * ep.Cout + co) * ep.Tout + tg) = ov;
                    if (em_hi_) {
                        // plane emission (as conv1d_bf16x3.hip's generic kernel), step 1: the consumer's activation of the finished
                        // values goes back into this lane's own cell of the transposition tile (rows past Cout hold act(0) = 0)
                        f32x4 ev = ov;
                        if (ea.em_a) {
                            const float ea = ep_lds[3][col], eib = ep_lds[4][col];
#pragma unroll
                            for (int e = 0; e < 4; ++e) ev[e] = sat_snake(ov[e], ea, eib);
                        }
                        if (!ok) ev = f32x4{0.f, 0.f, 0.f, 0.f};
                        *reinterpret_cast<f32x4*>(&tile[row][t4]) = ev;
                    }
                    if (bwd) {
                        if (!ok) { pda = 0.f; pdb = 0.f; }
#pragma unroll
                        for (int m = 8; m >= 1; m >>= 1) {     // sum over the 16 lanes that share this row
                            pda += __shfl_xor(pda, m);
                            pdb += __shfl_xor(pdb, m);
                        }
                        if ((lane & 15) == 0) {
                            red_lds[0][wave % TW][col] = pda;
                            red_lds[1][wave % TW][col] = pdb;
                        }
                    }
                }
                if (em_hi_) {
                    // step 2: the tile read COLUMN-wise — 8 consecutive channels of one time step = one 16-byte plane row
                    sat_wave_sync();
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c8i = ((co0 + co_w + mi * 32) >> 3) + g;
                        const int tq = t0 + t_w + lane;
                        float v8[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v8[e] = tile[g * 8 + e][lane];
                        uint32_t eh[4], el[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) sat_split2_pk(v8[2 * e], v8[2 * e + 1], &eh[e], &el[e]);
                        if (c8i < ea.em_c8 && tq < ep.Tout) {
                            const size_t o = (((size_t)b * ea.em_c8 + c8i) * ea.em_rows + SAT_K7P_LEAD + tq) * 8;
                            *reinterpret_cast<u32x4*>(em_hi_ + o) = u32x4{eh[0], eh[1], eh[2], eh[3]};
                            *reinterpret_cast<u32x4*>(em_lo_ + o) = u32x4{el[0], el[1], el[2], el[3]};
                        }
                    }
                }
            }
        }
    } else
    if (wave_on) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            if (mi == 1 && !mi1_on) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = co_w + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int co = co0 + col;
                const bool co_ok = co < ea.cout_v;
                const float bias = ep_lds[0][col];
                const float a2 = ep_lds[1][col], b2 = ep_lds[2][col];
                float pda = 0.f, pdb = 0.f;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int t = t0 + t_w + ni * 32 + l31;
                    if (co_ok && t < ep.Tout) {
                        const size_t o = ((size_t)b * ep.Cout + co) * ep.Tout + t;
                        float v = acc[mi][ni][r] + bias;
                        if (bwd) {
                            const SatSnakeGrad g = sat_snake_grad(ep.x2[o], a2, b2);
                            pda += v * g.dla;
                            pdb += v * g.dlb;
                            v *= g.dx;
                        }
                        if (res_in) v += res_in[o];
                        if (ep.tanh_out) v = tanhf(v);
                        y_out[o] = v;
                    }
                }
                if (bwd) {
                    pda = sat_half_sum(pda);
                    pdb = sat_half_sum(pdb);
                    if (l31 == 0) {
                        red_lds[0][wave % TW][col] = pda;
                        red_lds[1][wave % TW][col] = pdb;
                    }
                }
            }
        }
    }
    if (bwd) {
        __syncthreads();
        const int m = co0 + tid;
        if (tid < CO_T && m < ea.cout_v) {
            float sa = 0.f, sb = 0.f;
#pragma unroll
            for (int w = 0; w < TW; ++w) {
                sa += red_lds[0][w][tid];
                sb += red_lds[1][w][tid];
            }
            const size_t row = (size_t)b * t_tiles + t_tile;
            const size_t nrows_p = (size_t)ep.B * t_tiles;
            ep.part_da[(size_t)m * nrows_p + row] = sa;
            ep.part_db[(size_t)m * nrows_p + row] = sb;
        }
    }
    };

    if constexpr (FUSED) {
        // ---- h = acc + bias1 -> `h` (fp32) through the ordinary 16-byte epilogue (no residual, no emission) ----
        if (a.ru_h) {
            epilogue(a.ru_h, nullptr, false);
            __syncthreads();                               // the transposition windows are free again
        }
        // the 1x1 conv's A fragments (sat_pack_weights_k7q, K = 1) come from L2 straight into registers
        auto load_w1 = [&](int u, int ks) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                const short* wsrc = (pl ? a.ru_w1_lo : a.ru_w1_hi) + ((size_t)(ks * 2 + hi) * a.cout_pad + co_w + l31) * 8;
                fr[u].wa[0][pl] = *reinterpret_cast<const bf16x8*>(wsrc);
                fr[u].wa[1][pl] = *reinterpret_cast<const bf16x8*>(wsrc + 32 * 8);
            }
        };
        // ---- snake2(h) as the B-operand planes of the 1x1 conv, straight from the accumulator layout ----
        // accumulator register r of lane (l31, hi) is row (r & 3) + 8 (r >> 2) + 4 hi, column l31: a quad of registers 4k .. 4k+3 is
        // channels 8k + 4hi + (0..3) at one time step = 8 bytes of the plane row of channel group k
        char* pb = lds;                                    // [plane][16 groups][256 rows][16 B] = 128 KiB
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int tl = t_w + ni * 32 + l31;         // row of the planes = time step within the tile
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int cl = co_w + mi * 32 + 8 * k + 4 * hi;      // first of the quad's 4 channels
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        v[j] = sat_snake(acc[mi][ni][4 * k + j] + ep_lds[0][cl + j], ep_lds[1][cl + j], ep_lds[2][cl + j]);
                    uint32_t h0, l0, h1, l1;
                    sat_split2_pk(v[0], v[1], &h0, &l0);
                    sat_split2_pk(v[2], v[3], &h1, &l1);
                    const int g = (co_w + mi * 32) / 8 + k;
                    char* dst = pb + g * 4096 + tl * 16 + hi * 8;
                    *reinterpret_cast<u32x2_q*>(dst) = u32x2_q{h0, h1};
                    *reinterpret_cast<u32x2_q*>(dst + 65536) = u32x2_q{l0, l1};
                }
            }
        __syncthreads();                                   // the planes are complete (a wave reads all 128 channels of its 64 time steps)
        if (tid < CO_T) ep_lds[0][tid] = ep_lds[5][tid];   // the final epilogue adds the 1x1 conv's bias (row 0 was last read above)
        // ---- y_acc = W1 (128 x 128) . planes: 8 k-steps of 16 channels ----
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ks = half * 4 + u;               // k-step: channels 16 ks .. 16 ks + 15 (group 2 ks + hi)
                if (ks >= nchunks) continue;               // (C < 128: the packed 1x1 weight has ceil(C / 16) chunks)
                load_w1(u, ks);
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const char* ab = pb + pl * 65536 + (ks * 2 + hi) * 4096;
                    fr[u].xa[0][pl] = *reinterpret_cast<const bf16x8*>(ab + (t_w + l31) * 16);
                    fr[u].xa[1][pl] = *reinterpret_cast<const bf16x8*>(ab + (t_w + 32 + l31) * 16);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (half * 4 + u < nchunks) mfma_frags(fr[u]);
        }
        __syncthreads();                                   // the planes are dead: their memory serves the epilogue's transposition
    }

    epilogue(p.y, p.res, true);
    if constexpr (!PERSIST) {
        break;
    } else {
        if (!have_next) break;
        if (e_co0 != co0) {                                // a different channel tile: its bias / SnakeBeta constants
            __syncthreads();
            load_consts();
        }
        wave_on_co = (co0 + co_w) < a.cout_v;
        zero_acc();
#if !defined(SAT_HIPEMU)
        asm volatile("" : "+v"(lane));                     // opaque per-tile copies: nothing derived from them is a tile-loop invariant
        asm volatile("" : "+s"(wave));
#endif
        tid = wave * 64 + lane;
        l31 = lane & 31;
        hi = lane >> 5;
        lane16 = (unsigned)lane * 16u;
        wr = wave / TW;
        co_w = wr * 64;
        t_w = (wave % TW) * 64;
    }
  }
}

static void sat_bf_launch_k7q(SatConvBfLaunch& a, void* stream) {
    const long long total = (long long)(a.cout_pad / SAT_K7_CO) * sat_cdiv(a.nq, SAT_K7_T) * a.p.B;
    a.stagger = 1;                                         // measured: -1 % forward, -2.5 % data-gradient (tools/kq_ab.py); 2 and 4 lose
    if (total < 1024) a.stagger = 0;                       // few tiles per CU: the delay would not be paid back
    if (a.ru_w1_hi) { SAT_LAUNCH((sat_conv1d_bf16x3_k7q_kernel<1, true>), dim3((unsigned)total), dim3(SAT_K7_NT), stream, a); return; }
    const int cus = sat_cu_count();
    if (a.persist == 2 && total >= 2) {                    // test hook (flags bit 1): persistent at ANY size, three tiles per workgroup
        a.stagger = 0;
        if (a.dma_in_mfma) { SAT_LAUNCH((sat_conv1d_bf16x3_k7q_kernel<3, false, true>), dim3((unsigned)sat_cdivll(total, 3)), dim3(SAT_K7_NT), stream, a); }
        else { SAT_LAUNCH((sat_conv1d_bf16x3_k7q_kernel<1, false, true>), dim3((unsigned)sat_cdivll(total, 3)), dim3(SAT_K7_NT), stream, a); }
        return;
    }
    if (a.persist && total >= 2 * cus) {                   // PERSIST: one workgroup per CU walks total / cus tiles (>= 2 each)
        if (a.dma_in_mfma) { SAT_LAUNCH((sat_conv1d_bf16x3_k7q_kernel<3, false, true>), dim3((unsigned)cus), dim3(SAT_K7_NT), stream, a); }
        else { SAT_LAUNCH((sat_conv1d_bf16x3_k7q_kernel<1, false, true>), dim3((unsigned)cus), dim3(SAT_K7_NT), stream, a); }
        return;
    }
    if (a.dma_in_mfma) { SAT_LAUNCH((sat_conv1d_bf16x3_k7q_kernel<3>), dim3((unsigned)total), dim3(SAT_K7_NT), stream, a); }
    else { SAT_LAUNCH((sat_conv1d_bf16x3_k7q_kernel<1>), dim3((unsigned)total), dim3(SAT_K7_NT), stream, a); }
}
