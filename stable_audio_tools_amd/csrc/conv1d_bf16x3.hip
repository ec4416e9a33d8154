// conv1d_bf16x3.hip — the k = 5..8, stride-1 (dilated) Oobleck convolutions — 7/8 of the conv
// stack's FLOPs: the k7 convs of every ResidualUnit (autoencoders.py:58-83) and their data-gradients —
// on the bf16 matrix cores at fp32 accuracy.
//
// Why: with fp32 operands the stack is bound by v_mfma_f32_32x32x2_f32 (157 TFLOP/s); the bf16 pipe is
// 16x faster.  Every fp32 value is split as x = hi + lo (two bf16, |x - hi - lo| <= 2^-17 |x|) and each
// product is three MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulate): error ~2^-16 per product — far
// inside the 1e-3 parity bar — at 1/3 of the bf16 rate = 5.3x the fp32-MFMA rate.
//
// Implicit GEMM, same 128(co) x 128(t) workgroup tile / 2x2 32x32 accumulators per wave / epilogues
// as conv1d.hip.  What differs is the K side (v_mfma_f32_32x32x16_bf16 wants 8 consecutive k per lane):
//   * a K-chunk is 8 input channels x all taps = 8 "groups" of 8 k-values (tap g, 8 channels); group
//     7 is a zero pad when K = 7.  One MFMA k-step (16 k) = groups 2j (lanes 0-31) and 2j+1 (lanes 32-63).
//   * activations are staged TRANSPOSED: LDS rows are time steps holding 8 channels (16 B), so the B
//     fragment of (tap g, time t) is ONE 16-byte read of row t + g*dil — SnakeBeta and the hi/lo
//     split are applied once per element while staging.
//   * weights are pre-split and pre-arranged by sat_pack_weights_bf16x3 as [chunk][co][group][8],
//     so a chunk's slab is a straight 16-byte-per-lane copy into padded LDS rows.
#include "conv_common.h"
#include <type_traits>

#define SAT_K7P_LEAD_ROWS 32  // zero rows before t = 0 in an activation plane (= SAT_K7P_LEAD of conv1d_planes.h)
#define SAT_BF_AROWS1 192  // CS == 1: max staged time rows: 128 + (K-1)*dil <= 128 + 7*9 = 191
#define SAT_BF_AROWSN 136  // CS  > 1: 128 + (taps-1) rows, taps <= 4, dil = 1

struct SatConvBfLaunch {
    SatConvParams p;       // REAL tensor dims (Cin, Tin, Cout, Tout); p.K / p.dil / p.pad describe the VIRTUAL stride-1 conv;
                           // p.w unused; p.alpha / p.beta hold PRE-EXPONENTIATED snake constants: a = e^alpha, ib = 1/(e^beta+1e-9)
    const short* w_hi;     // [nchunks][cout_pad][NG][8]
    const short* w_lo;
    int cout_pad;          // virtual output channels rounded up to the 128 tile
    int cin_v, cout_v;     // virtual channel counts: Cin << sin_log2, Cout << sout_log2
    int sin_log2;          // space-to-depth of the input:  x'[ci*S + r][q] = x[ci][q*S + r - in_shift]   (strided conv)
    int sout_log2;         // depth-to-space of the output: y[co][q*S + r - out_shift] = y'[co*S + r][q]  (transposed conv)
    int in_shift, out_shift;
    int nq;                // virtual output positions
    // conv1d_planes.h: the activation as pre-split planes [B][xp_c8][xp_rows][8] (null: convert from p.x while staging)
    const short* xp_hi = nullptr;
    const short* xp_lo = nullptr;
    int xp_rows = 0, xp_c8 = 0;
    int wq = 0;            // conv1d_bf16x3_k7q.h: the weight planes are in sat_pack_weights_k7q layout ([chunk16][tap][group][co][8])
    int stagger = 0;       // conv1d_bf16x3_k7q.h: start delay units (x ~4 us x (0..7)) that de-phase the CUs' epilogue bursts
    int persist = 0;       // conv1d_bf16x3_k7q.h: one workgroup per CU walks the tiles, the next tile's first chunk requested before the epilogue
    int dma_in_mfma = 0;   // conv1d_bf16x3_k7q.h VARIANT 3: the next chunk's LDS-DMA issued inside the MFMA sections
    // plane EMISSION (generic kernel, 16-byte epilogue): besides y the kernel writes act(y) as the bf16 hi / lo planes the next k7
    // conv reads ([B][em_c8][em_rows][8], row = 32 + t; the zero rows around the sequence belong to the caller) — the consumer's
    // sat_k7_planes_kernel pre-pass (one read + one write of the tensor) disappears.  em_a / em_ib: pre-exponentiated SnakeBeta
    // constants of the CONSUMER's activation (null: planes of y itself, for a data-gradient consumer).
    short* em_hi = nullptr;
    short* em_lo = nullptr;
    const float* em_a = nullptr;
    const float* em_ib = nullptr;
    int em_rows = 0, em_c8 = 0;
    // fused ResidualUnit forward (conv1d_bf16x3_k7q.h, FUSED): the 1x1 conv's weights in sat_pack_weights_k7q layout (K = 1), its bias,
    // the second SnakeBeta's pre-exponentiated constants, and where the intermediate h goes (fp32 (B, C, T), or null: not kept)
    const short* ru_w1_hi = nullptr;
    const short* ru_w1_lo = nullptr;
    const float* ru_bias2 = nullptr;
    const float* ru_a2 = nullptr;
    const float* ru_ib2 = nullptr;
    float* ru_h = nullptr;
};

SAT_DEVICE void sat_split2(float x, short* hi, short* lo) {
    const short h = sat_f32_to_bf16(x);
    *hi = h;
    *lo = sat_f32_to_bf16(x - sat_bf16_to_f32(h));
}

// NG = k-groups (8 values each) per K-chunk, CS = 8-channel sub-blocks per chunk; taps per sub-block KT = NG / CS.
// group g of a chunk = (sub-block g / KT, tap g % KT).  GATHER: the input is read space-to-depth (sin_log2 > 0).
//
// Activation staging is the VALU-heavy part (SnakeBeta + hi/lo split per element), so it is laid out to keep every
// channel index wave-uniform (scalar address arithmetic, s_load for the snake constants): a wave owns one sub-block
// (CS > 1) and its lanes are time rows; the (K'-1) halo rows of a CS > 1 chunk are staged one ELEMENT per lane.
template <int NG, int CS, bool GATHER>
__global__ void __launch_bounds__(256)
#if !defined(SAT_HIPEMU)
// the k = 1 plan (NG = 4) is bound by bytes in flight (3 streams of 4*C*T bytes, 48 MFMAs per wave and tile-chunk): three workgroups per CU
__attribute__((amdgpu_waves_per_eu(NG == 4 ? 3 : 2)))
#endif
sat_conv1d_bf16x3_kernel(SatConvBfLaunch a) {
    constexpr int KT = NG / CS;
    constexpr int KROW = NG * 8 + 8;                      // bf16 per weight row in LDS (+8 pad: conflict-free b128 reads)
    constexpr int AROWS = (CS == 1) ? SAT_BF_AROWS1 : SAT_BF_AROWSN;
    constexpr int WPS = 4 / CS;                           // waves per sub-block
    constexpr int NU = (CS == 1) ? 1 : 128 / (64 * WPS);  // main staging items (one time row x 8 channels) per thread
    const SatConvParams& p = a.p;
    // one LDS pool: the weight slab and the activation slab of the K loop, reused by the epilogue as the per-wave
    // transposition tiles (4 waves x 32 rows x 68 floats)
    constexpr int W_BYTES = 2 * SAT_CO_T * KROW * 2, A_BYTES = 2 * CS * AROWS * 8 * 2, T_BYTES = 4 * 32 * 68 * 4;
    constexpr int POOL = (W_BYTES + A_BYTES) > T_BYTES ? (W_BYTES + A_BYTES) : T_BYTES;
    __shared__ __attribute__((aligned(16))) char lds_pool[POOL];
    short (*w_lds)[SAT_CO_T][KROW] = reinterpret_cast<short (*)[SAT_CO_T][KROW]>(lds_pool);                 // [plane][co][g*8+e]
    short (*a_lds)[CS][AROWS][8] = reinterpret_cast<short (*)[CS][AROWS][8]>(lds_pool + W_BYTES);           // [plane][sub-block][time row][8 ci]
    __shared__ float red_lds[2][2][SAT_CO_T];
    __shared__ float ep_lds[5][SAT_CO_T];
    // the input's SnakeBeta constants (pre-exponentiated), ALL channels, staged once (round 6).  They used to be read per chunk inside
    // write_lds: the channel is wave-uniform, but hipcc cannot prove the arrays unclobbered and emits sixteen VECTOR loads from a uniform
    // address behind the chunk's barrier — an L2 round trip on the critical path of every chunk, and a vmcnt(0) that also drains whatever
    // was prefetched further ahead.  (k = 1 plan: 1024 channels = 8 KB, still three workgroups per CU; the others 2048 = 16 KB, two.)
    constexpr int SN_MAX = (NG == 4) ? 1024 : 2048;
    __shared__ float sn_lds[2][SN_MAX];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    // grid = (time tiles, channel tiles, B): the channel tiles of one activation window share an XCD (sat_xcd_tile)
    int co_tile, win;
    sat_xcd_tile(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.y, gridDim.x * gridDim.z, &co_tile, &win);
    const int b = win / (int)gridDim.x, t_tile = win - b * (int)gridDim.x;
    const int t0 = t_tile * SAT_T_T;
    const int co0 = co_tile * SAT_CO_T;
    const int co_w = (wave >> 1) * 64, t_w = (wave & 1) * 64;
    const int K = p.K, dil = p.dil;
    const int nrows = SAT_T_T + (K - 1) * dil;
    const int q_in0 = t0 - p.pad;
    const int si = GATHER ? a.sin_log2 : 0, so = a.sout_log2;
    const int smask_i = (1 << si) - 1, smask_o = (1 << so) - 1;
    const float* xb = p.x + (size_t)b * p.Cin * p.Tin;
    const bool wave_on = (co0 + co_w) < a.cout_v;
    const bool mi1_on = (co0 + co_w + 32) < a.cout_v;
    const bool has_snake = p.alpha != nullptr;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    if (tid < SAT_CO_T) {
        const int m = co0 + tid;
        const bool ok = m < a.cout_v;
        const int co = m >> so;
        ep_lds[0][tid] = (ok && p.bias) ? p.bias[co] : 0.0f;
        ep_lds[1][tid] = (ok && p.x2) ? expf(p.alpha2[co]) : 1.0f;
        ep_lds[2][tid] = (ok && p.x2) ? expf(p.beta2[co]) : 1.0f;
        ep_lds[3][tid] = (ok && a.em_a) ? a.em_a[co] : 0.0f;
        ep_lds[4][tid] = (ok && a.em_a) ? a.em_ib[co] : 0.0f;
    }
    if (has_snake) {
        const int nsn = p.Cin < SN_MAX ? p.Cin : SN_MAX;
        for (int i = tid; i < nsn; i += 256) {
            sn_lds[0][i] = p.alpha[i];
            sn_lds[1][i] = p.beta[i];
        }
    }
    __syncthreads();                                       // (the first write_lds reads the table)

    const int nchunks = (a.cin_v + 8 * CS - 1) / (8 * CS);
    // Register-staged software pipeline: the global loads of chunk c+1 are issued right after the barrier that
    // publishes chunk c, so their HBM/L2 latency runs under chunk c's MFMA phase; they are converted and written
    // to LDS after the phase's closing barrier.
    const int cs_w = SAT_UNIFORM(wave / WPS);             // this wave's sub-block
    const int row0 = (wave % WPS) * 64 + lane;            // its first time row; item u is row0 + u * 64 * WPS
    // halo rows (CS > 1 only): row 128 + lane/8, element lane%8, staged by the first wave of each sub-block
    const int nhalo = (CS > 1) ? (nrows - SAT_T_T) * 8 : 0;
    const bool halo_on = (CS > 1) && (wave % WPS) == 0 && lane < nhalo;
    const int h_row = SAT_T_T + (lane >> 3), h_e = lane & 7;
    // DEEP (round 6, the strided / transposed plans <8, 4>): TWO staging register sets — chunk c + 2 is requested while chunk c multiplies and
    // is converted after chunk c + 1's MFMA phase, two phases of latency cover instead of one (these launches sat at 0.22 matrix-busy and
    // 0.8-1.1 TB/s: two workgroups of four waves per CU, each stalled on a single chunk of 4-byte loads per phase).  The k = 1 plan keeps one
    // set (three workgroups per CU at 164 registers).
    constexpr bool DEEP = (NG == 8 && CS == 4);
    constexpr int NSET = DEEP ? 2 : 1;
    float av[NSET][NU][8];
    float hv[NSET];
    bf16x8 wv[NG];                                        // the weight slab (L2-resident) stays one chunk ahead in ONE set
#pragma unroll
    for (int s_ = 0; s_ < NSET; ++s_) hv[s_] = 0.0f;
    // element (virtual channel v, time row) -> real input: channel v >> si at time ((q_in0 + row) << si) + (v & mask) - in_shift;
    // channels past the end are clamped (their weights are zero), times outside [0, Tin) read as 0 (= snake(0)).
    // The value comes back RAW (from a clamped address) with its in-range flag beside it: `ok ? val : 0` right behind the load makes hipcc wait
    // for every load of a chunk before the MFMA phase they were issued in front of (round 6: the ISA of rounds 1-5 shows vmcnt(15) ... vmcnt(0)
    // between the loads and the first MFMA — the "register-staged software pipeline" never overlapped anything).  The flags travel as one
    // bit mask per staging set and are applied when the set is converted (write_lds).
    auto load_elem = [&](int v, int row, bool* okp) -> float {
        int ch = v >> si;
        ch = ch < p.Cin ? ch : p.Cin - 1;
        const int tin = ((q_in0 + row) << si) + (v & smask_i) - (GATHER ? a.in_shift : 0);
        const bool ok = (unsigned)tin < (unsigned)p.Tin;
        *okp = ok;
        return xb[(size_t)ch * p.Tin + (ok ? tin : 0)];
    };
    unsigned okm[NSET];
#pragma unroll
    for (int s_ = 0; s_ < NSET; ++s_) okm[s_] = 0u;
    auto issue_loads = [&](int c, auto set_c) {
        constexpr int ST = decltype(set_c)::value;
        const int v0 = (c * CS + cs_w) * 8;
        unsigned m = 0u;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int row = row0 + u * 64 * WPS;
            if (CS == 1 && row >= nrows) continue;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bool ok;
                av[ST][u][e] = load_elem(v0 + e, row, &ok);
                m |= ok ? (1u << (u * 8 + e)) : 0u;
            }
        }
        static_assert(NU * 8 < 31, "the in-range flags of a staging set are one 32-bit mask (bit 31: the halo element)");
        if (CS > 1 && halo_on) {
            bool ok;
            hv[ST] = load_elem(v0 + h_e, h_row, &ok);
            m |= ok ? (1u << 31) : 0u;
        }
        okm[ST] = m;
    };
    auto issue_w = [&](int c) {
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int idx = tid + u * 256;                 // part = idx % NG, co = (idx / NG) & 127, plane = idx / (NG * 128)
            const int part = idx % NG, co = (idx / NG) & 127, pl = idx / (NG * 128);
            const short* src = (pl ? a.w_lo : a.w_hi) + (((size_t)c * a.cout_pad + co0 + co) * NG + part) * 8;
            wv[u] = *reinterpret_cast<const bf16x8*>(src);
        }
    };
    auto write_lds = [&](int c, auto set_c) {
        constexpr int ST = decltype(set_c)::value;
        const int v0 = (c * CS + cs_w) * 8;
        float sa[8], sib[8];
        if (has_snake) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {                  // wave-uniform channel: LDS broadcast reads of the staged (pre-exponentiated) constants
                int ch = (v0 + e) >> si;
                ch = ch < p.Cin ? ch : p.Cin - 1;
                sa[e] = sn_lds[0][ch];                     // (Cin <= SN_MAX: the host entry points check)
                sib[e] = sn_lds[1][ch];
            }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int row = row0 + u * 64 * WPS;
            if (CS == 1 && row >= nrows) continue;
            uint32_t ph[4], pl[4];
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                float o0 = (okm[ST] >> (u * 8 + e)) & 1u ? av[ST][u][e] : 0.0f, o1 = (okm[ST] >> (u * 8 + e + 1)) & 1u ? av[ST][u][e + 1] : 0.0f;
                if (has_snake) {
                    o0 = sat_snake(o0, sa[e], sib[e]);
                    o1 = sat_snake(o1, sa[e + 1], sib[e + 1]);
                }
                sat_split2_pk(o0, o1, &ph[e >> 1], &pl[e >> 1]);
            }
            *reinterpret_cast<u32x4*>(&a_lds[0][cs_w][row][0]) = u32x4{ph[0], ph[1], ph[2], ph[3]};
            *reinterpret_cast<u32x4*>(&a_lds[1][cs_w][row][0]) = u32x4{pl[0], pl[1], pl[2], pl[3]};
        }
        if (CS > 1 && halo_on) {
            float o = (okm[ST] >> 31) & 1u ? hv[ST] : 0.0f;
            if (has_snake) {
                int ch = (v0 + h_e) >> si;
                ch = ch < p.Cin ? ch : p.Cin - 1;
                o = sat_snake(o, sn_lds[0][ch], sn_lds[1][ch]);
            }
            short h, l;
            sat_split2(o, &h, &l);
            a_lds[0][cs_w][h_row][h_e] = h;
            a_lds[1][cs_w][h_row][h_e] = l;
        }
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int idx = tid + u * 256;
            const int part = idx % NG, co = (idx / NG) & 127, pl = idx / (NG * 128);
            *reinterpret_cast<bf16x8*>(&w_lds[pl][co][part * 8]) = wv[u];
        }
    };
    auto mfma_chunk = [&]() {
        if (wave_on) {
            SAT_MFMA_PRIO(1);
#pragma unroll
            for (int ks = 0; ks < NG / 2; ++ks) {
                const int g = 2 * ks + hi;                 // k-slots 0-7 <- group 2ks (lanes 0-31), 8-15 <- group 2ks+1
                const int cs = g / KT;
                int tap = g - cs * KT;
                if (tap >= K) tap = K - 1;                 // taps >= K are zero-weight pads; keep the row in range
                bf16x8 wa[2][2], xa[2][2];                 // [mi|ni][plane]
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    wa[0][pl] = *reinterpret_cast<const bf16x8*>(&w_lds[pl][co_w + l31][g * 8]);
                    wa[1][pl] = *reinterpret_cast<const bf16x8*>(&w_lds[pl][co_w + 32 + l31][g * 8]);
                    xa[0][pl] = *reinterpret_cast<const bf16x8*>(&a_lds[pl][cs][t_w + l31 + tap * dil][0]);
                    xa[1][pl] = *reinterpret_cast<const bf16x8*>(&a_lds[pl][cs][t_w + 32 + l31 + tap * dil][0]);
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    if (mi == 1 && !mi1_on) break;
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        acc[mi][ni] = sat_mfma_32x32x16_bf16(wa[mi][0], xa[ni][0], acc[mi][ni]);
                        acc[mi][ni] = sat_mfma_32x32x16_bf16(wa[mi][0], xa[ni][1], acc[mi][ni]);
                        acc[mi][ni] = sat_mfma_32x32x16_bf16(wa[mi][1], xa[ni][0], acc[mi][ni]);
                    }
                }
            }
            SAT_MFMA_PRIO(0);
        }
    };
    using SatSet0 = std::integral_constant<int, 0>;
    using SatSet1 = std::integral_constant<int, NSET - 1>;
    issue_loads(0, SatSet0{});
    issue_w(0);
    if constexpr (DEEP) {
        if (nchunks > 1) issue_loads(1, SatSet1{});
        for (int c = 0; c < nchunks; c += 2) {
            write_lds(c, SatSet0{});
            __syncthreads();
            // (oldest first: vmcnt retires in order — the weights of chunk c + 1 are needed one phase before the activations of c + 2)
            if (c + 1 < nchunks) issue_w(c + 1);
            SAT_SCHED_FENCE();                             // (the scheduler would otherwise mix the two groups: program order IS the retire order)
            if (c + 2 < nchunks) issue_loads(c + 2, SatSet0{});
            SAT_SCHED_FENCE();
            mfma_chunk();
            __syncthreads();
            if (c + 1 < nchunks) {                         // (block-uniform)
                write_lds(c + 1, SatSet1{});
                __syncthreads();
                if (c + 2 < nchunks) issue_w(c + 2);
                SAT_SCHED_FENCE();
                if (c + 3 < nchunks) issue_loads(c + 3, SatSet1{});
                SAT_SCHED_FENCE();
                mfma_chunk();
                __syncthreads();
            }
        }
    } else {
        for (int c = 0; c < nchunks; ++c) {
            write_lds(c, SatSet0{});
            __syncthreads();
            if (c + 1 < nchunks) {
                issue_loads(c + 1, SatSet0{});
                issue_w(c + 1);
            }
            mfma_chunk();
            __syncthreads();
        }
    }

    // ------------- epilogue (as conv1d.hip; virtual channel m = co*S + r lands on y[co][q*S + r - out_shift]) -------------
    const bool bwd = (p.x2 != nullptr);
    if (bwd) {
        for (int i = tid; i < 2 * 2 * SAT_CO_T; i += 256) (&red_lds[0][0][0])[i] = 0.0f;
        __syncthreads();
    }
    const bool vec4 = so == 0 && a.out_shift == 0 && (p.Tout & 3) == 0 &&
                      (((uintptr_t)p.y | (uintptr_t)p.x2 | (uintptr_t)p.res) & 15) == 0;
    if (vec4) {
        // 16-byte epilogue (plain output addressing): each wave transposes its accumulators through LDS so that a lane
        // owns 4 consecutive time steps of a row; the x2 / res loads of a 32-row half are all issued before use.
        if (!bwd) __syncthreads();                          // the K loop's slabs are free (bwd synchronised above)
        float (*tile)[68] = reinterpret_cast<float (*)[68]>(lds_pool) + wave * 32;
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
                if (co0 + SAT_CO_T <= p.Cout && t0 + SAT_T_T <= p.Tout) {
                    // a tile fully inside the tensor (block-uniform): the loads go out back to back, none wrapped in its own branch
                    const size_t o0 = ((size_t)b * p.Cout + co0 + co_w + mi * 32 + lr) * p.Tout + tg;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        xv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        rv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (bwd) xv[j] = *reinterpret_cast<const f32x4*>(p.x2 + o0 + (size_t)(j * 4) * p.Tout);
                        if (p.res) rv[j] = *reinterpret_cast<const f32x4*>(p.res + o0 + (size_t)(j * 4) * p.Tout);
                    }
                } else
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int co = co0 + co_w + mi * 32 + j * 4 + lr;
                    const bool ok = co < p.Cout && tg < p.Tout;
                    const size_t o = ((size_t)b * p.Cout + (ok ? co : 0)) * p.Tout + (ok ? tg : 0);
                    xv[j] = (bwd && ok) ? *reinterpret_cast<const f32x4*>(p.x2 + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                    rv[j] = (p.res && ok) ? *reinterpret_cast<const f32x4*>(p.res + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int row = j * 4 + lr;
                    const int col = co_w + mi * 32 + row;
                    const int co = co0 + col;
                    const bool ok = co < p.Cout && tg < p.Tout;
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
                        if (p.tanh_out) v = tanhf(v);
                        ov[e] = v;
                    }
                    if (ok) *reinterpret_cast<f32x4*>(p.y + ((size_t)b * p.Cout + co) * p.Tout + tg) = ov;
                    if (a.em_hi) {
                        // plane emission, step 1: the consumer's activation of the finished values goes back into this lane's own
                        // cell of the transposition tile (rows past Cout hold act(0) = 0)
                        f32x4 ev = ov;
                        if (a.em_a) {
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
                            red_lds[0][wave & 1][col] = pda;
                            red_lds[1][wave & 1][col] = pdb;
                        }
                    }
                }
                if (a.em_hi) {
                    // step 2: read the tile COLUMN-wise — an item is 8 consecutive channels of one time step = one 16-byte plane
                    // row — split and store: 64 lanes write 1 KiB of each plane contiguously (the wave's LDS ops execute in order)
                    sat_wave_sync();
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int tt = lane, g = it;
                        const int c8i = ((co0 + co_w + mi * 32) >> 3) + g;
                        const int tq = t0 + t_w + tt;
                        float v8[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v8[e] = tile[g * 8 + e][tt];
                        uint32_t h[4], l[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) sat_split2_pk(v8[2 * e], v8[2 * e + 1], &h[e], &l[e]);
                        if (c8i < a.em_c8 && tq < p.Tout) {
                            const size_t o = (((size_t)b * a.em_c8 + c8i) * a.em_rows + SAT_K7P_LEAD_ROWS + tq) * 8;
                            *reinterpret_cast<u32x4*>(a.em_hi + o) = u32x4{h[0], h[1], h[2], h[3]};
                            *reinterpret_cast<u32x4*>(a.em_lo + o) = u32x4{l[0], l[1], l[2], l[3]};
                        }
                    }
                }
            }
        }
    } else
    if (wave_on && so >= 1 && ((t0 << so) - a.out_shift) >= 0 && (((t0 + SAT_T_T - 1) << so) + smask_o - a.out_shift) < p.Tout) {
        // depth-to-space epilogue of an INTERIOR tile (round 6): the four consecutive accumulator rows of a lane are four consecutive TIME steps
        // of one real channel (S >= 4) or two pairs (S = 2): x2 / res / y move as 16- / 8-byte accesses at dword alignment, a quarter /
        // half of the memory instructions of the element-wise path below (which keeps the first and the last tile of an item).
        auto ep_vec = [&](auto lg_c) {
            constexpr int GL = 1 << decltype(lg_c)::value, NGP = 4 / GL;      // group length, groups per accumulator quad
            typedef float vecu __attribute__((ext_vector_type(GL), aligned(4)));
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                if (mi == 1 && !mi1_on) break;
                vecu xv[4][NGP][2], rv[4][NGP][2];
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                    for (int gp = 0; gp < NGP; ++gp) {
                        const int m = co0 + co_w + mi * 32 + 8 * rg + 4 * hi + gp * GL;
                        const bool m_ok = m < a.cout_v;
                        const int co = m_ok ? m >> so : 0;
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            const int q = t0 + t_w + ni * 32 + l31;
                            const size_t o = ((size_t)b * p.Cout + co) * p.Tout + ((q << so) + (m & smask_o) - a.out_shift);
#pragma unroll
                            for (int e = 0; e < GL; ++e) { xv[rg][gp][ni][e] = 0.0f; rv[rg][gp][ni][e] = 0.0f; }
                            if (bwd) xv[rg][gp][ni] = *reinterpret_cast<const vecu*>(p.x2 + o);
                            if (p.res) rv[rg][gp][ni] = *reinterpret_cast<const vecu*>(p.res + o);
                        }
                    }
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                    for (int gp = 0; gp < NGP; ++gp) {
                        const int col0 = co_w + mi * 32 + 8 * rg + 4 * hi + gp * GL;
                        const int m = co0 + col0;
                        const bool m_ok = m < a.cout_v;
                        const int co = m_ok ? m >> so : 0;
                        float pda[GL], pdb[GL];
#pragma unroll
                        for (int e = 0; e < GL; ++e) { pda[e] = 0.f; pdb[e] = 0.f; }
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            const int q = t0 + t_w + ni * 32 + l31;
                            vecu ov;
#pragma unroll
                            for (int e = 0; e < GL; ++e) {
                                const int col = col0 + e;
                                float v = acc[mi][ni][rg * 4 + gp * GL + e] + ep_lds[0][col];
                                if (bwd) {
                                    const SatSnakeGrad g = sat_snake_grad(xv[rg][gp][ni][e], ep_lds[1][col], ep_lds[2][col]);
                                    pda[e] += v * g.dla;
                                    pdb[e] += v * g.dlb;
                                    v *= g.dx;
                                }
                                v += rv[rg][gp][ni][e];
                                if (p.tanh_out) v = tanhf(v);
                                ov[e] = v;
                            }
                            if (m_ok) *reinterpret_cast<vecu*>(p.y + ((size_t)b * p.Cout + co) * p.Tout + ((q << so) + (m & smask_o) - a.out_shift)) = ov;
                        }
                        if (bwd) {
#pragma unroll
                            for (int e = 0; e < GL; ++e) {
                                const float sa_ = sat_half_sum(m_ok ? pda[e] : 0.f), sb_ = sat_half_sum(m_ok ? pdb[e] : 0.f);
                                if (l31 == 0) {
                                    red_lds[0][wave & 1][col0 + e] = sa_;
                                    red_lds[1][wave & 1][col0 + e] = sb_;
                                }
                            }
                        }
                    }
            }
        };
        if (so >= 2) ep_vec(std::integral_constant<int, 2>{});
        else ep_vec(std::integral_constant<int, 1>{});
    } else
    if (wave_on) {
        // 4-byte epilogue (depth-to-space outputs: transposed convs and the data-gradients of strided convs).  Round 6: the x2 / res loads of
        // a 32-row half are ALL issued (from clamped addresses, no branch around them) before the first is used — they used to sit inside
        // the per-element `if (in range)` block, one dependent load per element: 64 serialised HBM latencies per thread and tile in the
        // strided convs' data-gradients.
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            if (mi == 1 && !mi1_on) break;
            float xv[16][2], rv[16][2];
            unsigned okb = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = co_w + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int m = co0 + col;
                const bool m_ok = m < a.cout_v;
                const int co = m >> so;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int q = t0 + t_w + ni * 32 + l31;
                    const int t = (q << so) + (m & smask_o) - a.out_shift;
                    const bool ok = m_ok && t >= 0 && t < p.Tout;
                    okb |= ok ? (1u << (2 * r + ni)) : 0u;
                    const size_t o = ((size_t)b * p.Cout + (ok ? co : 0)) * p.Tout + (ok ? t : 0);
                    xv[r][ni] = bwd ? p.x2[o] : 0.0f;
                    rv[r][ni] = p.res ? p.res[o] : 0.0f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = co_w + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int m = co0 + col;
                const int co = m >> so;
                const float bias = ep_lds[0][col];
                const float a2 = ep_lds[1][col], b2 = ep_lds[2][col];
                float pda = 0.f, pdb = 0.f;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int q = t0 + t_w + ni * 32 + l31;
                    const int t = (q << so) + (m & smask_o) - a.out_shift;
                    const bool ok = (okb >> (2 * r + ni)) & 1u;
                    float v = acc[mi][ni][r] + bias;
                    if (bwd) {
                        const SatSnakeGrad g = sat_snake_grad(xv[r][ni], a2, b2);
                        pda += ok ? v * g.dla : 0.0f;
                        pdb += ok ? v * g.dlb : 0.0f;
                        v *= g.dx;
                    }
                    v += rv[r][ni];
                    if (p.tanh_out) v = tanhf(v);
                    if (ok) p.y[((size_t)b * p.Cout + co) * p.Tout + t] = v;
                }
                if (bwd) {
                    pda = sat_half_sum(pda);
                    pdb = sat_half_sum(pdb);
                    if (l31 == 0) {
                        red_lds[0][wave & 1][col] = pda;
                        red_lds[1][wave & 1][col] = pdb;
                    }
                }
            }
        }
    }
    if (bwd) {
        __syncthreads();
        const int m = co0 + tid;
        if (tid < SAT_CO_T && (tid & smask_o) == 0 && m < a.cout_v) {      // one thread per REAL channel: sum its S phases
            float sa = 0.f, sb = 0.f;
            for (int r = 0; r <= smask_o; ++r) {
                sa += red_lds[0][0][tid + r] + red_lds[0][1][tid + r];
                sb += red_lds[1][0][tid + r] + red_lds[1][1][tid + r];
            }
            const size_t row = (size_t)b * gridDim.x + t_tile;
            const size_t nrows_p = (size_t)p.B * gridDim.x;
            p.part_da[(size_t)(m >> so) * nrows_p + row] = sa;
            p.part_db[(size_t)(m >> so) * nrows_p + row] = sb;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Plans.  Every conv of the stack is a stride-1 implicit GEMM over virtual channels:
//   conv, stride 1:            K' = K,  plain channels
//   conv, stride S, K = 2S:    K' = 2,  in-channels (ci, r):   x'[(ci,r)][q] = x[ci][q*S + r - pad],   W'[co][(ci,r)][j] = w[co][ci][j*S + r]
//   conv_transpose, K = 2S:    K' = 2,  out-channels (co, r):  y[co][q*S + r - pad] = y'[(co,r)][q],
//                              y'[(co,r)][q] = sum_ci W[ci][co][r+S] x[ci][q-1] + W[ci][co][r] x[ci][q]
// ------------------------------------------------------------------------------------------------
struct SatBfPlan { int ng, cs, kv; };     // groups per chunk, sub-blocks per chunk, virtual taps
static bool sat_bf_plan(int K, int stride, int mode, SatBfPlan* pl) {
    if (mode == 2 || stride > 1) {
        if (K != 2 * stride || (stride & (stride - 1)) || stride > 64) return false;
        *pl = SatBfPlan{8, 4, 2};
        return true;
    }
    if (stride != 1 || K < 1 || K > 8) return false;
    if (K == 1) *pl = SatBfPlan{4, 4, 1};
    else if (K == 2) *pl = SatBfPlan{8, 4, K};
    else if (K <= 4) *pl = SatBfPlan{8, 2, K};
    else *pl = SatBfPlan{8, 1, K};
    return true;
}
static int sat_log2i(int s) { int l = 0; while ((1 << l) < s) ++l; return l; }

// ------------------------------------------------------------------------------------------------
// weight preparation: torch weight w[D0][D1][K] (fp32) -> hi/lo bf16 planes [chunk][m (padded to 128)][group g][e]
// holding W'[m][v][tap] with v = (chunk*CS + g/KT)*8 + e, tap = g % KT (0 where tap/m/v are out of range):
//   mode 0 (conv, w = [out][in][K]):            stride 1: W' = w[m][v][tap];   stride S: W' = w[m][v/S][tap*S + v%S]
//   mode 1 (data-gradient of a stride-1 conv):  W'[m][v][tap] = w[v][m][K-1-tap]
//   mode 2 (conv_transpose, w = [in][out][K]):  W'[m][v][0] = w[v][m/S][m%S + S],  W'[m][v][1] = w[v][m/S][m%S]
// ------------------------------------------------------------------------------------------------
struct SatPackBfParams {
    const float* w;
    short* hi;
    short* lo;
    int D0, D1, K, S, mode, ng, cs, kv, m_v, v_v, out_pad;
    long long total;
};
__global__ void __launch_bounds__(256) sat_pack_bf16x3_kernel(SatPackBfParams p) {
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= p.total) return;
    const int e = (int)(o & 7);
    long long q = o >> 3;
    const int g = (int)(q % p.ng);
    q /= p.ng;
    const int m = (int)(q % p.out_pad), chunk = (int)(q / p.out_pad);
    const int kt = p.ng / p.cs;
    const int v = (chunk * p.cs + g / kt) * 8 + e, tap = g % kt;
    float val = 0.0f;
    if (tap < p.kv && m < p.m_v && v < p.v_v) {
        if (p.mode == 0) val = p.w[((size_t)m * p.D1 + v / p.S) * p.K + tap * p.S + v % p.S];
        else if (p.mode == 1) val = p.w[((size_t)v * p.D1 + m) * p.K + (p.K - 1 - tap)];
        else val = p.w[((size_t)v * p.D1 + m / p.S) * p.K + (m % p.S) + (tap == 0 ? p.S : 0)];
    }
    short h, l;
    sat_split2(val, &h, &l);
    p.hi[o] = h;
    p.lo[o] = l;
}

static bool sat_pack_bf_geometry(int D0, int D1, int K, int stride, int mode, SatPackBfParams* p) {
    SatBfPlan pl;
    if (mode < 0 || mode > 2 || D0 <= 0 || D1 <= 0 || !sat_bf_plan(K, stride, mode, &pl)) return false;
    if (mode == 1 && stride != 1) return false;
    p->D0 = D0; p->D1 = D1; p->K = K; p->S = stride; p->mode = mode;
    p->ng = pl.ng; p->cs = pl.cs; p->kv = pl.kv;
    if (mode == 0) { p->m_v = D0; p->v_v = D1 * stride; }
    else if (mode == 1) { p->m_v = D1; p->v_v = D0; }
    else { p->m_v = D1 * stride; p->v_v = D0; }
    p->out_pad = sat_cdiv(p->m_v, SAT_CO_T) * SAT_CO_T;
    p->total = (long long)sat_cdiv(p->v_v, 8 * pl.cs) * p->out_pad * pl.ng * 8;
    return true;
}
extern "C" long long sat_pack_weights_bf16x3_size(int D0, int D1, int K, int stride, int mode) {
    SatPackBfParams p{};
    return sat_pack_bf_geometry(D0, D1, K, stride, mode, &p) ? p.total : -1;
}
extern "C" int sat_pack_weights_bf16x3(const float* w, short* hi, short* lo, int D0, int D1, int K, int stride, int mode, void* stream) {
    SatPackBfParams p{};
    if (!sat_pack_bf_geometry(D0, D1, K, stride, mode, &p)) {
        sat_set_error("sat_pack_weights_bf16x3: needs stride 1 with K <= 8, or K == 2*stride with a power-of-two stride; mode 0|1|2");
        return 1;
    }
    p.w = w; p.hi = hi; p.lo = lo;
    SAT_LAUNCH(sat_pack_bf16x3_kernel, dim3((unsigned)sat_cdivll(p.total, 256)), dim3(256), stream, p);
    return sat_check_launch("sat_pack_weights_bf16x3");
}

// snake constants: a = e^alpha, ib = 1 / (e^beta + 1e-9)
struct SatSnakeConstParams { const float* alpha; const float* beta; float* a; float* ib; int C; };
__global__ void __launch_bounds__(256) sat_snake_consts_kernel(SatSnakeConstParams p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.C) return;
    p.a[i] = expf(p.alpha[i]);
    p.ib[i] = 1.0f / (expf(p.beta[i]) + 1e-9f);
}
extern "C" int sat_snake_consts(const float* alpha, const float* beta, float* a, float* ib, int C, void* stream) {
    if (C <= 0) { sat_set_error("sat_snake_consts: empty"); return 1; }
    SatSnakeConstParams p{alpha, beta, a, ib, C};
    SAT_LAUNCH(sat_snake_consts_kernel, dim3(sat_cdiv(C, 256)), dim3(256), stream, p);
    return sat_check_launch("sat_snake_consts");
}

#include "conv1d_bf16x3_k7.h"     // the pipelined kernel of the (8, 1) plan
#include "conv1d_planes.h"        // activation planes [B][Cin/8][rows][8] (pre-pass kernel; layout constants)
#include "conv1d_bf16x3_k7q.h"    // planes, 16-channel chunks (one tap per MFMA k-step), two wave rows one barrier apart

static int sat_bf_launch(const char* what, SatConvBfLaunch& a, const SatBfPlan& pl, void* stream) {
    if (pl.ng == 8 && pl.cs == 1 && a.sin_log2 == 0 && a.sout_log2 == 0) {
        if (a.xp_hi && a.wq) sat_bf_launch_k7q(a, stream);
        else sat_bf_launch_k7(a, stream);
        return sat_check_launch(what);
    }
    // the generic kernel stages the input's SnakeBeta constants of ALL channels in LDS: 1024 channels on the k = 1 plan (4, 4), 2048 on the others
    // (every Oobleck level: channels 128 x c_mults <= 16)
    if (a.p.alpha && a.p.Cin > ((pl.ng == 4) ? 1024 : 2048)) {
        sat_set_error("conv1d_bf16x3: an activated input of more than 2048 channels (1024 for K <= 4 at stride 1) is not served by the bf16x3 kernels (ops.use_bf16x3 = False takes the fp32-MFMA ones)");
        return 1;
    }
    dim3 grid(sat_cdiv(a.nq, SAT_T_T), a.cout_pad / SAT_CO_T, a.p.B);
    if (a.sin_log2 > 0) { SAT_LAUNCH((sat_conv1d_bf16x3_kernel<8, 4, true>), grid, dim3(256), stream, a); }     // strided: always plan (8, 4)
    else if (pl.ng == 8 && pl.cs == 1) { SAT_LAUNCH((sat_conv1d_bf16x3_kernel<8, 1, false>), grid, dim3(256), stream, a); }
    else if (pl.ng == 8 && pl.cs == 2) { SAT_LAUNCH((sat_conv1d_bf16x3_kernel<8, 2, false>), grid, dim3(256), stream, a); }
    else if (pl.ng == 8 && pl.cs == 4) { SAT_LAUNCH((sat_conv1d_bf16x3_kernel<8, 4, false>), grid, dim3(256), stream, a); }
    else { SAT_LAUNCH((sat_conv1d_bf16x3_kernel<4, 4, false>), grid, dim3(256), stream, a); }
    return sat_check_launch(what);
}

// Same contract as sat_conv1d (y[co][t] = sum W[co][ci][k] act(x)[ci][t*stride + k*dil - pad]), with the weights given
// as sat_pack_weights_bf16x3 planes (mode 0, or mode 1 for a stride-1 data-gradient) and the SnakeBeta constants given
// pre-exponentiated (sat_snake_consts), or NULL for no activation.  stride 1: K <= 8; stride S: K == 2S, S a power of two.
static int sat_conv1d_bf16x3_impl(const char* what, const float* x, const short* w_hi, const short* w_lo, const float* bias,
                                  const float* snake_a, const float* snake_ib, const float* res, float* y,
                                  const float* x2, const float* alpha2, const float* beta2, float* part_da,
                                  float* part_db, int B, int Cin, int Cout, int Tin, int Tout, int K, int stride, int dil,
                                  int pad, int tanh_out, short* em_hi, short* em_lo, const float* em_a, const float* em_ib, int em_rows,
                                  void* stream) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || Tin <= 0 || Tout <= 0) { sat_set_error("sat_conv1d_bf16x3: empty shape"); return 1; }
    SatBfPlan pl;
    if (!sat_bf_plan(K, stride, 0, &pl) || dil < 1 || (stride > 1 && dil != 1)) {
        sat_set_error("sat_conv1d_bf16x3: needs stride 1 with K <= 8, or K == 2*stride (power-of-two stride, dilation 1)");
        return 1;
    }
    if ((pl.kv - 1) * dil > (pl.cs == 1 ? 62 : SAT_BF_AROWSN - SAT_T_T)) { sat_set_error("sat_conv1d_bf16x3: receptive field too large for the LDS slab"); return 1; }
    if ((snake_a == nullptr) != (snake_ib == nullptr)) { sat_set_error("sat_conv1d_bf16x3: snake constants must both be given"); return 1; }
    if (x2 && (!alpha2 || !beta2 || !part_da || !part_db)) { sat_set_error("sat_conv1d_bf16x3: backward epilogue needs alpha2/beta2/partials"); return 1; }
    SatConvBfLaunch a;
    a.p = SatConvParams{x, nullptr, bias, snake_a, snake_ib, res, y, x2, alpha2, beta2, part_da, part_db,
                        B, Cin, Cout, Tin, Tout, pl.kv, 1, dil, stride == 1 ? pad : 0, tanh_out};
    a.w_hi = w_hi;
    a.w_lo = w_lo;
    a.cout_v = Cout;
    a.cout_pad = sat_cdiv(Cout, SAT_CO_T) * SAT_CO_T;
    a.cin_v = Cin * stride;
    a.sin_log2 = sat_log2i(stride);
    a.sout_log2 = 0;
    a.in_shift = stride == 1 ? 0 : pad;
    a.out_shift = 0;
    a.nq = Tout;
    if (em_hi) {
        // plane emission lives in the generic kernel's 16-byte epilogue
        const bool generic = !(pl.ng == 8 && pl.cs == 1);
        const bool vec4 = (Tout & 3) == 0 && (((uintptr_t)y | (uintptr_t)x2 | (uintptr_t)res) & 15) == 0;
        if (!generic || !vec4 || !em_lo || (em_a == nullptr) != (em_ib == nullptr) || em_rows < SAT_K7P_LEAD_ROWS + Tout ||
            (((uintptr_t)em_hi | (uintptr_t)em_lo) & 15)) {
            sat_set_error("sat_conv1d_bf16x3_emit: emission needs K <= 4 or a strided conv, Tout % 4 == 0, 16-byte aligned tensors, rows >= 32 + Tout");
            return 1;
        }
        a.em_hi = em_hi; a.em_lo = em_lo; a.em_a = em_a; a.em_ib = em_ib; a.em_rows = em_rows; a.em_c8 = sat_cdiv(Cout, 8);
    }
    return sat_bf_launch(what, a, pl, stream);
}
extern "C" int sat_conv1d_bf16x3(const float* x, const short* w_hi, const short* w_lo, const float* bias,
                                 const float* snake_a, const float* snake_ib, const float* res, float* y,
                                 const float* x2, const float* alpha2, const float* beta2, float* part_da,
                                 float* part_db, int B, int Cin, int Cout, int Tin, int Tout, int K, int stride, int dil,
                                 int pad, int tanh_out, void* stream) {
    return sat_conv1d_bf16x3_impl("sat_conv1d_bf16x3", x, w_hi, w_lo, bias, snake_a, snake_ib, res, y, x2, alpha2, beta2, part_da, part_db,
                                  B, Cin, Cout, Tin, Tout, K, stride, dil, pad, tanh_out, nullptr, nullptr, nullptr, nullptr, 0, stream);
}
// sat_conv1d_bf16x3 that ALSO writes act(y) as the activation planes of the k7 conv that consumes y next (sat_conv1d_bf16x3_planes /
// _planesq): em_hi / em_lo [B][ceil(Cout/8)][em_rows][8] bf16 (row 32 + t; the caller keeps the rows around the sequence zero),
// em_a / em_ib the consumer's pre-exponentiated SnakeBeta constants (sat_snake_consts) or NULL for planes of y itself.  The k = 1 /
// K <= 4 / strided plans only (the producers of a ResidualUnit's input), Tout % 4 == 0.
extern "C" int sat_conv1d_bf16x3_emit(const float* x, const short* w_hi, const short* w_lo, const float* bias,
                                      const float* snake_a, const float* snake_ib, const float* res, float* y,
                                      const float* x2, const float* alpha2, const float* beta2, float* part_da,
                                      float* part_db, int B, int Cin, int Cout, int Tin, int Tout, int K, int stride, int dil,
                                      int pad, int tanh_out, void* em_hi, void* em_lo, const float* em_a, const float* em_ib, int em_rows,
                                      void* stream) {
    if (!em_hi) { sat_set_error("sat_conv1d_bf16x3_emit: planes missing"); return 1; }
    return sat_conv1d_bf16x3_impl("sat_conv1d_bf16x3_emit", x, w_hi, w_lo, bias, snake_a, snake_ib, res, y, x2, alpha2, beta2, part_da, part_db,
                                  B, Cin, Cout, Tin, Tout, K, stride, dil, pad, tanh_out, (short*)em_hi, (short*)em_lo, em_a, em_ib, em_rows, stream);
}

// ---- the k = 5..7 stride-1 convs from pre-split activation planes (conv1d_planes.h, conv1d_bf16x3_k7q.h) ----
// rows of one (batch, 8-channel chunk) plane: 32 zero rows, the sequence, zero rows up to the last window's halo
extern "C" int sat_conv1d_k7_plane_rows(int Tin, int Tout, int pad) {
    if (Tin <= 0 || Tout <= 0 || pad < 0 || pad > SAT_K7P_LEAD) return -1;
    const int need = sat_cdiv(Tout, SAT_K7_T) * SAT_K7_T - pad + SAT_K7_AROWS;   // last window: rows [t0 - pad, t0 - pad + 320)
    const int body = need > Tin ? need : Tin;
    return SAT_K7P_LEAD + sat_cdiv(body, 64) * 64;
}
// x (B, Cin, Tin) fp32 -> act(x) as bf16 hi / lo planes [B][ceil(Cin/8)][rows][8]  (act = SnakeBeta with pre-exponentiated constants, or none)
extern "C" int sat_conv1d_k7_planes(const float* x, const float* snake_a, const float* snake_ib, short* xp_hi, short* xp_lo, int B,
                                    int Cin, int Tin, int rows, void* stream) {
    if (B <= 0 || Cin <= 0 || Tin <= 0 || rows < SAT_K7P_LEAD + Tin) { sat_set_error("sat_conv1d_k7_planes: bad shape"); return 1; }
    if ((snake_a == nullptr) != (snake_ib == nullptr)) { sat_set_error("sat_conv1d_k7_planes: snake constants must both be given"); return 1; }
    SatK7PlaneParams p{x, snake_a, snake_ib, xp_hi, xp_lo, B, Cin, Tin, rows, sat_cdiv(Cin, 8)};
    SAT_LAUNCH(sat_k7_planes_kernel, dim3(sat_cdiv(rows, 256), p.c8, B), dim3(256), stream, p);
    return sat_check_launch("sat_conv1d_k7_planes");
}
// sat_conv1d_bf16x3 for stride 1, 5 <= K <= 7, with the (activated) input given as planes (sat_conv1d_k7_planes or a producer's emission)
// and the weights packed by sat_pack_weights_k7q (mode 0, or mode 1 for a data-gradient): conv1d_bf16x3_k7q.h.
extern "C" int sat_conv1d_bf16x3_planesq(const short* xp_hi, const short* xp_lo, int rows, const short* w_hi, const short* w_lo,
                                         const float* bias, const float* res, float* y, const float* x2, const float* alpha2,
                                         const float* beta2, float* part_da, float* part_db, int B, int Cin, int Cout, int Tin, int Tout,
                                         int K, int dil, int pad, int tanh_out, int flags, void* stream) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || Tin <= 0 || Tout <= 0) { sat_set_error("sat_conv1d_bf16x3_planesq: empty shape"); return 1; }
    if (K < 5 || K > SAT_K7Q_TAPS || dil < 1 || (K - 1) * dil > 62) {
        sat_set_error("sat_conv1d_bf16x3_planesq: needs stride 1, 5 <= K <= 7, (K-1)*dil <= 62");
        return 1;
    }
    if (rows < sat_conv1d_k7_plane_rows(Tin, Tout, pad)) { sat_set_error("sat_conv1d_bf16x3_planesq: rows must be >= sat_conv1d_k7_plane_rows(Tin, Tout, pad)"); return 1; }
    if (x2 && (!alpha2 || !beta2 || !part_da || !part_db)) { sat_set_error("sat_conv1d_bf16x3_planesq: backward epilogue needs alpha2/beta2/partials"); return 1; }
    SatConvBfLaunch a;
    a.p = SatConvParams{nullptr, nullptr, bias, nullptr, nullptr, res, y, x2, alpha2, beta2, part_da, part_db,
                        B, Cin, Cout, Tin, Tout, K, 1, dil, pad, tanh_out};
    a.w_hi = w_hi;
    a.w_lo = w_lo;
    a.cout_v = Cout;
    a.cout_pad = sat_cdiv(Cout, SAT_CO_T) * SAT_CO_T;
    a.cin_v = Cin;
    a.sin_log2 = 0;
    a.sout_log2 = 0;
    a.in_shift = 0;
    a.out_shift = 0;
    a.nq = Tout;
    a.xp_hi = xp_hi;
    a.xp_lo = xp_lo;
    a.xp_rows = rows;
    a.xp_c8 = sat_cdiv(Cin, 8);
    a.wq = 1;
    a.dma_in_mfma = (flags & 4) ? 1 : 0;                     // flags bit 2: VARIANT 3 of the k7q kernel
    a.persist = (flags & 1) ? 0 : ((flags & 2) ? 2 : 1);      // flags bit 0: one workgroup per tile (round 5's launch; A/B runs and the tests'
                                                              // second arm); bit 1: persistent at any size (the tests: small shapes)
    SatBfPlan pl{8, 1, K};
    return sat_bf_launch("sat_conv1d_bf16x3_planesq", a, pl, stream);
}

// The whole ResidualUnit forward in one launch (autoencoders.py:58-83), C <= 128 channels:
//   h = conv7(planes of snake1(x)) + bias1          (stored to `h` if given: the backward needs it)
//   y = x + conv1(snake2(h)) + bias2                 (+ optionally written as the NEXT unit's activation planes: em_*)
// xp_hi / xp_lo: activation planes of snake1(x) (sat_conv1d_k7_planes or a producer's emission); w7_*: sat_pack_weights_k7q(K, mode 0);
// w1_*: sat_pack_weights_k7q of the (C, C, 1) weight (K = 1, mode 0); a2 / ib2: sat_snake_consts of the second activation.
extern "C" int sat_residual_unit_fwd(const short* xp_hi, const short* xp_lo, int rows, const short* w7_hi, const short* w7_lo,
                                     const float* bias1, const float* a2, const float* ib2, const short* w1_hi, const short* w1_lo,
                                     const float* bias2, const float* x, float* h, float* y, int B, int C, int T, int K, int dil, int pad,
                                     void* em_hi, void* em_lo, const float* em_a, const float* em_ib, int em_rows, void* stream) {
    if (B <= 0 || C <= 0 || T <= 0 || C > SAT_K7_CO) { sat_set_error("sat_residual_unit_fwd: needs 1 <= C <= 128"); return 1; }
    if (K < 5 || K > SAT_K7Q_TAPS || dil < 1 || (K - 1) * dil > 62 || 2 * pad != (K - 1) * dil) {
        sat_set_error("sat_residual_unit_fwd: needs 5 <= K <= 7, (K-1)*dil <= 62, 'same' padding");
        return 1;
    }
    if (rows < sat_conv1d_k7_plane_rows(T, T, pad)) { sat_set_error("sat_residual_unit_fwd: rows must be >= sat_conv1d_k7_plane_rows(T, T, pad)"); return 1; }
    if (!xp_hi || !xp_lo || !w7_hi || !w7_lo || !w1_hi || !w1_lo || !a2 || !ib2 || !x || !y) { sat_set_error("sat_residual_unit_fwd: missing operand"); return 1; }
    if ((T & 3) || (((uintptr_t)y | (uintptr_t)x) & 15)) { sat_set_error("sat_residual_unit_fwd: T % 4 == 0 and 16-byte aligned x / y"); return 1; }
    SatConvBfLaunch a;
    a.p = SatConvParams{nullptr, nullptr, bias1, nullptr, nullptr, x, y, nullptr, nullptr, nullptr, nullptr, nullptr,
                        B, C, C, T, T, K, 1, dil, pad, 0};
    a.w_hi = w7_hi;
    a.w_lo = w7_lo;
    a.cout_v = C;
    a.cout_pad = SAT_CO_T;
    a.cin_v = C;
    a.sin_log2 = 0; a.sout_log2 = 0; a.in_shift = 0; a.out_shift = 0;
    a.nq = T;
    a.xp_hi = xp_hi; a.xp_lo = xp_lo; a.xp_rows = rows; a.xp_c8 = sat_cdiv(C, 8);
    a.wq = 1;
    a.ru_w1_hi = w1_hi; a.ru_w1_lo = w1_lo; a.ru_bias2 = bias2; a.ru_a2 = a2; a.ru_ib2 = ib2; a.ru_h = h;
    if (em_hi) {
        if (!em_lo || (em_a == nullptr) != (em_ib == nullptr) || em_rows < SAT_K7P_LEAD_ROWS + T || (((uintptr_t)em_hi | (uintptr_t)em_lo) & 15)) {
            sat_set_error("sat_residual_unit_fwd: bad emission planes");
            return 1;
        }
        a.em_hi = (short*)em_hi; a.em_lo = (short*)em_lo; a.em_a = em_a; a.em_ib = em_ib; a.em_rows = em_rows; a.em_c8 = sat_cdiv(C, 8);
    }
    SatBfPlan pl{8, 1, K};
    return sat_bf_launch("sat_residual_unit_fwd", a, pl, stream);
}

// Same contract as sat_convtr1d (y[co][q*stride + k - pad] += W[ci][co][k] act(x)[ci][q], K == 2*stride, power-of-two
// stride), weights as sat_pack_weights_bf16x3 planes (mode 2).
extern "C" int sat_convtr1d_bf16x3(const float* x, const short* w_hi, const short* w_lo, const float* bias,
                                   const float* snake_a, const float* snake_ib, const float* res, float* y,
                                   const float* x2, const float* alpha2, const float* beta2, float* part_da,
                                   float* part_db, int B, int Cin, int Cout, int Tin, int Tout, int K, int stride,
                                   int pad, int tanh_out, void* stream) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || Tin <= 0 || Tout <= 0) { sat_set_error("sat_convtr1d_bf16x3: empty shape"); return 1; }
    SatBfPlan pl;
    if (!sat_bf_plan(K, stride, 2, &pl) || pad < 0) { sat_set_error("sat_convtr1d_bf16x3: needs K == 2*stride with a power-of-two stride"); return 1; }
    if ((snake_a == nullptr) != (snake_ib == nullptr)) { sat_set_error("sat_convtr1d_bf16x3: snake constants must both be given"); return 1; }
    if (x2 && (!alpha2 || !beta2 || !part_da || !part_db)) { sat_set_error("sat_convtr1d_bf16x3: backward epilogue needs alpha2/beta2/partials"); return 1; }
    SatConvBfLaunch a;
    a.p = SatConvParams{x, nullptr, bias, snake_a, snake_ib, res, y, x2, alpha2, beta2, part_da, part_db,
                        B, Cin, Cout, Tin, Tout, 2, 1, 1, 1, tanh_out};
    a.w_hi = w_hi;
    a.w_lo = w_lo;
    a.cout_v = Cout * stride;
    a.cout_pad = sat_cdiv(a.cout_v, SAT_CO_T) * SAT_CO_T;
    a.cin_v = Cin;
    a.sin_log2 = 0;
    a.sout_log2 = sat_log2i(stride);
    a.in_shift = 0;
    a.out_shift = pad;
    a.nq = sat_cdiv(Tout + pad, stride);
    return sat_bf_launch("sat_convtr1d_bf16x3", a, pl, stream);
}
// rows of the snake-gradient partial-sum planes written by sat_conv1d_bf16x3 (one per batch item and time tile)
extern "C" int sat_conv1d_bf16x3_partial_rows(int B, int Tout, int K, int stride) {
    SatBfPlan pl;
    if (B <= 0 || Tout <= 0 || !sat_bf_plan(K, stride, 0, &pl)) return -1;
    const bool k7 = pl.ng == 8 && pl.cs == 1 && stride == 1;
    return B * sat_cdiv(Tout, k7 ? SAT_K7_T : SAT_T_T);
}
extern "C" int sat_convtr1d_bf16x3_partial_rows(int B, int Tout, int stride, int pad) {
    if (B <= 0 || Tout <= 0 || stride < 1 || pad < 0) return -1;
    return B * sat_cdiv(sat_cdiv(Tout + pad, stride), SAT_T_T);
}
