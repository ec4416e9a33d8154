// conv_wgrad7_bf16x3_pipe.h — weight gradient of the k = 7 stride-1 (dilated) convs as an 8-wave double-buffered
// pipeline.  Included by conv_wgrad_bf16x3.hip (shares SatWgBfParams, sat_alignbit and the fragment arithmetic of
// sat_wgrad7_bf16x3_kernel, which stays as the small-shape variant).
//
//   dW[co][ci][tap] = sum_b sum_t  dy[b][co][t] * snake(x)[b][ci][t + tap*dil - pad]
//
// The 4-wave kernel's 128(co) x 32(ci) tile re-reads the dy slab once per 32 input channels; its L2->LDS traffic (47 KB
// per 64 time steps and workgroup), not the matrix pipe, bounds it.  Here one workgroup = 8 waves = 128(co) x 64(ci) x 7
// taps (wave w: co rows 32*(w&3), ci columns 32*(w>>2); 7 accumulator tiles each), one workgroup per CU (143 KB LDS):
// the dy slab is staged once for 64 input channels (31 KB per MFMA-equivalent instead of 47).  Per 64-step stage a phase
// issues the global loads of stage c+2 into a register set, runs stage c's 84 MFMAs per wave out of LDS buffer c & 1 and
// converts stage c+1 (hi/lo split; SnakeBeta on x) into the other buffer — ONE barrier per stage.  Waves 0-3 run the
// MFMAs first, waves 4-7 the conversion first: wave w and w+4 share a SIMD (and their dy fragments).
#pragma once

#define SAT_WP_NT 512
#define SAT_WP_NI 64                 // input channels per workgroup

template <int DIL>
__global__ void __launch_bounds__(SAT_WP_NT) sat_wgrad7_bf16x3_pipe_kernel(SatWgBfParams p) {
    constexpr int NCH = (6 * DIL + 7) / 8 + 1;                       // aligned 8-element chunks covering all 7 taps
    constexpr int HSPAN = SAT_WB_TT + 6 * DIL;                       // activation samples needed per stage
    constexpr int HP = HSPAN / 2;                                    // ... in pairs (HSPAN is even)
    constexpr int NXU = (HP + 15) / 16;                              // x staging: 16 threads per row walk its pairs 16 at a time,
    constexpr int NXP = 2 * NXU;                                     // two passes of 32 rows -> pairs per thread and stage (<= 8)
    constexpr int NDY = SAT_CO_T * SAT_WB_TT / 4 / SAT_WP_NT;        // dy float4 per thread and stage (4)
    __shared__ __attribute__((aligned(16))) short lo_lds0[2][SAT_CO_T][SAT_WB_LOROW], lo_lds1[2][SAT_CO_T][SAT_WB_LOROW];   // dy  [plane][co][t]
    __shared__ __attribute__((aligned(16))) short hi_lds0[2][SAT_WP_NI][SAT_WB_HIROW], hi_lds1[2][SAT_WP_NI][SAT_WB_HIROW]; // act [plane][ci][t]
    __shared__ float sn_a[SAT_WP_NI], sn_ib[SAT_WP_NI];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    int mn_tile, split;
    sat_xcd_tile(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y, gridDim.z, &mn_tile, &split);
    const int m0 = (mn_tile % (int)gridDim.x) * SAT_CO_T, n0 = (mn_tile / (int)gridDim.x) * SAT_WP_NI;
    const int m_w = (wave & 3) * 32, n_w = (wave >> 2) * 32;
    const bool mfma_first = wave < 4;
    const bool snake = p.alpha != nullptr;

    f32x16 acc[7];
#pragma unroll
    for (int k = 0; k < 7; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;

    if (tid < SAT_WP_NI) {
        const int c = n0 + tid;
        float sa = 1.f, sib = 0.f;
        if (snake && c < p.N) {
            sa = expf(p.alpha[c]);
            sib = 1.0f / (expf(p.beta[c]) + 1e-9f);
        }
        sn_a[tid] = sa;
        sn_ib[tid] = sib;
    }
    // hi-tile columns past the staged span are read by the chunk overrun of the last k-step: keep them zero
    for (int i = tid; i < 2 * SAT_WP_NI * SAT_WB_HIROW; i += SAT_WP_NT) {
        (&hi_lds0[0][0][0])[i] = 0;
        (&hi_lds1[0][0][0])[i] = 0;
    }

    const int c_begin = split * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > p.nchunks) c_end = p.nchunks;
    const int nst = c_end - c_begin;                                 // stages of this workgroup (>= 1)
    auto clampc = [&](int c) { return c < nst - 1 ? c : nst - 1; };  // (redundant reloads past the end keep phases branch-free)

    // ONE staging register set: a phase converts it into LDS and only then refills it with the loads of the stage after
    // next (MFMA-first waves: MFMA, convert, load; the others: convert, load, MFMA) — either way the data has a whole
    // MFMA phase (~1 us) to arrive before it is converted, and 32 registers fewer are live (7 accumulator tiles = 112)
    float4 dyv[1][NDY];
    float xv[1][NXP][2];
    auto issue_loads = [&](int c) {
        constexpr int st = 0;
        const int ch = c_begin + c;
        const int b = ch / p.nT;
        const int tt0 = (ch - b * p.nT) * SAT_WB_TT;
        // dy: 16 threads per row, one float4 each (T % 4 == 0 is a launch condition: a float4 is all in or all out)
        const float* sdy = p.dy + (size_t)b * p.M * p.T;
        const int c4 = (tid & 15) * 4;
        const bool t_ok = tt0 + c4 < p.T;
        // INTERIOR stages (every sample the stage reads lies inside the sequence — all but the first and the last of a batch item; block-uniform):
        // unconditional loads from clamped rows, the activation pairs as 8-byte loads at dword alignment.  No select and no branch behind a
        // load: rows past M / N and pairs past HP carry finite garbage that is never stored (write_lds skips pi >= HP; dW rows / columns
        // past M / N are not written out).  Round 6: the ablation of this kernel put its global loads at 25 % of the launch — sixteen
        // `s_cbranch_execz`-wrapped 4-byte loads per thread and stage (profiles/r06_experiments/wgrad7_ablation/).
        if (tt0 - p.pad >= 0 && tt0 - p.pad + HSPAN <= p.T && tt0 + SAT_WB_TT <= p.T) {
            typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
#pragma unroll
            for (int u = 0; u < NDY; ++u) {
                int m = m0 + (tid >> 4) + u * 32;
                m = m < p.M ? m : p.M - 1;
                dyv[st][u] = *reinterpret_cast<const float4*>(sdy + (size_t)m * p.T + tt0 + c4);
            }
            const float* sx = p.x + (size_t)b * p.N * p.T + (tt0 - p.pad);
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                int n = n0 + (tid >> 4) + 32 * v;
                n = n < p.N ? n : p.N - 1;
                const float* s = sx + (size_t)n * p.T;
#pragma unroll
                for (int u = 0; u < NXU; ++u) {
                    int pi = (tid & 15) + 16 * u;
                    pi = pi < HP ? pi : HP - 1;
                    const f2u q = *reinterpret_cast<const f2u*>(s + 2 * pi);
                    xv[st][v * NXU + u][0] = q[0];
                    xv[st][v * NXU + u][1] = q[1];
                }
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < NDY; ++u) {
            const int m = m0 + (tid >> 4) + u * 32;
            const bool ok = t_ok && m < p.M;
            const float4 q = *reinterpret_cast<const float4*>(sdy + (size_t)(ok ? m : 0) * p.T + (ok ? tt0 + c4 : 0));
            dyv[st][u] = ok ? q : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // x: 16 threads per row, pair (tid & 15) + 16 u of row (tid >> 4) + 32 v
        const float* sx = p.x + (size_t)b * p.N * p.T;
        const int th0 = tt0 - p.pad;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int n = n0 + (tid >> 4) + 32 * v;
            const float* s = sx + (size_t)(n < p.N ? n : 0) * p.T;
#pragma unroll
            for (int u = 0; u < NXU; ++u) {
                const int pi = (tid & 15) + 16 * u;
                const int t = th0 + 2 * pi;
                const bool ok = pi < HP && n < p.N;
                xv[st][v * NXU + u][0] = (ok && t >= 0 && t < p.T) ? s[t] : 0.0f;
                xv[st][v * NXU + u][1] = (ok && t + 1 >= 0 && t + 1 < p.T) ? s[t + 1] : 0.0f;
            }
        }
    };
    auto write_lds = [&](auto buf_c) {
        constexpr int st = 0;
        auto& lo_lds = sat_pick<decltype(buf_c)::value>(lo_lds0, lo_lds1);
        auto& hi_lds = sat_pick<decltype(buf_c)::value>(hi_lds0, hi_lds1);
#pragma unroll
        for (int u = 0; u < NDY; ++u) {
            const int row = (tid >> 4) + u * 32, c4 = (tid & 15) * 4;
            uint32_t h0, h1, l0, l1;
            sat_split2_pk(dyv[st][u].x, dyv[st][u].y, &h0, &l0);
            sat_split2_pk(dyv[st][u].z, dyv[st][u].w, &h1, &l1);
            typedef uint32_t u2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u2*>(&lo_lds[0][row][c4]) = u2{h0, h1};
            *reinterpret_cast<u2*>(&lo_lds[1][row][c4]) = u2{l0, l1};
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int row = (tid >> 4) + 32 * v;
            float sa = 0.f, sib = 0.f;
            if (snake) { sa = sn_a[row]; sib = sn_ib[row]; }
#pragma unroll
            for (int u = 0; u < NXU; ++u) {
                const int pi = (tid & 15) + 16 * u;
                if (pi < HP) {
                    float o0 = xv[st][v * NXU + u][0], o1 = xv[st][v * NXU + u][1];
                    if (snake) {
                        o0 = sat_snake(o0, sa, sib);
                        o1 = sat_snake(o1, sa, sib);
                    }
                    uint32_t h, l;
                    sat_split2_pk(o0, o1, &h, &l);
                    *reinterpret_cast<uint32_t*>(&hi_lds[0][row][2 * pi]) = h;
                    *reinterpret_cast<uint32_t*>(&hi_lds[1][row][2 * pi]) = l;
                }
            }
        }
    };
    auto mfma_phase = [&](auto buf_c) {
        auto& lo_lds = sat_pick<decltype(buf_c)::value>(lo_lds0, lo_lds1);
        auto& hi_lds = sat_pick<decltype(buf_c)::value>(hi_lds0, hi_lds1);
        SAT_MFMA_PRIO(1);
#pragma unroll 2
        for (int ks = 0; ks < SAT_WB_TT / 16; ++ks) {
            const int tb = 16 * ks + 8 * hi;
            bf16x8 af[2];
            af[0] = *reinterpret_cast<const bf16x8*>(&lo_lds[0][m_w + l31][tb]);
            af[1] = *reinterpret_cast<const bf16x8*>(&lo_lds[1][m_w + l31][tb]);
            // one activation plane at a time (halves the live chunk registers): hi plane pairs with dy hi + lo, lo plane with dy hi.
            // (Reading each tap's fragment as an unaligned 16-byte LDS load works on gfx950 but measured 20 % slower.)
            {
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    u32x4 cw[NCH];
#if !defined(SAT_HIPEMU)
                    asm volatile("" ::: "memory");
#endif
#pragma unroll
                    for (int j = 0; j < NCH; ++j) cw[j] = *reinterpret_cast<const u32x4*>(&hi_lds[pl][n_w + l31][tb + 8 * j]);
#pragma unroll
                    for (int k = 0; k < 7; ++k) {
                        const int off = k * DIL;
                        const int wbase = (off >> 3) * 4 + ((off & 7) >> 1);
                        const bool odd = (off & 1) != 0;
                        u32x4 r;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int w0 = wbase + i, w1 = wbase + i + 1;
                            const unsigned a0 = cw[w0 >> 2][w0 & 3];
                            if (odd) r[i] = sat_alignbit(cw[w1 >> 2][w1 & 3], a0, 16);
                            else r[i] = a0;
                        }
                        const bf16x8 bf = __builtin_bit_cast(bf16x8, r);
                        acc[k] = sat_mfma_32x32x16_bf16(af[0], bf, acc[k]);
                        if (pl == 0) acc[k] = sat_mfma_32x32x16_bf16(af[1], bf, acc[k]);
                    }
                }
            }
        }
        SAT_MFMA_PRIO(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto phase = [&](auto bc, auto bn, int next) {         // MFMAs on buffer bc; registers -> buffer bn; refill with stage `next`
        if (!mfma_first) {
            write_lds(bn);
            issue_loads(next);
        }
        mfma_phase(bc);
        if (mfma_first) {
            write_lds(bn);
            issue_loads(next);
        }
    };

    // prologue: stage 0 into buffer 0, stage 1's data in flight in the registers
    issue_loads(0);
    __syncthreads();                                       // zero fill + snake constants visible
    write_lds(I0{});
    issue_loads(clampc(1));
    __syncthreads();
    // ONE loop body for every stage (round 6): the last stages run the same two phases with clamped (redundant) refills instead of a
    // separate remainder — three more inlined copies of the MFMA phase, in which the register allocator spilled 34-172 registers
    // (tools/check_resources.py allow-listed them: scratch outside the steady-state loop, but scratch)
    for (int c = 0; c < nst; c += 2) {
        phase(I0{}, I1{}, clampc(c + 2));                  // stage c out of buffer 0; stage c+1 -> buffer 1
        __syncthreads();
        if (c + 1 < nst) {                                 // (block-uniform)
            phase(I1{}, I0{}, clampc(c + 3));              // stage c+1 out of buffer 1; stage c+2 -> buffer 0
            __syncthreads();
        }
    }

    if (m0 + m_w < p.M) {
        float* ob = p.out + (size_t)split * p.so_split;
        const int n = n0 + n_w + l31;
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + m_w + (r & 3) + 8 * (r >> 2) + 4 * hi;
                if (m < p.M && n < p.N) ob[(size_t)m * p.so_m + (size_t)n * p.so_n + (size_t)k * p.so_k] = acc[k][r];
            }
    }
}
