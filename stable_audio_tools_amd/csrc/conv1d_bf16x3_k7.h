// conv1d_bf16x3_k7.h — the k = 5..8 stride-1 (dilated) convolutions of the ResidualUnits (autoencoders.py:58-83) and
// their data-gradients: the plan (NG = 8, CS = 1) of conv1d_bf16x3.hip as a double-buffered pipeline.
// Included by conv1d_bf16x3.hip (shares SatConvBfLaunch, the weight planes and the epilogue conventions).
//
// One workgroup = 8 waves = 128 (co) x 256 (t) outputs, each wave a 64 x 64 block (2 x 2 MFMA tiles), one workgroup per
// CU (100 KB of LDS).  Per K-chunk (8 input channels x 8 tap groups) a phase does three independent things:
//   (1) issue the global loads of chunk c+2 into a register set (activations: 5 elements per thread, the channel is
//       wave-uniform; weights: four 16-byte pieces per thread),
//   (2) run chunk c's 48 MFMAs per wave out of LDS buffer c & 1,
//   (3) convert chunk c+1 (SnakeBeta, hi/lo split) from the other register set into LDS buffer (c+1) & 1,
// then ONE barrier.  Waves 0-3 do (2) then (3), waves 4-7 do (3) then (2): wave w and wave w+4 share a SIMD, so the
// matrix pipe of every SIMD is fed by one of them while the other does its VALU / LDS-store work.
// The SnakeBeta constants of a chunk travel one phase ahead of its data through a two-slot LDS table.
#pragma once

#define SAT_K7_CO 128
#define SAT_K7_T 256
#define SAT_K7_NT 512
#define SAT_K7_AROWS 320          // 256 main rows + up to 62 halo rows + scratch rows; the last one is the dummy row
#define SAT_K7_KROW 72            // 8 groups x 8 + 8 pad bf16 per weight row (144 B: conflict-free b128 reads)

template <bool SNAKE, bool EXACT>
__global__ void __launch_bounds__(SAT_K7_NT) sat_conv1d_bf16x3_k7_kernel(SatConvBfLaunch a) {
    constexpr int CO_T = SAT_K7_CO, T_T = SAT_K7_T, NT = SAT_K7_NT, AROWS = SAT_K7_AROWS, KROW = SAT_K7_KROW;
    constexpr int DUMMY = AROWS - 1;
    constexpr int TW = T_T / 64;                          // waves along time
    constexpr int NEL = 8 * AROWS / NT;                   // activation elements per thread and chunk (5)
    constexpr int NWV = 2 * CO_T * 8 / NT;                // 16-byte weight pieces per thread and chunk (4)
    const SatConvParams& p = a.p;
    // two buffers each, as separate objects so that the compiler knows reads of one and writes of the other never alias
    __shared__ __attribute__((aligned(16))) short w_lds0[2][CO_T][KROW], w_lds1[2][CO_T][KROW];   // [plane][co][g*8+e]
    __shared__ __attribute__((aligned(16))) short a_lds0[2][AROWS][8], a_lds1[2][AROWS][8];       // [plane][time row][8 ci]
    __shared__ float c_lds0[2][8], c_lds1[2][8];          // [a | ib][channel of the chunk], slot = chunk & 1
    __shared__ float red_lds[2][TW][CO_T];
    __shared__ float ep_lds[3][CO_T];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    // grid = (channel tiles, time tiles, B): the channel tiles of one activation window share an XCD (sat_xcd_tile)
    int co_tile, win;
    sat_xcd_tile(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x, gridDim.y * gridDim.z, &co_tile, &win);
    const int b = win / (int)gridDim.y, t_tile = win - b * (int)gridDim.y;
    const int co0 = co_tile * CO_T;
    const int t0 = t_tile * T_T;
    const int co_w = (wave / TW) * 64, t_w = (wave % TW) * 64;
    const int K = p.K, dil = p.dil;
    const int nrows = T_T + (K - 1) * dil;
    const int q_in0 = t0 - p.pad;
    const float* xb = p.x + (size_t)b * p.Cin * p.Tin;
    const bool mfma_first = wave < 4;
    const bool wave_on_co = (co0 + co_w) < a.cout_v;      // a wave whose 64 output channels are all past Cout (Cout <= 64: the discriminator's
                                                          // convs) skips its MFMAs — its accumulators stay zero and its epilogue is skipped

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    if (tid < CO_T) {
        const int m = co0 + tid;
        const bool ok = m < a.cout_v;
        ep_lds[0][tid] = (ok && p.bias) ? p.bias[m] : 0.0f;
        ep_lds[1][tid] = (ok && p.x2) ? expf(p.alpha2[m]) : 1.0f;
        ep_lds[2][tid] = (ok && p.x2) ? expf(p.beta2[m]) : 1.0f;
    }

    const int nchunks = (a.cin_v + 7) / 8;
    const int last = nchunks - 1;
    auto clampc = [&](int c) { return c < last ? c : last; };        // (redundant reloads past the end keep the phases branch-free)

    // ---- staging: thread's element `it` is channel e = id / AROWS (wave-uniform: AROWS = 5 waves), row id % AROWS ----
    float ev[2][NEL];
    bf16x8 wv[2][NWV];
    float cv[2] = {0.0f, 0.0f};
    const int c_idx = tid < 16 ? tid : 15;                            // (every thread takes part: no divergent branch)
    auto load_consts = [&](int c) -> float {
        int ch = c * 8 + (c_idx & 7);
        ch = ch < p.Cin ? ch : p.Cin - 1;
        return (c_idx >= 8) ? p.beta[ch] : p.alpha[ch];               // pre-exponentiated: a, 1/(b + 1e-9)
    };
    auto issue_loads = [&](int c, auto set_c) {
        constexpr int st = decltype(set_c)::value;
#pragma unroll
        for (int it = 0; it < NEL; ++it) {
            const int id = tid + it * NT;
            const int e = SAT_UNIFORM(id / AROWS);
            int ch = c * 8 + e;
            if (!EXACT) ch = ch < p.Cin ? ch : p.Cin - 1;             // past the end: any finite data, the weights are zero
            const int tin = q_in0 + (id - e * AROWS);
            const bool ok = (unsigned)tin < (unsigned)p.Tin;          // outside the sequence: 0 (= snake(0))
            const float val = xb[(size_t)ch * p.Tin + (ok ? tin : 0)];
            ev[st][it] = ok ? val : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            const int idx = tid + u * NT;                             // part = idx % 8, co = (idx / 8) % 128, plane = idx / 1024
            const int part = idx & 7, co = (idx >> 3) & (CO_T - 1), pl = idx >> 10;
            const short* src = (pl ? a.w_lo : a.w_hi) + (((size_t)c * a.cout_pad + co0 + co) * 8 + part) * 8;
            wv[st][u] = *reinterpret_cast<const bf16x8*>(src);
        }
    };
    auto write_lds = [&](auto buf_c, auto set_c) {
        constexpr int st = decltype(set_c)::value;
        auto& w_lds = sat_pick<decltype(buf_c)::value>(w_lds0, w_lds1);
        auto& a_lds = sat_pick<decltype(buf_c)::value>(a_lds0, a_lds1);
        auto& c_lds = sat_pick<decltype(buf_c)::value>(c_lds0, c_lds1);     // chunk c's constants live in slot c & 1 = buffer index
#pragma unroll
        for (int it = 0; it < NEL; ++it) {
            const int id = tid + it * NT;
            const int e = SAT_UNIFORM(id / AROWS);
            int row = id - e * AROWS;
            row = row < nrows ? row : DUMMY;
            float o = ev[st][it];
            if (SNAKE) o = sat_snake(o, c_lds[0][e], c_lds[1][e]);
            uint32_t h, l;
            sat_split2_pk(o, 0.0f, &h, &l);
            a_lds[0][row][e] = (short)h;
            a_lds[1][row][e] = (short)l;
        }
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            const int idx = tid + u * NT;
            const int part = idx & 7, co = (idx >> 3) & (CO_T - 1), pl = idx >> 10;
            *reinterpret_cast<bf16x8*>(&w_lds[pl][co][part * 8]) = wv[st][u];
        }
    };
    auto store_consts = [&](auto slot_c, float v) {
        auto& c_lds = sat_pick<decltype(slot_c)::value>(c_lds0, c_lds1);
        (&c_lds[0][0])[c_idx] = v;
    };
    auto mfma_phase = [&](auto buf_c) {
        auto& w_lds = sat_pick<decltype(buf_c)::value>(w_lds0, w_lds1);
        auto& a_lds = sat_pick<decltype(buf_c)::value>(a_lds0, a_lds1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int g = 2 * ks + hi;                     // k-slots 0-7 <- tap group 2ks (lanes 0-31), 8-15 <- group 2ks+1
            const int tap = g < K ? g : K - 1;             // groups >= K are zero-weight pads; keep the row in range
            bf16x8 wa[2][2], xa[2][2];                     // [mi|ni][plane]
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                wa[0][pl] = *reinterpret_cast<const bf16x8*>(&w_lds[pl][co_w + l31][g * 8]);
                wa[1][pl] = *reinterpret_cast<const bf16x8*>(&w_lds[pl][co_w + 32 + l31][g * 8]);
                xa[0][pl] = *reinterpret_cast<const bf16x8*>(&a_lds[pl][t_w + l31 + tap * dil][0]);
                xa[1][pl] = *reinterpret_cast<const bf16x8*>(&a_lds[pl][t_w + 32 + l31 + tap * dil][0]);
            }
            // rows of out-of-range channels have zero weights, so every wave runs all twelve MFMAs; consecutive MFMAs
            // go to different accumulators
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = sat_mfma_32x32x16_bf16(wa[mi][0], xa[ni][0], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = sat_mfma_32x32x16_bf16(wa[mi][0], xa[ni][1], acc[mi][ni]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = sat_mfma_32x32x16_bf16(wa[mi][1], xa[ni][0], acc[mi][ni]);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    // one pipeline phase: MFMAs on buffer BC (chunk c), conversion of register set / into buffer BN (chunk c+1); the two
    // halves of the workgroup take them in opposite order
    auto phase = [&](auto bc, auto bn) {
        if (mfma_first) {
            if (wave_on_co) mfma_phase(bc);
            write_lds(bn, bn);
        } else {
            write_lds(bn, bn);
            if (wave_on_co) mfma_phase(bc);
        }
    };

    // prologue: constants of chunks 0 and 1 in their slots, chunk 0 converted into buffer 0, chunk 1's data in flight in
    // register set 1, chunk 2's constants in flight in cv[0]
    if (SNAKE) {
        store_consts(I0{}, load_consts(0));
        store_consts(I1{}, load_consts(clampc(1)));
        __syncthreads();
        cv[0] = load_consts(clampc(2));
    }
    issue_loads(0, I0{});
    write_lds(I0{}, I0{});
    if (nchunks > 1) issue_loads(1, I1{});
    __syncthreads();
    int c = 0;
    // steady state, two chunks per trip so that buffer and register-set indices are compile-time constants
    for (; c + 2 < nchunks; c += 2) {
        issue_loads(c + 2, I0{});                          // register set 0 was consumed in the previous phase
        phase(I0{}, I1{});                                 // chunk c out of buffer 0; chunk c+1: set 1 -> buffer 1
        if (SNAKE) {
            store_consts(I0{}, cv[0]);                     // chunk c+2's constants (slot 0 was last read in phase c-1)
            cv[1] = load_consts(clampc(c + 3));
        }
        __syncthreads();
        issue_loads(clampc(c + 3), I1{});
        phase(I1{}, I0{});                                 // chunk c+1 out of buffer 1; chunk c+2: set 0 -> buffer 0
        if (SNAKE) {
            store_consts(I1{}, cv[1]);
            cv[0] = load_consts(clampc(c + 4));
        }
        __syncthreads();
    }
    // tail: chunk c is published in buffer 0; chunk c+1 (if any) is in register set 1, its constants in slot 1
    if (c + 1 < nchunks) {
        phase(I0{}, I1{});
        __syncthreads();
        if (wave_on_co) mfma_phase(I1{});
    } else {
        if (wave_on_co) mfma_phase(I0{});
    }

    // ------------------------------------ epilogue (as the generic kernel) ------------------------------------
    const bool bwd = (p.x2 != nullptr);
    const bool wave_on = (co0 + co_w) < a.cout_v;
    const bool mi1_on = (co0 + co_w + 32) < a.cout_v;
    if (bwd) {
        __syncthreads();
        for (int i = tid; i < 2 * TW * CO_T; i += NT) (&red_lds[0][0][0])[i] = 0.0f;
        __syncthreads();
    }
    const bool vec4 = (p.Tout & 3) == 0 && (((uintptr_t)p.y | (uintptr_t)p.x2 | (uintptr_t)p.res) & 15) == 0;
    if (vec4) {
        // 16-byte epilogue: each wave transposes its accumulators through LDS (the weight buffers are free now) so that
        // a lane owns 4 consecutive time steps of a row; the x2 / res loads of a 32-row half are all issued before use.
        if (!bwd) __syncthreads();                          // (bwd already synchronised above)
        float (*tile)[68] = reinterpret_cast<float (*)[68]>(wave < 4 ? &w_lds0[0][0][0] : &w_lds1[0][0][0]) + (wave & 3) * 32;
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
                    const bool ok = co < a.cout_v && tg < p.Tout;
                    const size_t o = ((size_t)b * p.Cout + (ok ? co : 0)) * p.Tout + (ok ? tg : 0);
                    xv[j] = (bwd && ok) ? *reinterpret_cast<const f32x4*>(p.x2 + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                    rv[j] = (p.res && ok) ? *reinterpret_cast<const f32x4*>(p.res + o) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int row = j * 4 + lr;
                    const int col = co_w + mi * 32 + row;
                    const int co = co0 + col;
                    const bool ok = co < a.cout_v && tg < p.Tout;
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
                const bool co_ok = co < a.cout_v;
                const float bias = ep_lds[0][col];
                const float a2 = ep_lds[1][col], b2 = ep_lds[2][col];
                float pda = 0.f, pdb = 0.f;
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int t = t0 + t_w + ni * 32 + l31;
                    if (co_ok && t < p.Tout) {
                        const size_t o = ((size_t)b * p.Cout + co) * p.Tout + t;
                        float v = acc[mi][ni][r] + bias;
                        if (bwd) {
                            const SatSnakeGrad g = sat_snake_grad(p.x2[o], a2, b2);
                            pda += v * g.dla;
                            pdb += v * g.dlb;
                            v *= g.dx;
                        }
                        if (p.res) v += p.res[o];
                        if (p.tanh_out) v = tanhf(v);
                        p.y[o] = v;
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
        if (tid < CO_T && m < a.cout_v) {
            float sa = 0.f, sb = 0.f;
#pragma unroll
            for (int w = 0; w < TW; ++w) {
                sa += red_lds[0][w][tid];
                sb += red_lds[1][w][tid];
            }
            const size_t row = (size_t)b * gridDim.y + t_tile;
            const size_t nrows_p = (size_t)p.B * gridDim.y;
            p.part_da[(size_t)m * nrows_p + row] = sa;
            p.part_db[(size_t)m * nrows_p + row] = sb;
        }
    }
}

static void sat_bf_launch_k7(SatConvBfLaunch& a, void* stream) {
    dim3 grid(a.cout_pad / SAT_K7_CO, sat_cdiv(a.nq, SAT_K7_T), a.p.B);
    const bool exact = (a.cin_v & 7) == 0;
    if (a.p.alpha) {
        if (exact) { SAT_LAUNCH((sat_conv1d_bf16x3_k7_kernel<true, true>), grid, dim3(SAT_K7_NT), stream, a); }
        else { SAT_LAUNCH((sat_conv1d_bf16x3_k7_kernel<true, false>), grid, dim3(SAT_K7_NT), stream, a); }
    } else {
        if (exact) { SAT_LAUNCH((sat_conv1d_bf16x3_k7_kernel<false, true>), grid, dim3(SAT_K7_NT), stream, a); }
        else { SAT_LAUNCH((sat_conv1d_bf16x3_k7_kernel<false, false>), grid, dim3(SAT_K7_NT), stream, a); }
    }
}
