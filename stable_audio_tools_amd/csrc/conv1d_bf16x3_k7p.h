// conv1d_bf16x3_k7p.h — the k = 5..8 stride-1 (dilated) convolutions of the ResidualUnits (autoencoders.py:58-83) and their
// data-gradients, fed from pre-split activation PLANES.  Included by conv1d_bf16x3.hip.
//
// conv1d_bf16x3_k7.h applies SnakeBeta and the bf16 hi/lo split while staging, per workgroup: at C = 128 that VALU + ds_write
// work (5 elements per thread and chunk, two 2-byte LDS stores each) costs as much as the chunk's MFMAs, and every one of the
// Cout / 128 channel tiles repeats it on the same input.  Here the activation is converted ONCE by sat_k7_planes_kernel into two
// bf16 planes laid out [B][Cin/8][rows][8 channels] (row = 32 + t, zero rows around the sequence: snake(0) = 0), and the conv
// kernel's K loop is matrix work only:
//   * per K-chunk (8 input channels x 8 tap groups) a stage holds the weight slab [2 planes][128 co][8 groups x 8] (32 KiB, 128-byte
//     rows, 16-byte chunks XOR-swizzled through the DMA source address) and the activation slab [2 planes][320 rows][8] (10 KiB);
//     both arrive by LDS-DMA (42 one-KiB pieces per chunk, 5-6 per wave), no staging registers, no VALU, no ds_write;
//   * two stages in a ring: chunk c+1 is requested right after the barrier that publishes chunk c; ONE barrier per chunk;
//   * PERSISTENT workgroups (one per CU) walk the tiles of the XCD-aware order: the next tile's first two chunks are requested before
//     the epilogue of the current one (which has its own LDS transposition space), its stores drain under the next K loop;
//   * fragment reads run one k-step ahead of the MFMAs (two register sets);
//   * the MFMA work (48 per wave and chunk: 2x2 tiles x 4 k-steps x 3 products) and the epilogue are those of conv1d_bf16x3_k7.h.
#pragma once

#define SAT_K7P_LEAD 32           // zero rows before t = 0 in a plane (>= pad)
static_assert(SAT_K7P_LEAD == SAT_K7P_LEAD_ROWS, "plane emission (conv1d_bf16x3.hip) writes row 32 + t");
#define SAT_K7P_WBYTES 32768      // weight slab of a stage
#define SAT_K7P_ABYTES 10240      // activation slab of a stage
#define SAT_K7P_STAGE (SAT_K7P_WBYTES + SAT_K7P_ABYTES)
#define SAT_K7P_NSTAGE 2

struct SatK7PlaneParams {
    const float* x;       // (B, Cin, Tin)
    const float* a;       // pre-exponentiated snake constants (Cin) or null
    const float* ib;
    short* hi;            // [B][c8][rows][8]
    short* lo;
    int B, Cin, Tin, rows, c8;
};

// one thread = one plane row (time step) of one 8-channel chunk: eight coalesced 4-byte loads, two 16-byte stores
__global__ void __launch_bounds__(256) sat_k7_planes_kernel(SatK7PlaneParams p) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    const int chunk = blockIdx.y, b = blockIdx.z;
    if (row >= p.rows) return;
    const int t = row - SAT_K7P_LEAD;
    const bool t_ok = (unsigned)t < (unsigned)p.Tin;
    uint32_t h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int ch = chunk * 8 + 2 * j + e;
            const int chc = ch < p.Cin ? ch : p.Cin - 1;
            const bool ok = t_ok && ch < p.Cin;
            float o = ok ? p.x[((size_t)b * p.Cin + chc) * p.Tin + t] : 0.0f;
            if (p.a) o = sat_snake(o, p.a[chc], p.ib[chc]);          // snake(0) = 0: the zero rows stay zero
            v[e] = o;
        }
        sat_split2_pk(v[0], v[1], &h[j], &l[j]);
    }
    const size_t o = (((size_t)b * p.c8 + chunk) * p.rows + row) * 8;
    *reinterpret_cast<u32x4*>(p.hi + o) = u32x4{h[0], h[1], h[2], h[3]};
    *reinterpret_cast<u32x4*>(p.lo + o) = u32x4{l[0], l[1], l[2], l[3]};
}

template <int DUMMY_UNUSED = 0>
__global__ void __launch_bounds__(SAT_K7_NT) sat_conv1d_bf16x3_k7p_kernel(SatConvBfLaunch a) {
    constexpr int CO_T = SAT_K7_CO, T_T = SAT_K7_T, NT = SAT_K7_NT;
    constexpr int TW = T_T / 64;                          // waves along time
    const SatConvParams& p = a.p;
    __shared__ __attribute__((aligned(1024))) char ring[SAT_K7P_NSTAGE * SAT_K7P_STAGE];
    __shared__ __attribute__((aligned(16))) float epi[SAT_K7_NT / 64][32][68];      // the epilogue's per-wave transposition tiles
    __shared__ float red_lds[2][TW][CO_T];
    __shared__ float ep_lds[3][CO_T];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int co_w = (wave / TW) * 64, t_w = (wave % TW) * 64;
    const int K = p.K, dil = p.dil;
    const int nchunks = (a.cin_v + 7) / 8;
    // PERSISTENT workgroups: workgroup w takes the tiles w, w + gridDim.x, ... of the XCD-aware order (gridDim.x is a multiple of 8 or the
    // whole tile count, so a workgroup stays on the channel tiles of "its" XCD).  The next tile's first two chunks are requested BEFORE
    // this tile's epilogue, the epilogue's stores drain under the next K loop, and there is no workgroup turnover between tiles.
    const int co_tiles = a.cout_pad / CO_T, t_tiles = (a.nq + T_T - 1) / T_T;
    const int total = co_tiles * t_tiles * p.B;
    struct Tile { int co0, b, t_tile, t0, row_in0; };
    auto tile_of = [&](int vt) {
        int co_tile, win;
        sat_xcd_tile(vt, co_tiles, t_tiles * p.B, &co_tile, &win);
        Tile t;
        t.b = win / t_tiles;
        t.t_tile = win - t.b * t_tiles;
        t.co0 = co_tile * CO_T;
        t.t0 = t.t_tile * T_T;
        t.row_in0 = SAT_K7P_LEAD + t.t0 - p.pad;           // plane row of the window's first input step (>= 0: pad <= LEAD)
        return t;
    };
    // ---- LDS-DMA of one chunk: weights 32 pieces (4 per wave), activations 10 pieces (1 per wave, waves 0-1 a second one) ----
    const int srow = lane >> 3, sslot = lane & 7;
    auto issue = [&](const Tile& tl, int c, int stage) {
        char* base = ring + stage * SAT_K7P_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = wave * 4 + i, pl = q >> 4, r = (q & 15) * 8 + srow;
            const int g = sslot ^ ((r >> 1) & 7);
            const short* src = (pl ? a.w_lo : a.w_hi) + (((size_t)c * a.cout_pad + tl.co0 + r) * 8 + g) * 8;
            sat_glds16(src, base + q * 1024);
        }
        {
            const int pl = wave / 5, sub = wave % 5;          // pieces 0..7
            const short* src = (pl ? a.xp_lo : a.xp_hi) + (((size_t)tl.b * a.xp_c8 + c) * a.xp_rows + tl.row_in0 + sub * 64 + lane) * 8;
            sat_glds16(src, base + SAT_K7P_WBYTES + wave * 1024);
        }
        if (wave < 2) {                                        // pieces 8, 9 (plane 1, sub 3 and 4)
            const short* src = a.xp_lo + (((size_t)tl.b * a.xp_c8 + c) * a.xp_rows + tl.row_in0 + (3 + wave) * 64 + lane) * 8;
            sat_glds16(src, base + SAT_K7P_WBYTES + (8 + wave) * 1024);
        }
    };
    f32x16 acc[2][2];
    // Fragment reads run one k-step AHEAD of the MFMAs that consume them (two register sets): the LDS latency of a k-step's eight
    // 16-byte reads is paid once per chunk instead of once per k-step — both waves of a SIMD run this same code in near lock-step, so
    // a stall of one is not covered by the other.
    struct Frags { bf16x8 wa[2][2], xa[2][2]; };       // [mi|ni][plane]
    auto load_frags = [&](Frags& f, const char* wb, const char* ab, int ks) {
        const int g = 2 * ks + hi;                     // k-slots 0-7 <- tap group 2ks (lanes 0-31), 8-15 <- group 2ks+1
        const int tap = g < K ? g : K - 1;             // groups >= K are zero-weight pads; keep the row in range
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int r = co_w + mi * 32 + l31;
                f.wa[mi][pl] = *reinterpret_cast<const bf16x8*>(wb + pl * 16384 + r * 128 + ((g ^ ((r >> 1) & 7)) << 4));
            }
            f.xa[0][pl] = *reinterpret_cast<const bf16x8*>(ab + pl * 5120 + (t_w + l31 + tap * dil) * 16);
            f.xa[1][pl] = *reinterpret_cast<const bf16x8*>(ab + pl * 5120 + (t_w + 32 + l31 + tap * dil) * 16);
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
    auto mfma_phase = [&](int stage) {
        const char* wb = ring + stage * SAT_K7P_STAGE;
        const char* ab = wb + SAT_K7P_WBYTES;
        Frags f0, f1;
        load_frags(f0, wb, ab, 0);
        load_frags(f1, wb, ab, 1);
        SAT_SCHED_FENCE();
        mfma_frags(f0);
        SAT_SCHED_FENCE();
        load_frags(f0, wb, ab, 2);
        SAT_SCHED_FENCE();
        mfma_frags(f1);
        SAT_SCHED_FENCE();
        load_frags(f1, wb, ab, 3);
        SAT_SCHED_FENCE();
        mfma_frags(f0);
        SAT_SCHED_FENCE();
        mfma_frags(f1);
    };


    int vt = blockIdx.x;
    if (vt >= total) return;                               // (never: the grid is capped at the tile count)
    Tile cur = tile_of(vt);
    issue(cur, 0, 0);
    if (nchunks > 1) issue(cur, 1, 1);
    int ep_co0 = -1;
    for (; vt < total; vt += gridDim.x) {
        const int co0 = cur.co0, b = cur.b, t_tile = cur.t_tile, t0 = cur.t0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
        if (co0 != ep_co0) {                               // per-channel epilogue constants (the previous tile's epilogue ended with a barrier)
            if (tid < CO_T) {
                const int m = co0 + tid;
                const bool ok = m < a.cout_v;
                ep_lds[0][tid] = (ok && p.bias) ? p.bias[m] : 0.0f;
                ep_lds[1][tid] = (ok && p.x2) ? expf(p.alpha2[m]) : 1.0f;
                ep_lds[2][tid] = (ok && p.x2) ? expf(p.beta2[m]) : 1.0f;
            }
            ep_co0 = co0;
        }
        // K loop over a two-stage ring: chunk c+1 is requested right after the barrier that publishes chunk c (its stage was read by
        // chunk c-1); chunks 0 and 1 of a tile were requested before the previous tile's epilogue
        for (int c = 0; c < nchunks; ++c) {
            SAT_WAIT_VMCNT(0);                             // this wave's pieces of chunk c (and, at c == 0, the previous epilogue's stores)
            SAT_RAW_BARRIER();                             // everyone's pieces; every wave is done with chunk c-1
            if (c >= 1 && c + 1 < nchunks) issue(cur, c + 1, (c + 1) & 1);
            mfma_phase(c & 1);
        }
        __syncthreads();                                   // every wave is done with the ring
        Tile nxt = cur;
        if (vt + (int)gridDim.x < total) {
            nxt = tile_of(vt + gridDim.x);
            issue(nxt, 0, 0);
            if (nchunks > 1) issue(nxt, 1, 1);
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
            // 16-byte epilogue: each wave transposes its accumulators through LDS (its own LDS space: the ring already receives the next tile) so that
            // a lane owns 4 consecutive time steps of a row; the x2 / res loads of a 32-row half are all issued before use.
            if (!bwd) __syncthreads();                          // (bwd already synchronised above)
            float (*tile)[68] = epi[wave];                         // 32 x 68 floats per wave, apart from the stage ring
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
                const size_t row = (size_t)b * t_tiles + t_tile;
                const size_t nrows_p = (size_t)p.B * t_tiles;
                p.part_da[(size_t)m * nrows_p + row] = sa;
                p.part_db[(size_t)m * nrows_p + row] = sb;
            }
        }
        __syncthreads();                                   // epi / ep_lds / red_lds are free for the next tile
        cur = nxt;
    }
}

static void sat_bf_launch_k7p(SatConvBfLaunch& a, void* stream) {
    const long long total = (long long)(a.cout_pad / SAT_K7_CO) * sat_cdiv(a.nq, SAT_K7_T) * a.p.B;
    int ncu = 256;                                         // MI355X: one persistent workgroup per CU (158 KB of LDS each)
    if (const char* e = getenv("SAT_K7P_MAX_WGS")) {       // tests: force several tiles per workgroup on small shapes
        const int v = atoi(e);
        if (v > 0) ncu = v;
    }
    dim3 grid((unsigned)(total < ncu ? total : ncu));
    SAT_LAUNCH((sat_conv1d_bf16x3_k7p_kernel<0>), grid, dim3(SAT_K7_NT), stream, a);
}
