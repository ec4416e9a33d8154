// edge_conv.hip — the two-channel ends of the Oobleck stack (round 6): the encoder's first conv (stereo audio -> 128 channels,
// models/autoencoders.py:303 WNConv1d(in_channels, c_mults[0] * channels, 7, padding=3)), the decoder's last conv (Snake(128) ->
// WNConv1d(128, out_channels, 7, padding=3, bias=False), :355-356), the data-gradient of the latter and both weight gradients.
//
// On the matrix kernels a 2-channel operand is padded to a 16-wide MFMA k-chunk or a 128-row output tile: 0.012 of the matrix peak
// forward (profiles/r05_vae_train_kernel_stats.csv: sat_conv1d_bf16x3_k7_kernel 0.53 / 1.22 ms per launch) and a full 128-row
// weight-gradient tile for two rows.  The work is 14 multiply-adds per element of the 128-channel tensor — far under the vector
// rate needed to keep up with HBM (1 GiB at T = 2 097 152) — so these kernels are plain fp32 FMA streams over (B, C, T), exact fp32
// arithmetic, bound by the one pass over the wide tensor:
//   sat_edge_conv_in_kernel   Cin <= 2 -> Cout: y[co][t] = b[co] + sum_{ci,k} W[co][ci][k] x[ci][t + k - pad]; optional data-gradient
//                             epilogue (x dsnake(x2), per-channel d log-alpha / d log-beta partial sums) — the decoder's last conv
//                             backward is this kernel on the transposed, tap-flipped weight (mode 1)
//   sat_edge_conv_out_kernel  Cin -> Cout <= 2 with the SnakeBeta prologue: activations staged ONCE per 8 channels through LDS (each
//                             sin^2 evaluated once, not once per tap), 7 x Cout FMAs per staged value
//   sat_edge_wgrad_kernel     dW[m][n][k] = sum_t dy[m][t] act(x)[n][t + k - pad] when one of (M, N) is <= 2: the wide operand is
//                             streamed row by row, the narrow one held as a 12-value window; per-wave accumulators over a range of
//                             time tiles, one slab per range (WgradSlabs, torch order) + the wide operand's row sums (bias gradient)
// Time tile: 1024 steps per 256-thread workgroup, four consecutive steps per lane (16-byte accesses, 1 KiB per wave instruction).
#include "conv_common.h"

#define SAT_EC_TT 1024
#define SAT_EC_ROWS 32          // wide-tensor rows per workgroup (8 per wave in the wgrad kernel)

struct SatEdgeParams {
    const float* x;        // conv input (B, Cin, T)            | wgrad: the WIDE operand (B, R, T)
    const float* w;        // torch weight                      | wgrad: the NARROW operand (B, S, T)
    const float* bias;     // (Cout) or null
    const float* alpha;    // SnakeBeta log-alpha / log-beta of the conv INPUT (out kernel; wgrad: of the wide operand), or null
    const float* beta;
    float* y;              // (B, Cout, T)                      | wgrad: slabs [nsplit][M * N * K]
    const float* x2;       // data-gradient epilogue: y *= dsnake(x2); (B, Cout, T) or null
    const float* alpha2;
    const float* beta2;
    float* part_da;        // [Cout][B * tiles]                 | wgrad: row sums of the wide operand [R][nsplit] or null
    float* part_db;
    int B, Cin, Cout, T, K, pad;
    int mode;              // conv: 0 = W[co][ci][k] is w[co][ci][k]; 1 = W[co][ci][k] = w[ci][co][K-1-k] (data-gradient of w's conv)
    int tanh_out;
    int tiles_per_split, nsplit_t;      // wgrad
    int flip;              // wgrad: 0: dW[row][s][k] (wide = dy); 1: dW[s][row][K-1-k] (wide = x, narrow = dy)
    // plane emission (narrow-input conv): act_next(y) as the bf16 hi / lo planes [B][em_c8][em_rows][8] of the k7 conv that reads y next
    // (conv1d_planes.h layout: row SAT_EC_LEAD + t); em_alpha / em_beta: that conv's SnakeBeta log parameters, or null (identity)
    short* em_hi;
    short* em_lo;
    const float* em_alpha;
    const float* em_beta;
    int em_rows, em_c8;
};
#define SAT_EC_LEAD 32      // == SAT_K7P_LEAD (conv1d_planes.h): zero rows in front of t = 0 in a plane

// the 12-value window t0 - 4 .. t0 + 7 of one row (zero outside [0, T)); vec: 16-byte loads allowed (T % 4 == 0, aligned base)
SAT_DEVICE void sat_ec_window(const float* row, int t0, int T, bool vec, float (&win)[12]) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int t = t0 - 4 + 4 * q;
        if (vec && t >= 0 && t + 3 < T) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + t);
#pragma unroll
            for (int e = 0; e < 4; ++e) win[4 * q + e] = v[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) win[4 * q + e] = (t + e >= 0 && t + e < T) ? row[t + e] : 0.0f;
        }
    }
}
SAT_DEVICE f32x4 sat_ec_load4(const float* row, int t0, int T, bool vec) {
    if (vec && t0 + 3 < T) return *reinterpret_cast<const f32x4*>(row + t0);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (t0 + e < T) ? row[t0 + e] : 0.0f;
    return v;
}
SAT_DEVICE void sat_ec_store4(float* row, int t0, int T, bool vec, f32x4 v) {
    if (vec && t0 + 3 < T) {
        *reinterpret_cast<f32x4*>(row + t0) = v;
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (t0 + e < T) row[t0 + e] = v[e];
}

// ---------------------------------------------------------------------------------------------
// narrow input: Cin = S <= 2.  Everything that depends on the output channel only — the 7 x S taps (zero outside the conv's K), the
// bias, the data-gradient's and the emission's SnakeBeta constants — is staged in LDS once per workgroup and read as broadcasts:
// the first version read them per row with scalar / uniform global loads behind their own waits (14 dependent L2 latencies per row:
// 0.8 ms forward, 1.56 ms data-gradient at T = 2 097 152 — profiles/r06_experiments/edge_convs/).  Rows are processed in groups of 8
// in one basic block (no per-row early exits: the x2 loads of a group are all in flight before the first is used).
// ---------------------------------------------------------------------------------------------
template <int S>
__global__ void __launch_bounds__(256) sat_edge_conv_in_kernel(SatEdgeParams p) {
    __shared__ __attribute__((aligned(16))) float wl[SAT_EC_ROWS][16];      // [row][ci * 8 + (d + 3)], d = tap offset -3 .. 3
    __shared__ float cl[5][SAT_EC_ROWS];                                    // bias, a2, b2, em_a, em_ib
    __shared__ float red[2][SAT_EC_ROWS][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.z, co0 = blockIdx.y * SAT_EC_ROWS;
    const int t0 = blockIdx.x * SAT_EC_TT + threadIdx.x * 4;
    const bool vec = (p.T & 3) == 0 && ((((uintptr_t)p.x) | ((uintptr_t)p.y) | ((uintptr_t)p.x2)) & 15) == 0;
    const int K = p.K, pad = p.pad;
    const bool bwd = p.x2 != nullptr;
    for (int i = threadIdx.x; i < SAT_EC_ROWS * 16; i += 256) {
        const int j = i >> 4, ci = (i >> 3) & 1, d = (i & 7) - 3, k = d + pad;
        const int co = co0 + j;
        float v = 0.0f;
        if (co < p.Cout && ci < S && d <= 3 && k >= 0 && k < K)
            v = p.mode == 0 ? p.w[((size_t)co * S + ci) * K + k] : p.w[((size_t)ci * p.Cout + co) * K + (K - 1 - k)];
        wl[j][i & 15] = v;
    }
    if (threadIdx.x < SAT_EC_ROWS) {
        const int co = co0 + threadIdx.x;
        const bool ok = co < p.Cout;
        cl[0][threadIdx.x] = (ok && p.bias) ? p.bias[co] : 0.0f;
        cl[1][threadIdx.x] = (ok && bwd) ? expf(p.alpha2[co]) : 1.0f;
        cl[2][threadIdx.x] = (ok && bwd) ? expf(p.beta2[co]) : 1.0f;
        cl[3][threadIdx.x] = (ok && p.em_alpha) ? expf(p.em_alpha[co]) : 0.0f;
        cl[4][threadIdx.x] = (ok && p.em_alpha) ? 1.0f / (expf(p.em_beta[co]) + 1e-9f) : 0.0f;
    }
    float win[S][12];
#pragma unroll
    for (int ci = 0; ci < S; ++ci) sat_ec_window(p.x + ((size_t)b * S + ci) * p.T, t0, p.T, vec, win[ci]);
    __syncthreads();
    const int last = p.Cout - 1;
    for (int jg = 0; jg < SAT_EC_ROWS / 8; ++jg) {
        if (co0 + jg * 8 >= p.Cout) break;            // block-uniform
        f32x4 xv[8];
        if (bwd) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int co = co0 + jg * 8 + jj;
                xv[jj] = sat_ec_load4(p.x2 + ((size_t)b * p.Cout + (co < p.Cout ? co : last)) * p.T, t0, p.T, vec);
            }
        }
        f32x4 out[8];
        float pda[8], pdb[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = jg * 8 + jj;
            float wr[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&wl[j][4 * q]);      // broadcast reads
#pragma unroll
                for (int e = 0; e < 4; ++e) wr[4 * q + e] = v[e];
            }
            f32x4 acc;
            const float bv = cl[0][j];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = bv;
#pragma unroll
            for (int ci = 0; ci < S; ++ci)
#pragma unroll
                for (int d = -3; d <= 3; ++d)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(wr[ci * 8 + d + 3], win[ci][4 + e + d], acc[e]);
            pda[jj] = 0.0f;
            pdb[jj] = 0.0f;
            if (bwd) {
                const float a2 = cl[1][j], b2 = cl[2][j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const SatSnakeGrad g = sat_snake_grad(xv[jj][e], a2, b2);
                    const bool ok = t0 + e < p.T;
                    pda[jj] += ok ? acc[e] * g.dla : 0.0f;
                    pdb[jj] += ok ? acc[e] * g.dlb : 0.0f;
                    acc[e] *= g.dx;
                }
            }
            if (p.tanh_out) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = tanhf(acc[e]);
            }
            out[jj] = acc;
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int co = co0 + jg * 8 + jj;
            if (co < p.Cout) sat_ec_store4(p.y + ((size_t)b * p.Cout + co) * p.T, t0, p.T, vec, out[jj]);
        }
        if (bwd) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {           // 16 independent reductions: their cross-lane steps overlap
                pda[jj] = sat_wave_sum(pda[jj]);
                pdb[jj] = sat_wave_sum(pdb[jj]);
            }
            if (lane == 0) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    red[0][jg * 8 + jj][wave] = pda[jj];
                    red[1][jg * 8 + jj][wave] = pdb[jj];
                }
            }
        }
        if (p.em_hi) {
            // act_next(y): 8 consecutive channels of one time step = one 16-byte plane row; a lane's four steps are 64 contiguous bytes per
            // plane (channels past Cout: weights and bias were staged as 0, act(0) = 0)
            if (p.em_alpha) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const float ea = cl[3][jg * 8 + jj], eib = cl[4][jg * 8 + jj];
#pragma unroll
                    for (int e = 0; e < 4; ++e) out[jj][e] = sat_snake(out[jj][e], ea, eib);
                }
            }
            const int c8i = (co0 >> 3) + jg;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (t0 + e < p.T) {
                    uint32_t eh[4], el[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) sat_split2_pk(out[2 * q][e], out[2 * q + 1][e], &eh[q], &el[q]);
                    const size_t o = (((size_t)b * p.em_c8 + c8i) * p.em_rows + SAT_EC_LEAD + t0 + e) * 8;
                    *reinterpret_cast<u32x4*>(p.em_hi + o) = u32x4{eh[0], eh[1], eh[2], eh[3]};
                    *reinterpret_cast<u32x4*>(p.em_lo + o) = u32x4{el[0], el[1], el[2], el[3]};
                }
            }
        }
    }
    if (bwd) {
        __syncthreads();
        const int j = threadIdx.x;
        if (j < SAT_EC_ROWS && co0 + j < p.Cout) {
            const size_t nrows = (size_t)p.B * gridDim.x, row = (size_t)b * gridDim.x + blockIdx.x;
            p.part_da[(size_t)(co0 + j) * nrows + row] = (red[0][j][0] + red[0][j][1]) + (red[0][j][2] + red[0][j][3]);
            p.part_db[(size_t)(co0 + j) * nrows + row] = (red[1][j][0] + red[1][j][1]) + (red[1][j][2] + red[1][j][3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// narrow output: Cout = S <= 2, SnakeBeta prologue on the Cin-channel input.  Batches of 8 input channels: their activated steps
// (+ halo) and their 8 x S x 7 taps go through LDS; the NEXT batch's global loads are issued before the current batch's arithmetic.
// ---------------------------------------------------------------------------------------------
#define SAT_EC_CB 8                      // channels staged per barrier pair
#define SAT_EC_LROW (SAT_EC_TT + 8)      // t0_tile - 4 .. t0_tile + 1028
template <int S>
__global__ void __launch_bounds__(256) sat_edge_conv_out_kernel(SatEdgeParams p) {
    __shared__ __attribute__((aligned(16))) float a_lds[SAT_EC_CB][SAT_EC_LROW];
    __shared__ __attribute__((aligned(16))) float wl[SAT_EC_CB][16];         // [channel of the batch][co * 8 + (d + 3)]
    const int b = blockIdx.z;
    const int tile0 = blockIdx.x * SAT_EC_TT;
    const int t0 = tile0 + threadIdx.x * 4;
    const bool vec = (p.T & 3) == 0 && ((((uintptr_t)p.x) | ((uintptr_t)p.y)) & 15) == 0;
    const int K = p.K, pad = p.pad;
    f32x4 acc[S];
#pragma unroll
    for (int co = 0; co < S; ++co) {
        const float bv = p.bias ? p.bias[co] : 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[co][e] = bv;
    }
    // per-thread roles that do not change over the batches
    const int hc = threadIdx.x >> 3, hh = threadIdx.x & 7;                   // halo element (threads 0-63): channel of the batch, slot
    const int hidx = hh < 4 ? hh : SAT_EC_TT + hh;                           // LDS index 0..3 | 1028..1031
    const int ht = tile0 - 4 + hidx;
    const int wc = threadIdx.x >> 4, wi = threadIdx.x & 15;                  // weight element (threads 0-127): channel of the batch, slot
    const int wco = wi >> 3, wd = (wi & 7) - 3, wk = wd + pad;
    f32x4 mine[SAT_EC_CB];
    float halo = 0.0f, wv = 0.0f, sa = 1.0f, sib = 0.0f;                     // sa / sib: this thread's halo channel's SnakeBeta constants
    auto fetch = [&](int c0) {
#pragma unroll
        for (int c = 0; c < SAT_EC_CB; ++c) {
            const int ci = c0 + c < p.Cin ? c0 + c : p.Cin - 1;
            mine[c] = sat_ec_load4(p.x + ((size_t)b * p.Cin + ci) * p.T, t0, p.T, vec);
        }
        halo = 0.0f;
        wv = 0.0f;
        if (threadIdx.x < 8 * SAT_EC_CB) {
            const int ci = c0 + hc;
            if (ci < p.Cin && ht >= 0 && ht < p.T) halo = p.x[((size_t)b * p.Cin + ci) * p.T + ht];
            if (p.alpha && ci < p.Cin) {
                sa = expf(p.alpha[ci]);
                sib = 1.0f / (expf(p.beta[ci]) + 1e-9f);
            }
        }
        if (threadIdx.x < 16 * SAT_EC_CB) {
            const int ci = c0 + wc;
            if (ci < p.Cin && wco < S && wd <= 3 && wk >= 0 && wk < K)
                wv = p.mode == 0 ? p.w[((size_t)wco * p.Cin + ci) * K + wk] : p.w[((size_t)ci * S + wco) * K + (K - 1 - wk)];
        }
    };
    fetch(0);
    for (int c0 = 0; c0 < p.Cin; c0 += SAT_EC_CB) {
        __syncthreads();                                                     // the previous batch's readers are done
#pragma unroll
        for (int c = 0; c < SAT_EC_CB; ++c) {
            f32x4 v = mine[c];
            if (p.alpha) {
                const int ci = c0 + c < p.Cin ? c0 + c : p.Cin - 1;
                const float a = expf(p.alpha[ci]), ib = 1.0f / (expf(p.beta[ci]) + 1e-9f);      // (block-uniform: scalar loads, batched)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = sat_snake(v[e], a, ib);
            }
            *reinterpret_cast<f32x4*>(&a_lds[c][4 + threadIdx.x * 4]) = v;   // (steps past T were loaded as 0: act(0) = 0)
        }
        if (threadIdx.x < 8 * SAT_EC_CB) a_lds[hc][hidx] = p.alpha ? sat_snake(halo, sa, sib) : halo;
        if (threadIdx.x < 16 * SAT_EC_CB) wl[wc][wi] = wv;                   // (channels past Cin: zero taps)
        if (c0 + SAT_EC_CB < p.Cin) fetch(c0 + SAT_EC_CB);                   // next batch: in flight under this batch's arithmetic
        __syncthreads();
#pragma unroll
        for (int c = 0; c < SAT_EC_CB; ++c) {
            float win[12], wr[16];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&a_lds[c][threadIdx.x * 4 + 4 * q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) win[4 * q + e] = v[e];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&wl[c][4 * q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) wr[4 * q + e] = v[e];
            }
#pragma unroll
            for (int co = 0; co < S; ++co)
#pragma unroll
                for (int d = -3; d <= 3; ++d)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[co][e] = fmaf(wr[co * 8 + d + 3], win[4 + e + d], acc[co][e]);
        }
    }
#pragma unroll
    for (int co = 0; co < S; ++co) {
        if (p.tanh_out) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[co][e] = tanhf(acc[co][e]);
        }
        sat_ec_store4(p.y + ((size_t)b * S + co) * p.T, t0, p.T, vec, acc[co]);
    }
}

// ---------------------------------------------------------------------------------------------
// weight gradient with one narrow side.  wide (B, R, T) is streamed (optionally through SnakeBeta), narrow (B, S, T) windowed:
//   acc[row][s][d + 3] += sum_t wide[row][t] * narrow[s][t + d],   d = -3 .. 3   (tap k = d + pad)
// wave w of workgroup (split, row group, b) owns rows rg * 32 + w * 8 .. + 7 over the split's time tiles; slab (b * nsplit_t + split).
// The 8 row loads of a 256-step chunk are issued together (one basic block: rows past R re-read the last row and are not stored).
// ---------------------------------------------------------------------------------------------
template <int S>
__global__ void __launch_bounds__(256) sat_edge_wgrad_kernel(SatEdgeParams p) {
    const int lane = threadIdx.x & 63, wave = SAT_UNIFORM((int)(threadIdx.x >> 6));
    const int b = blockIdx.z, split = blockIdx.x;
    const int R = p.Cin;                                   // rows of the wide operand
    const int row0 = blockIdx.y * SAT_EC_ROWS + wave * 8;
    const bool vec = (p.T & 3) == 0 && ((((uintptr_t)p.x) | ((uintptr_t)p.w)) & 15) == 0;
    const int K = p.K, pad = p.pad;
    float acc[8][S][7];
    float rs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        rs[j] = 0.0f;
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int k = 0; k < 7; ++k) acc[j][s][k] = 0.0f;
    }
    float sa[8], sib[8];
    const float* rowp[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = row0 + j < R ? row0 + j : R - 1;
        rowp[j] = p.x + ((size_t)b * R + r) * p.T;
        sa[j] = p.alpha ? expf(p.alpha[r]) : 1.0f;
        sib[j] = p.alpha ? 1.0f / (expf(p.beta[r]) + 1e-9f) : 0.0f;
    }
    const int ntiles = (p.T + SAT_EC_TT - 1) / SAT_EC_TT;
    const int tile_a = split * p.tiles_per_split;
    const int tile_b = tile_a + p.tiles_per_split < ntiles ? tile_a + p.tiles_per_split : ntiles;
    const int t_end = tile_b * SAT_EC_TT < p.T ? tile_b * SAT_EC_TT : p.T;
    for (int tc = tile_a * SAT_EC_TT; tc < t_end; tc += 256) {      // 256-step chunks: a wave instruction covers 1 KiB of a row
        const int t0 = tc + lane * 4;
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = sat_ec_load4(rowp[j], t0, p.T, vec);       // (steps past T load as 0)
        float win[S][12];
#pragma unroll
        for (int s = 0; s < S; ++s) sat_ec_window(p.w + ((size_t)b * S + s) * p.T, t0, p.T, vec, win[s]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (p.alpha) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[j][e] = sat_snake(v[j][e], sa[j], sib[j]);
            }
            rs[j] += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int d = -3; d <= 3; ++d)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][s][d + 3] = fmaf(v[j][e], win[s][4 + e + d], acc[j][s][d + 3]);
        }
    }
    const int M = p.flip ? S : R, N = p.flip ? R : S;      // dW is (M, N, K)
    float* slab = p.y + ((size_t)b * p.nsplit_t + split) * ((size_t)M * N * K);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int r = row0 + j;
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int d = -3; d <= 3; ++d) acc[j][s][d + 3] = sat_wave_sum(acc[j][s][d + 3]);
        rs[j] = sat_wave_sum(rs[j]);
        if (lane == 0 && r < R) {
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int d = -3; d <= 3; ++d) {
                    const int k = d + pad;
                    if (k >= 0 && k < K) {
                        if (p.flip) slab[((size_t)s * N + r) * K + (K - 1 - k)] = acc[j][s][d + 3];
                        else slab[((size_t)r * N + s) * K + k] = acc[j][s][d + 3];
                    }
                }
            if (p.part_da) p.part_da[(size_t)r * ((size_t)p.B * p.nsplit_t) + (size_t)b * p.nsplit_t + split] = rs[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// 1: the shape is served — stride-1 'same' conv with odd K <= 7 (2 * pad == K - 1: taps t - 3 .. t + 3 of the 12-value window),
// one side <= 2 channels and the other >= 8
extern "C" int sat_edge_conv_ok(int cin, int cout, int k, int stride, int dil, int pad) {
    if (stride != 1 || dil != 1 || k < 1 || k > 7 || 2 * pad != k - 1) return 0;
    if (cin >= 1 && cin <= 2 && cout >= 8) return 1;
    if (cout >= 1 && cout <= 2 && cin >= 8) return 1;
    return 0;
}
// rows of the data-gradient epilogue's partial sums (part_da / part_db are [Cout][rows])
extern "C" int sat_edge_conv_partial_rows(int B, int T) { return B * sat_cdiv(T, SAT_EC_TT); }

// y (B, Cout, T) = conv1d(act(x), W) + bias, stride 1, T_out == T (2 * pad == K - 1).  w: the torch weight — (Cout, Cin, K) with
// mode 0, or (Cin, Cout, K) with mode 1 (the data-gradient of that weight's conv: transposed, taps flipped; pass pad' = K - 1 - pad).
// alpha / beta (log, per INPUT channel; Cout <= 2 kernel only) | NULL.  x2 / alpha2 / beta2 / part_da / part_db: data-gradient epilogue
// (Cin <= 2 kernel only).  em_hi / em_lo (Cin <= 2 kernel only) | NULL: also write act_next(y) — SnakeBeta with log parameters em_alpha /
// em_beta, or the identity — as the activation planes [B][ceil(Cout / 8)][em_rows][8] (row 32 + t) of the k7 conv that reads y next
// (sat_conv1d_bf16x3_planesq); rows around the sequence stay as the caller zeroed them.  No residual input.
extern "C" int sat_edge_conv(const float* x, const float* w, const float* bias, const float* alpha, const float* beta, float* y,
                             const float* x2, const float* alpha2, const float* beta2, float* part_da, float* part_db, short* em_hi,
                             short* em_lo, const float* em_alpha, const float* em_beta, int em_rows, int B, int Cin, int Cout, int T,
                             int K, int pad, int mode, int tanh_out, void* stream) {
    if (B <= 0 || T <= 0 || !x || !w || !y) { sat_set_error("sat_edge_conv: empty shape / missing buffer"); return 1; }
    if (!sat_edge_conv_ok(Cin, Cout, K, 1, 1, pad) || 2 * pad != K - 1) { sat_set_error("sat_edge_conv: shape not served (sat_edge_conv_ok; 2 * pad == K - 1)"); return 1; }
    if (mode != 0 && mode != 1) { sat_set_error("sat_edge_conv: mode must be 0 or 1"); return 1; }
    if ((alpha == nullptr) != (beta == nullptr)) { sat_set_error("sat_edge_conv: alpha and beta come together"); return 1; }
    SatEdgeParams p{};
    p.x = x; p.w = w; p.bias = bias; p.alpha = alpha; p.beta = beta; p.y = y; p.x2 = x2; p.alpha2 = alpha2; p.beta2 = beta2;
    p.part_da = part_da; p.part_db = part_db; p.B = B; p.Cin = Cin; p.Cout = Cout; p.T = T; p.K = K; p.pad = pad; p.mode = mode;
    p.tanh_out = tanh_out;
    if (em_hi || em_lo) {
        if (Cin > 2 || !em_hi || !em_lo || em_rows < SAT_EC_LEAD + T || (((uintptr_t)em_hi | (uintptr_t)em_lo) & 15) || (em_alpha == nullptr) != (em_beta == nullptr)) {
            sat_set_error("sat_edge_conv: plane emission needs the narrow-input form, both planes 16-byte aligned, em_rows >= 32 + T");
            return 1;
        }
        p.em_hi = em_hi; p.em_lo = em_lo; p.em_alpha = em_alpha; p.em_beta = em_beta; p.em_rows = em_rows; p.em_c8 = sat_cdiv(Cout, 8);
    }
    const int tiles = sat_cdiv(T, SAT_EC_TT);
    if (Cin <= 2) {
        if (alpha) { sat_set_error("sat_edge_conv: no SnakeBeta prologue on the narrow-input kernel"); return 1; }
        if (x2 && (!alpha2 || !beta2 || !part_da || !part_db)) { sat_set_error("sat_edge_conv: data-gradient epilogue incomplete"); return 1; }
        dim3 grid(tiles, sat_cdiv(Cout, SAT_EC_ROWS), B);
        if (Cin == 1) SAT_LAUNCH(sat_edge_conv_in_kernel<1>, grid, dim3(256), stream, p);
        else SAT_LAUNCH(sat_edge_conv_in_kernel<2>, grid, dim3(256), stream, p);
    } else {
        if (x2) { sat_set_error("sat_edge_conv: no data-gradient epilogue on the narrow-output kernel"); return 1; }
        dim3 grid(tiles, 1, B);
        if (Cout == 1) SAT_LAUNCH(sat_edge_conv_out_kernel<1>, grid, dim3(256), stream, p);
        else SAT_LAUNCH(sat_edge_conv_out_kernel<2>, grid, dim3(256), stream, p);
    }
    return sat_check_launch("sat_edge_conv");
}

// slabs per batch item of sat_edge_conv_wgrad (the call writes B * this many slabs of M * N * K floats)
static int sat_edge_wgrad_plan(int B, int rows, int T, int* per) {
    const int tiles = sat_cdiv(T, SAT_EC_TT);
    int want = sat_cdiv(4 * sat_cu_count(), B * sat_cdiv(rows, SAT_EC_ROWS));      // ~4 workgroups per CU
    if (want < 1) want = 1;
    if (want > tiles) want = tiles;
    *per = sat_cdiv(tiles, want);
    return sat_cdiv(tiles, *per);
}
extern "C" int sat_edge_conv_wgrad_nsplit(int B, int M, int N, int T) {
    if (B <= 0 || M <= 0 || N <= 0 || T <= 0) return -1;
    int per;
    return B * sat_edge_wgrad_plan(B, M > N ? M : N, T, &per);
}
// dW (M, N, K) of y = conv1d(act(x), W) (stride 1, pad) as slabs [nsplit][M * N * K] in torch order, one of M / N <= 2.
// dy (B, M, T), x (B, N, T) pre-activation; alpha / beta: log parameters of x's SnakeBeta (only with M <= 2: the wide operand) | NULL.
// rowsum [M][nsplit] | NULL: per-slab sums of dy's rows (bias gradient; only with N <= 2: dy is the streamed operand).
extern "C" int sat_edge_conv_wgrad(const float* dy, const float* x, const float* alpha, const float* beta, float* partial, float* rowsum,
                                   int B, int M, int N, int T, int K, int pad, void* stream) {
    if (B <= 0 || T <= 0 || !dy || !x || !partial) { sat_set_error("sat_edge_conv_wgrad: empty shape / missing buffer"); return 1; }
    if (!sat_edge_conv_ok(N, M, K, 1, 1, pad) || 2 * pad != K - 1) { sat_set_error("sat_edge_conv_wgrad: shape not served (sat_edge_conv_ok; 2 * pad == K - 1)"); return 1; }
    if ((alpha == nullptr) != (beta == nullptr)) { sat_set_error("sat_edge_conv_wgrad: alpha and beta come together"); return 1; }
    SatEdgeParams p{};
    p.B = B; p.T = T; p.K = K; p.y = partial;
    const bool narrow_in = N <= 2;
    int S;
    if (narrow_in) {             // wide = dy (M rows), narrow = x: acc[m][n][k] = sum dy[m][t] x[n][t + k - pad]
        if (alpha) { sat_set_error("sat_edge_conv_wgrad: SnakeBeta on a narrow input is not served"); return 1; }
        p.x = dy; p.w = x; p.Cin = M; S = N; p.pad = pad; p.flip = 0; p.part_da = rowsum;
    } else {                     // wide = act(x) (N rows), narrow = dy: acc[n][m][k'] = sum act(x)[n][u] dy[m][u + k' - (K-1-pad)], k = K-1-k'
        if (rowsum) { sat_set_error("sat_edge_conv_wgrad: row sums come with the narrow-input form only"); return 1; }
        p.x = x; p.w = dy; p.Cin = N; S = M; p.pad = K - 1 - pad; p.flip = 1; p.alpha = alpha; p.beta = beta;
    }
    p.nsplit_t = sat_edge_wgrad_plan(B, p.Cin, T, &p.tiles_per_split);
    dim3 grid(p.nsplit_t, sat_cdiv(p.Cin, SAT_EC_ROWS), B);
    if (S == 1) SAT_LAUNCH(sat_edge_wgrad_kernel<1>, grid, dim3(256), stream, p);
    else SAT_LAUNCH(sat_edge_wgrad_kernel<2>, grid, dim3(256), stream, p);
    return sat_check_launch("sat_edge_conv_wgrad");
}
