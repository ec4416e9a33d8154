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

#define SAT_BF_KROW 72    // bf16 per weight row in LDS: 8 groups x 8 + 8 pad (144 B stride, conflict-free b128)
#define SAT_BF_AROWS 192  // max staged time rows: 128 + (K-1)*dil <= 128 + 7*9 = 191

struct SatConvBfLaunch {
    SatConvParams p;       // p.w unused; p.alpha / p.beta hold PRE-EXPONENTIATED snake constants: a = e^alpha, ib = 1/(e^beta+1e-9)
    const short* w_hi;     // [nchunks][CoutPad][8][8]
    const short* w_lo;
    int cout_pad;
};

SAT_DEVICE void sat_split2(float x, short* hi, short* lo) {
    const short h = sat_f32_to_bf16(x);
    *hi = h;
    *lo = sat_f32_to_bf16(x - sat_bf16_to_f32(h));
}

__global__ void __launch_bounds__(256) sat_conv1d_bf16x3_kernel(SatConvBfLaunch a) {
    const SatConvParams& p = a.p;
    __shared__ __attribute__((aligned(16))) short w_lds[2][SAT_CO_T][SAT_BF_KROW];   // [plane][co][g*8+e]
    __shared__ __attribute__((aligned(16))) short a_lds[2][SAT_BF_AROWS][8];         // [plane][time row][8 ci]
    __shared__ float red_lds[2][2][SAT_CO_T];
    __shared__ float ep_lds[3][SAT_CO_T];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int t0 = blockIdx.x * SAT_T_T;
    const int co0 = blockIdx.y * SAT_CO_T;
    const int b = blockIdx.z;
    const int co_w = (wave >> 1) * 64, t_w = (wave & 1) * 64;
    const int K = p.K, dil = p.dil;
    const int nrows = SAT_T_T + (K - 1) * dil;
    const int tin0 = t0 - p.pad;
    const float* xb = p.x + (size_t)b * p.Cin * p.Tin;
    const bool wave_on = (co0 + co_w) < p.Cout;
    const bool mi1_on = (co0 + co_w + 32) < p.Cout;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    if (tid < SAT_CO_T) {
        const int co = co0 + tid;
        const bool ok = co < p.Cout;
        ep_lds[0][tid] = (ok && p.bias) ? p.bias[co] : 0.0f;
        ep_lds[1][tid] = (ok && p.x2) ? expf(p.alpha2[co]) : 1.0f;
        ep_lds[2][tid] = (ok && p.x2) ? expf(p.beta2[co]) : 1.0f;
    }

    const int nchunks = p.Cin >> 3;
    // Register-staged software pipeline: the global loads of chunk c+1 are issued right after the barrier that
    // publishes chunk c, so their HBM/L2 latency runs under chunk c's MFMA phase; they are converted and written
    // to LDS after the phase's closing barrier.
    const int srow = tid;                                 // activation time row owned by this thread (if < nrows)
    const int stin = tin0 + srow;
    const bool srow_ok = (srow < nrows) && stin >= 0 && stin < p.Tin;
    float av[8];
    bf16x8 wv[8];
    auto issue_loads = [&](int c) {
        const int ci0 = c * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = srow_ok ? xb[(size_t)(ci0 + e) * p.Tin + stin] : 0.0f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = tid + u * 256;                 // plane = idx>>10, co = (idx>>3)&127, part = idx&7
            const int pl = idx >> 10, co = (idx >> 3) & 127, part = idx & 7;
            const short* src = (pl ? a.w_lo : a.w_hi) + (((size_t)c * a.cout_pad + co0 + co) * 64 + part * 8);
            wv[u] = *reinterpret_cast<const bf16x8*>(src);
        }
    };
    auto write_lds = [&](int c) {
        const int ci0 = c * 8;
        if (srow < nrows) {
            bf16x8 vh, vl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float o = av[e];
                if (p.alpha) o = sat_snake(o, p.alpha[ci0 + e], p.beta[ci0 + e]);   // pre-exponentiated constants
                short h, l;
                sat_split2(o, &h, &l);
                vh[e] = h;
                vl[e] = l;
            }
            *reinterpret_cast<bf16x8*>(&a_lds[0][srow][0]) = vh;
            *reinterpret_cast<bf16x8*>(&a_lds[1][srow][0]) = vl;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = tid + u * 256;
            const int pl = idx >> 10, co = (idx >> 3) & 127, part = idx & 7;
            *reinterpret_cast<bf16x8*>(&w_lds[pl][co][part * 8]) = wv[u];
        }
    };

    issue_loads(0);
    for (int c = 0; c < nchunks; ++c) {
        write_lds(c);
        __syncthreads();
        if (c + 1 < nchunks) issue_loads(c + 1);

        if (wave_on) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int g = 2 * ks + hi;                 // k-slots 0-7 <- group 2ks (lanes 0-31), 8-15 <- group 2ks+1
                const int tap = (g < K) ? g : (K - 1);     // group >= K is a zero-weight pad; keep the row in range
                bf16x8 wa[2][2], xa[2][2];                 // [mi|ni][plane]
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    wa[0][pl] = *reinterpret_cast<const bf16x8*>(&w_lds[pl][co_w + l31][g * 8]);
                    wa[1][pl] = *reinterpret_cast<const bf16x8*>(&w_lds[pl][co_w + 32 + l31][g * 8]);
                    xa[0][pl] = *reinterpret_cast<const bf16x8*>(&a_lds[pl][t_w + l31 + tap * dil][0]);
                    xa[1][pl] = *reinterpret_cast<const bf16x8*>(&a_lds[pl][t_w + 32 + l31 + tap * dil][0]);
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
        }
        __syncthreads();
    }

    // ---------------------------------- epilogue (as conv1d.hip) ----------------------------------
    const bool bwd = (p.x2 != nullptr);
    if (bwd) {
        for (int i = tid; i < 2 * 2 * SAT_CO_T; i += 256) (&red_lds[0][0][0])[i] = 0.0f;
        __syncthreads();
    }
    if (wave_on) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            if (mi == 1 && !mi1_on) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = co_w + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int co = co0 + col;
                const bool co_ok = co < p.Cout;
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
                        red_lds[0][wave & 1][col] = pda;
                        red_lds[1][wave & 1][col] = pdb;
                    }
                }
            }
        }
    }
    if (bwd) {
        __syncthreads();
        if (tid < SAT_CO_T && co0 + tid < p.Cout) {
            const size_t row = (size_t)b * gridDim.x + blockIdx.x;
            const size_t nrows_p = (size_t)p.B * gridDim.x;
            p.part_da[(size_t)(co0 + tid) * nrows_p + row] = red_lds[0][0][tid] + red_lds[0][1][tid];
            p.part_db[(size_t)(co0 + tid) * nrows_p + row] = red_lds[1][0][tid] + red_lds[1][1][tid];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight preparation: torch conv weight w[Cout][Cin][K] (fp32) -> hi/lo bf16 planes
//   [chunk = ci/8][co (padded to 128)][group g][e]  with value W_eff[co][8*chunk + e][tap g] (0 for g >= K, co >= Cout)
//   mode 0 (forward):        W_eff[co][ci][tap] = w[co][ci][tap]
//   mode 1 (data-gradient):  the flipped/transposed conv (in = Cout, out = Cin): W_eff[o][i][tap] = w[i][o][K-1-tap]
// ------------------------------------------------------------------------------------------------
struct SatPackBfParams {
    const float* w;
    short* hi;
    short* lo;
    int D0, D1, K, mode, n_out, n_in, out_pad;
    long long total;
};
__global__ void __launch_bounds__(256) sat_pack_bf16x3_kernel(SatPackBfParams p) {
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= p.total) return;
    const int e = (int)(o & 7), g = (int)((o >> 3) & 7);
    const long long q = o >> 6;
    const int co = (int)(q % p.out_pad), chunk = (int)(q / p.out_pad);
    const int ci = chunk * 8 + e;
    float v = 0.0f;
    if (g < p.K && co < p.n_out && ci < p.n_in) {
        if (p.mode == 0) v = p.w[((size_t)co * p.D1 + ci) * p.K + g];
        else v = p.w[((size_t)ci * p.D1 + co) * p.K + (p.K - 1 - g)];
    }
    short h, l;
    sat_split2(v, &h, &l);
    p.hi[o] = h;
    p.lo[o] = l;
}

extern "C" long long sat_pack_weights_bf16x3_size(int D0, int D1, int K, int mode) {
    const int n_out = mode == 0 ? D0 : D1, n_in = mode == 0 ? D1 : D0;
    if (K < 1 || K > 8 || (n_in & 7)) return -1;
    return (long long)(n_in / 8) * (sat_cdiv(n_out, SAT_CO_T) * SAT_CO_T) * 64;
}
extern "C" int sat_pack_weights_bf16x3(const float* w, short* hi, short* lo, int D0, int D1, int K, int mode, void* stream) {
    const long long total = sat_pack_weights_bf16x3_size(D0, D1, K, mode);
    if (total <= 0 || (mode != 0 && mode != 1)) { sat_set_error("sat_pack_weights_bf16x3: needs K <= 8, in-channels % 8 == 0, mode 0|1"); return 1; }
    SatPackBfParams p{w, hi, lo, D0, D1, K, mode, mode == 0 ? D0 : D1, mode == 0 ? D1 : D0, 0, total};
    p.out_pad = sat_cdiv(p.n_out, SAT_CO_T) * SAT_CO_T;
    SAT_LAUNCH(sat_pack_bf16x3_kernel, dim3((unsigned)sat_cdivll(total, 256)), dim3(256), stream, p);
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

// Same contract as sat_conv1d (stride 1), with the weights given as sat_pack_weights_bf16x3 planes and the
// SnakeBeta constants given pre-exponentiated (sat_snake_consts), or NULL for no activation.
extern "C" int sat_conv1d_bf16x3(const float* x, const short* w_hi, const short* w_lo, const float* bias,
                                 const float* snake_a, const float* snake_ib, const float* res, float* y,
                                 const float* x2, const float* alpha2, const float* beta2, float* part_da,
                                 float* part_db, int B, int Cin, int Cout, int Tin, int Tout, int K, int dil, int pad,
                                 int tanh_out, void* stream) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || Tin <= 0 || Tout <= 0) { sat_set_error("sat_conv1d_bf16x3: empty shape"); return 1; }
    if (K < 1 || K > 8 || dil < 1 || (Cin & 7)) { sat_set_error("sat_conv1d_bf16x3: needs K <= 8 and Cin % 8 == 0"); return 1; }
    if (SAT_T_T + (K - 1) * dil > SAT_BF_AROWS) { sat_set_error("sat_conv1d_bf16x3: receptive field too large for the LDS slab"); return 1; }
    if ((snake_a == nullptr) != (snake_ib == nullptr)) { sat_set_error("sat_conv1d_bf16x3: snake constants must both be given"); return 1; }
    if (x2 && (!alpha2 || !beta2 || !part_da || !part_db)) { sat_set_error("sat_conv1d_bf16x3: backward epilogue needs alpha2/beta2/partials"); return 1; }
    SatConvBfLaunch a;
    a.p = SatConvParams{x, nullptr, bias, snake_a, snake_ib, res, y, x2, alpha2, beta2, part_da, part_db,
                        B, Cin, Cout, Tin, Tout, K, 1, dil, pad, tanh_out};
    a.w_hi = w_hi;
    a.w_lo = w_lo;
    a.cout_pad = sat_cdiv(Cout, SAT_CO_T) * SAT_CO_T;
    dim3 grid(sat_cdiv(Tout, SAT_T_T), sat_cdiv(Cout, SAT_CO_T), B);
    SAT_LAUNCH(sat_conv1d_bf16x3_kernel, grid, dim3(256), stream, a);
    return sat_check_launch("sat_conv1d_bf16x3");
}
