// convtr1d.hip — polyphase transposed 1-D convolution (k = 2*stride) on the fp32 matrix cores.
//
// Covers
//   * DecoderBlock up-sampling WNConvTranspose1d(k=2s, stride=s, pad=ceil(s/2))  (autoencoders.py:266-268)
//   * the data-gradient of the EncoderBlock strided down-conv                    (autoencoders.py:245-247)
// with the same fusions as conv1d.hip (SnakeBeta prologue; bias / dsnake+param-grad epilogue).
//
// Polyphase identity (torch ConvTranspose1d: out[t] = sum_i sum_k x[i] w[k], t = i*s - pad + k):
// with u = t + pad, q = u / s, r = u % s only taps k = r (input q) and k = r + s (input q-1)
// contribute, so every output phase r is a 2-tap stride-1 GEMM with its own weights.  One wave
// owns a 32(co) x 32(q) MFMA tile for ALL s phases: each lane then holds s consecutive output
// samples of a row, so stores are contiguous in time and no phase is ever computed with zero taps.
#include "conv_common.h"

struct SatConvTrLaunch {
    SatConvParams p;  // p.w packed as [r][j][Cin][Cout]  (r = phase, j = tap index)
    int ci_t;
};

template <int S, int MI, int NI>
__global__ void __launch_bounds__(256) sat_convtr1d_kernel(SatConvTrLaunch a) {
    const SatConvParams& p = a.p;
    constexpr int WQ = (MI == 2) ? 2 : 1;   // waves along q
    constexpr int QB = WQ * NI * 32;        // q per block
    __shared__ float w_lds[SAT_W_ROWS][SAT_CO_T];  // [(r*2+j)*CI_T + c][co]
    __shared__ float a_lds[32][QB + 4];            // [c][q - Q0 + 1]
    __shared__ float red_lds[2][2][SAT_CO_T];
    __shared__ float ep_lds[3][SAT_CO_T];  // per-row epilogue constants: bias, e^alpha2, e^beta2

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int Q0 = blockIdx.x * QB;
    const int co0 = blockIdx.y * SAT_CO_T;
    const int b = blockIdx.z;
    const int wq = (WQ == 2) ? (wave & 1) : 0;
    const int co_w = (WQ == 2) ? (wave >> 1) * 64 : wave * 32;
    const int q_w = wq * NI * 32;
    const int CI_T = a.ci_t;
    const float* xb = p.x + (size_t)b * p.Cin * p.Tin;

    const bool wave_on = (co0 + co_w) < p.Cout;
    const bool mi1_on = (MI == 2) && (co0 + co_w + 32) < p.Cout;

    f32x16 acc[S][MI][NI];
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][i][j][r] = 0.0f;

    if (tid < SAT_CO_T) {
        const int co = co0 + tid;
        const bool ok = co < p.Cout;
        ep_lds[0][tid] = (ok && p.bias) ? p.bias[co] : 0.0f;
        ep_lds[1][tid] = (ok && p.x2) ? expf(p.alpha2[co]) : 1.0f;
        ep_lds[2][tid] = (ok && p.x2) ? expf(p.beta2[co]) : 1.0f;
    }

    const int tpc = 256 / CI_T;
    const int sc = tid / tpc, sj0 = tid - sc * tpc;

    for (int ci0 = 0; ci0 < p.Cin; ci0 += CI_T) {
        if (sc < CI_T) {
            const int ci = ci0 + sc;
            const bool ch_ok = ci < p.Cin;
            float sa = 1.0f, sib = 0.0f;
            const bool use_snake = (p.alpha != nullptr) && ch_ok;
            if (use_snake) {
                sa = expf(p.alpha[ci]);
                sib = 1.0f / (expf(p.beta[ci]) + 1e-9f);
            }
            const float* xr = xb + (size_t)ci * p.Tin;
            for (int j = sj0; j < QB + 1; j += tpc) {
                const int q = Q0 - 1 + j;
                float v = 0.0f;
                if (ch_ok && q >= 0 && q < p.Tin) {
                    v = xr[q];
                    if (use_snake) v = sat_snake(v, sa, sib);
                }
                a_lds[sc][j] = v;
            }
        }
        {
            // rows: (r*2+j)*CI_T + c  <-  packed[((r*2+j)*Cin + ci0 + c)*Cout + co]
            const int nrows = 2 * S * CI_T;   // <= 64 rows -> <= 8 float4 per thread, all issued before any is stored
            const bool vec_ok = ((p.Cout & 3) == 0);
            float4 wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = tid + u * 256;
                const int row = idx >> 5, c4 = (idx & 31) * 4;
                const int rj = row / CI_T, c = row - rj * CI_T;
                const int ci = ci0 + c;
                const int co = co0 + c4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < nrows && ci < p.Cin) {
                    const float* src = p.w + ((size_t)rj * p.Cin + ci) * p.Cout + co;
                    if (vec_ok && co + 3 < p.Cout) {
                        v = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (co + 0 < p.Cout) v.x = src[0];
                        if (co + 1 < p.Cout) v.y = src[1];
                        if (co + 2 < p.Cout) v.z = src[2];
                        if (co + 3 < p.Cout) v.w = src[3];
                    }
                }
                wv[u] = v;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = tid + u * 256;
                const int row = idx >> 5, c4 = (idx & 31) * 4;
                if (row < nrows) *reinterpret_cast<float4*>(&w_lds[row][c4]) = wv[u];
            }
        }
        __syncthreads();

        if (wave_on) {
            int npairs = (p.Cin - ci0 + 1) >> 1;
            if (npairs > (CI_T >> 1)) npairs = CI_T >> 1;
            for (int cp = 0; cp < npairs; ++cp) {
                const int c = 2 * cp + hi;
                float bq[NI], bqm1[NI];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    const float* ar = &a_lds[c][q_w + ni * 32 + l31];
                    bqm1[ni] = ar[0];  // input q-1
                    bq[ni] = ar[1];    // input q
                }
#pragma unroll
                for (int s = 0; s < S; ++s) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        if (mi == 1 && !mi1_on) break;
                        const float w0 = w_lds[(s * 2 + 0) * CI_T + c][co_w + mi * 32 + l31];
                        const float w1 = w_lds[(s * 2 + 1) * CI_T + c][co_w + mi * 32 + l31];
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) {
                            acc[s][mi][ni] = sat_mfma_32x32x2_f32(w0, bq[ni], acc[s][mi][ni]);
                            acc[s][mi][ni] = sat_mfma_32x32x2_f32(w1, bqm1[ni], acc[s][mi][ni]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    const bool bwd = (p.x2 != nullptr);
    if (bwd) {
        for (int i = tid; i < 2 * 2 * SAT_CO_T; i += 256) (&red_lds[0][0][0])[i] = 0.0f;
        __syncthreads();
    }
    if (wave_on) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
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
                for (int ni = 0; ni < NI; ++ni) {
                    const int q = Q0 + q_w + ni * 32 + l31;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const int t = q * S + s - p.pad;
                        if (co_ok && t >= 0 && t < p.Tout) {
                            const size_t o = ((size_t)b * p.Cout + co) * p.Tout + t;
                            float v = acc[s][mi][ni][r] + bias;
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
                }
                if (bwd) {
                    pda = sat_half_sum(pda);
                    pdb = sat_half_sum(pdb);
                    if (l31 == 0) {
                        red_lds[0][wq][col] = pda;
                        red_lds[1][wq][col] = pdb;
                    }
                }
            }
        }
    }
    if (bwd) {
        __syncthreads();
        if (tid < SAT_CO_T && co0 + tid < p.Cout) {
            const size_t row = (size_t)b * gridDim.x + blockIdx.x;
            const size_t nrows = (size_t)p.B * gridDim.x;  // layout [Cout][rows]: reduced by sat_rowsum
            p.part_da[(size_t)(co0 + tid) * nrows + row] = red_lds[0][0][tid] + red_lds[0][1][tid];
            p.part_db[(size_t)(co0 + tid) * nrows + row] = red_lds[1][0][tid] + red_lds[1][1][tid];
        }
    }
}

template <int S> struct SatTrCfg { static constexpr int MI = 1, NI = 1; };
template <> struct SatTrCfg<2> { static constexpr int MI = 2, NI = 2; };
template <> struct SatTrCfg<3> { static constexpr int MI = 2, NI = 1; };
template <> struct SatTrCfg<4> { static constexpr int MI = 2, NI = 1; };

static int sat_convtr_qb(int S) {
    switch (S) {
        case 2: return 128;
        case 3: case 4: return 64;
        default: return 32;
    }
}

// number of q-tiles: q = (t + pad) / S for t in [0, Tout)  ->  q in [0, (Tout - 1 + pad) / S]
static int sat_convtr_qtiles(int Tout, int S, int pad) { return sat_cdiv((Tout - 1 + pad) / S + 1, sat_convtr_qb(S)); }

extern "C" int sat_convtr1d_partial_rows(int B, int Tout, int stride, int pad) {
    if (stride < 2 || stride > 8) return -1;
    return B * sat_convtr_qtiles(Tout, stride, pad);
}

template <int S>
static void sat_convtr_launch(const SatConvTrLaunch& a, dim3 grid, void* stream) {
    SAT_LAUNCH((sat_convtr1d_kernel<S, SatTrCfg<S>::MI, SatTrCfg<S>::NI>), grid, dim3(256), stream, a);
}

extern "C" int sat_convtr1d(const float* x, const float* w_packed, const float* bias, const float* alpha,
                            const float* beta, const float* res, float* y, const float* x2,
                            const float* alpha2, const float* beta2, float* part_da, float* part_db,
                            int B, int Cin, int Cout, int Tin, int Tout, int K, int stride, int pad,
                            int tanh_out, void* stream) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || Tin <= 0 || Tout <= 0) { sat_set_error("sat_convtr1d: empty shape"); return 1; }
    if (stride < 2 || stride > 8) { sat_set_error("sat_convtr1d: stride must be in [2, 8]"); return 1; }
    if (K != 2 * stride) { sat_set_error("sat_convtr1d: only kernel_size == 2*stride (the Oobleck resampling convs) is supported"); return 1; }
    if ((alpha == nullptr) != (beta == nullptr)) { sat_set_error("sat_convtr1d: alpha/beta must both be given"); return 1; }
    if (x2 && (!alpha2 || !beta2 || !part_da || !part_db)) { sat_set_error("sat_convtr1d: backward epilogue needs alpha2/beta2/partials"); return 1; }
    SatConvTrLaunch a;
    a.p = SatConvParams{x, w_packed, bias, alpha, beta, res, y, x2, alpha2, beta2, part_da, part_db,
                        B, Cin, Cout, Tin, Tout, K, stride, 1, pad, tanh_out};
    int ci_t = (SAT_W_ROWS / (2 * stride)) & ~1;
    if (ci_t > 32) ci_t = 32;
    a.ci_t = ci_t;
    dim3 grid(sat_convtr_qtiles(Tout, stride, pad), sat_cdiv(Cout, SAT_CO_T), B);
    switch (stride) {
        case 2: sat_convtr_launch<2>(a, grid, stream); break;
        case 3: sat_convtr_launch<3>(a, grid, stream); break;
        case 4: sat_convtr_launch<4>(a, grid, stream); break;
        case 5: sat_convtr_launch<5>(a, grid, stream); break;
        case 6: sat_convtr_launch<6>(a, grid, stream); break;
        case 7: sat_convtr_launch<7>(a, grid, stream); break;
        case 8: sat_convtr_launch<8>(a, grid, stream); break;
    }
    return sat_check_launch("sat_convtr1d");
}
