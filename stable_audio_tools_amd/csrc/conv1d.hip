// conv1d.hip — Oobleck 1-D convolution as an LDS-tiled implicit GEMM on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, so results track the reference's fp32 F.conv1d
// to rounding order only).
//
// Covers, with one kernel:
//   * WNConv1d k in {1,3,7}, dilation {1,3,9}, symmetric padding  (autoencoders.py:58-83, :298, :311, :333, :353)
//   * the EncoderBlock down-conv: k = 2*stride, stride s, pad ceil(s/2) (autoencoders.py:245-247)
//   * their data-gradients: dgrad of a stride-1 conv is the same conv with flipped/transposed
//     weights; dgrad of the DecoderBlock ConvTranspose1d is the strided conv (autoencoders.py:267)
// Fused around the GEMM:
//   prologue : SnakeBeta on the input while it is staged into LDS (blocks.py:291-292) — the
//              activation tensor is never materialised in HBM
//   epilogue : + bias, + residual (ResidualUnit skip, autoencoders.py:83), tanh (OobleckDecoder
//              final_tanh, autoencoders.py:354), or — in backward — multiply by dsnake/dx of the
//              upstream activation and reduce the SnakeBeta parameter gradients per tile.
//
// GEMM view: M = Cout (MFMA rows), N = time (MFMA cols, contiguous in HBM and across lanes),
// K = (ci, tap).  The two k-slots of the 32x32x2 MFMA (lane>>5) are two input channels at one tap.
#include "conv_common.h"

struct SatConvTile {
    int ci_t;  // input channels per K-chunk (even)
    int cs;    // activation-slab channel stride (floats)
    int L;     // strided mode: per-phase row length; stride-1: unused
    int nj;    // input samples staged per channel per chunk
};

struct SatConvLaunch {
    SatConvParams p;
    SatConvTile t;
};

__global__ void __launch_bounds__(256) sat_conv1d_kernel(SatConvLaunch a) {
    const SatConvParams& p = a.p;
    __shared__ float w_lds[SAT_W_ROWS][SAT_CO_T];  // [(c, tap)][co]
    __shared__ __attribute__((aligned(16))) float a_lds[SAT_A_FLOATS];   // [c][...] activation slab
    __shared__ float red_lds[2][2][SAT_CO_T];      // [quantity][t-wave][co] (backward epilogue)
    __shared__ float ep_lds[3][SAT_CO_T];          // per-row epilogue constants: bias, e^alpha2, e^beta2

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int t0 = blockIdx.x * SAT_T_T;
    const int co0 = blockIdx.y * SAT_CO_T;
    const int b = blockIdx.z;
    const int co_w = (wave >> 1) * 64, t_w = (wave & 1) * 64;

    const int K = p.K, S = p.stride, dil = p.dil;
    const int CI_T = a.t.ci_t, cs = a.t.cs, L = a.t.L, nj = a.t.nj;
    const int tin0 = t0 * S - p.pad;
    const float* xb = p.x + (size_t)b * p.Cin * p.Tin;

    const bool wave_on = (co0 + co_w) < p.Cout;         // wave-uniform
    const bool mi1_on = (co0 + co_w + 32) < p.Cout;     // wave-uniform

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
    // (visibility of ep_lds is ordered by the barriers of the K loop; Cin >= 1 guarantees one pass)

    // fixed staging channel per thread
    const int tpc = 256 / CI_T;
    const int sc = tid / tpc, sj0 = tid - sc * tpc;

    for (int ci0 = 0; ci0 < p.Cin; ci0 += CI_T) {
        // ---- stage activations (snake applied once per element here) ----
        if (K == 1 && S == 1 && CI_T == 32 && p.pad == 0 && (p.Tin & 3) == 0) {
            // 1x1 convs (HBM-bound: the k1 conv of every ResidualUnit and its data-gradient): 32 channels x 128
            // samples per chunk as 4 float4 per thread, 32 lanes x 16 B = 512 contiguous bytes per channel row
            float4 v[4];
            float sa[4], sib[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = (tid >> 5) + 8 * u, t4 = (tid & 31) * 4;
                const int ci = ci0 + c, tin = t0 + t4;
                float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
                sa[u] = 1.0f;
                sib[u] = 0.0f;
                if (ci < p.Cin) {
                    const float* src = xb + (size_t)ci * p.Tin + tin;
                    if (tin + 3 < p.Tin) {
                        q = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (tin + 0 < p.Tin) q.x = src[0];
                        if (tin + 1 < p.Tin) q.y = src[1];
                        if (tin + 2 < p.Tin) q.z = src[2];
                    }
                    if (p.alpha) {
                        sa[u] = expf(p.alpha[ci]);
                        sib[u] = 1.0f / (expf(p.beta[ci]) + 1e-9f);
                    }
                }
                v[u] = q;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = (tid >> 5) + 8 * u, t4 = (tid & 31) * 4;
                float4 q = v[u];
                if (p.alpha) {
                    q.x = sat_snake(q.x, sa[u], sib[u]);
                    q.y = sat_snake(q.y, sa[u], sib[u]);
                    q.z = sat_snake(q.z, sa[u], sib[u]);
                    q.w = sat_snake(q.w, sa[u], sib[u]);
                }
                *reinterpret_cast<float4*>(&a_lds[c * cs + t4]) = q;
            }
        } else if (sc < CI_T) {
            const int ci = ci0 + sc;
            const bool ch_ok = ci < p.Cin;
            float sa = 1.0f, sib = 0.0f;
            const bool use_snake = (p.alpha != nullptr) && ch_ok;
            if (use_snake) {
                sa = expf(p.alpha[ci]);
                sib = 1.0f / (expf(p.beta[ci]) + 1e-9f);
            }
            const float* xr = xb + (size_t)ci * p.Tin;
            float* arow = a_lds + sc * cs;
            // loads are issued in batches of 8 per thread BEFORE any of them is consumed, so a chunk costs one
            // HBM round trip instead of one per element
            for (int jb = sj0; jb < nj; jb += 8 * tpc) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = jb + u * tpc;
                    const int tin = tin0 + j;
                    v[u] = (ch_ok && j < nj && tin >= 0 && tin < p.Tin) ? xr[tin] : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = jb + u * tpc;
                    if (j < nj) {
                        const float o = use_snake ? sat_snake(v[u], sa, sib) : v[u];   // snake(0) == 0: padding stays 0
                        int pos = j;
                        if (S != 1) {
                            const int q = j / S;
                            pos = (j - q * S) * L + q;
                        }
                        arow[pos] = o;
                    }
                }
            }
        }
        // ---- stage weights: rows (c, tap) of the packed [Cin][K][Cout] tensor (<= 64 rows -> <= 8 float4 per thread) ----
        {
            const int nrows = CI_T * K;
            const float* wbase = p.w + (size_t)ci0 * K * p.Cout;
            const bool vec_ok = ((p.Cout & 3) == 0);
            float4 wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = tid + u * 256;
                const int r = idx >> 5, c4 = (idx & 31) * 4;
                const int co = co0 + c4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < nrows && ci0 + r / K < p.Cin) {
                    const float* src = wbase + (size_t)r * p.Cout + co;
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
                const int r = idx >> 5, c4 = (idx & 31) * 4;
                if (r < nrows) *reinterpret_cast<float4*>(&w_lds[r][c4]) = wv[u];
            }
        }
        __syncthreads();

        if (wave_on) {
            // Two channel pairs (8 MFMAs) per step; the operands of step i+1 are fetched from LDS before the
            // MFMAs of step i issue, so the ds_read latency hides under 512 matrix-pipe cycles.  CI_T is a
            // multiple of 4 and the slabs are zero-filled past Cin, so rounding the pair count up is safe.
            int npairs = (p.Cin - ci0 + 1) >> 1;
            if (npairs > (CI_T >> 1)) npairs = CI_T >> 1;
            const int npr = (npairs + 1) >> 1;       // steps per tap
            const int nsteps = K * npr;
            float an[2][2], bn[2][2];                // [pair in step][mi | ni]
            auto fetch = [&](int step) {
                const int tap = step / npr, cq = step - tap * npr;
                int toff;
                if (S == 1) {
                    toff = tap * dil;
                } else {
                    const int dq = tap / S;
                    toff = (tap - dq * S) * L + dq;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int c = 2 * (2 * cq + u) + hi;
                    const float* wr = &w_lds[c * K + tap][co_w + l31];
                    const float* ar = a_lds + c * cs + toff + t_w + l31;
                    an[u][0] = wr[0];
                    an[u][1] = wr[32];
                    bn[u][0] = ar[0];
                    bn[u][1] = ar[32];
                }
            };
            fetch(0);
            for (int step = 0; step < nsteps; ++step) {
                float ac[2][2], bc[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    ac[u][0] = an[u][0];
                    ac[u][1] = an[u][1];
                    bc[u][0] = bn[u][0];
                    bc[u][1] = bn[u][1];
                }
                if (step + 1 < nsteps) fetch(step + 1);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc[0][0] = sat_mfma_32x32x2_f32(ac[u][0], bc[u][0], acc[0][0]);
                    acc[0][1] = sat_mfma_32x32x2_f32(ac[u][0], bc[u][1], acc[0][1]);
                    if (mi1_on) {
                        acc[1][0] = sat_mfma_32x32x2_f32(ac[u][1], bc[u][0], acc[1][0]);
                        acc[1][1] = sat_mfma_32x32x2_f32(ac[u][1], bc[u][1], acc[1][1]);
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---------------------------------- epilogue ----------------------------------
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
            const size_t nrows = (size_t)p.B * gridDim.x;  // layout [Cout][rows]: reduced by sat_rowsum
            p.part_da[(size_t)(co0 + tid) * nrows + row] = red_lds[0][0][tid] + red_lds[0][1][tid];
            p.part_db[(size_t)(co0 + tid) * nrows + row] = red_lds[1][0][tid] + red_lds[1][1][tid];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
extern "C" int sat_conv1d_partial_rows(int B, int Tout) { return B * sat_cdiv(Tout, SAT_T_T); }

extern "C" int sat_conv1d(const float* x, const float* w_packed, const float* bias, const float* alpha,
                          const float* beta, const float* res, float* y, const float* x2,
                          const float* alpha2, const float* beta2, float* part_da, float* part_db, int B,
                          int Cin, int Cout, int Tin, int Tout, int K, int stride, int dil, int pad,
                          int tanh_out, void* stream) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || Tin <= 0 || Tout <= 0) { sat_set_error("sat_conv1d: empty shape"); return 1; }
    if (K < 1 || stride < 1 || dil < 1) { sat_set_error("sat_conv1d: bad kernel geometry"); return 1; }
    if (stride > 1 && dil != 1) { sat_set_error("sat_conv1d: strided conv requires dilation 1"); return 1; }
    if ((alpha == nullptr) != (beta == nullptr)) { sat_set_error("sat_conv1d: alpha/beta must both be given"); return 1; }
    if (x2 && (!alpha2 || !beta2 || !part_da || !part_db)) { sat_set_error("sat_conv1d: backward epilogue needs alpha2/beta2/partials"); return 1; }
    SatConvLaunch a;
    a.p = SatConvParams{x, w_packed, bias, alpha, beta, res, y, x2, alpha2, beta2, part_da, part_db,
                        B, Cin, Cout, Tin, Tout, K, stride, dil, pad, tanh_out};
    int ci_t = (SAT_W_ROWS / K) & ~3;   // multiple of 4: the MFMA loop consumes two channel pairs per step
    if (ci_t > 32) ci_t = 32;
    if (ci_t < 4) { sat_set_error("sat_conv1d: kernel too wide (K > 16)"); return 1; }
    const int nj = (SAT_T_T - 1) * stride + (K - 1) * dil + 1;
    int cs, L = 0;
    if (stride == 1) {
        cs = nj;
    } else {
        const int lmin = SAT_T_T + (K - 1) / stride;
        L = lmin;
        if (32 % stride == 0) {
            const int want = 32 / stride;
            L = lmin + (((want - lmin) % 32) + 32) % 32;
        }
        cs = stride * L;
    }
    while (ci_t > 4 && ci_t * cs > SAT_A_FLOATS) ci_t -= 4;
    if (ci_t * cs > SAT_A_FLOATS) { sat_set_error("sat_conv1d: receptive field too large for the LDS slab"); return 1; }
    if (256 / ci_t < 1) { sat_set_error("sat_conv1d: internal tiling error"); return 1; }
    a.t = SatConvTile{ci_t, cs, L, nj};
    dim3 grid(sat_cdiv(Tout, SAT_T_T), sat_cdiv(Cout, SAT_CO_T), B);
    SAT_LAUNCH(sat_conv1d_kernel, grid, dim3(256), stream, a);
    return sat_check_launch("sat_conv1d");
}
