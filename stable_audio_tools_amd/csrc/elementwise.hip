// elementwise.hip — HBM-bound helpers of the Oobleck path: weight-norm fold/grad, weight packing,
// VAE bottleneck sample/KL, fused AdamW on flat parameter buffers, status plumbing.
#include "sat_device.h"

#include <string.h>

// ---------------------------------------------------------------------------------------------
// status plumbing (C-ABI contract: int status, last-error string, never throw)
// ---------------------------------------------------------------------------------------------
static thread_local char g_sat_err[512] = "";
void sat_set_error(const char* msg) {
    strncpy(g_sat_err, msg, sizeof(g_sat_err) - 1);
    g_sat_err[sizeof(g_sat_err) - 1] = 0;
}
int sat_check_launch(const char* what) {
    const auto e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[512];
        snprintf(buf, sizeof(buf), "%s: kernel launch failed: %s", what, hipGetErrorString(e));
        sat_set_error(buf);
        return 2;
    }
    return 0;
}
extern "C" const char* sat_last_error() { return g_sat_err; }
// compute units of the current device, queried once per thread (persistent kernels launch one workgroup per CU); 256 on the simulator
int sat_cu_count() {
#if defined(SAT_HIPEMU)
    return 256;
#else
    static thread_local int dev_cached = -1, cus = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != dev_cached) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n;
        dev_cached = dev;
    }
    return cus;
#endif
}

extern "C" int sat_abi_version() { return 1; }
extern "C" int sat_is_simulator() {
#if defined(SAT_HIPEMU)
    return 1;
#else
    return 0;
#endif
}

// ---------------------------------------------------------------------------------------------
// weight norm  (torch.nn.utils.weight_norm, dim=0: w = g * v / ||v||, norm over all dims but 0;
// reference call sites autoencoders.py:23-27.  For ConvTranspose1d dim 0 is the INPUT channel.)
// ---------------------------------------------------------------------------------------------
struct SatWnParams {
    const float* v;     // (D0, R)
    const float* g;     // (D0)
    const float* dw;    // (D0, R)   (grad only)
    float* w;           // (D0, R)   (fold)  |  dv (grad)
    float* norm;        // (D0)      (fold: out) | (grad: in)
    float* dg;          // (D0)      (grad)
    int D0, R;
};

SAT_DEVICE float sat_block_sum_256(float s, float* red) {
    s = sat_wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) sat_wn_fold_kernel(SatWnParams p) {
    __shared__ float red[4];
    const int d = blockIdx.x;
    const float* v = p.v + (size_t)d * p.R;
    float s = 0.f;
    for (int i = threadIdx.x; i < p.R; i += 256) s += v[i] * v[i];
    s = sat_block_sum_256(s, red);
    const float nrm = sqrtf(s);
    const float sc = p.g[d] / nrm;
    float* w = p.w + (size_t)d * p.R;
    for (int i = threadIdx.x; i < p.R; i += 256) w[i] = v[i] * sc;
    if (threadIdx.x == 0) p.norm[d] = nrm;
}

__global__ void __launch_bounds__(256) sat_wn_grad_kernel(SatWnParams p) {
    __shared__ float red[4];
    const int d = blockIdx.x;
    const float* v = p.v + (size_t)d * p.R;
    const float* dw = p.dw + (size_t)d * p.R;
    float s = 0.f;
    for (int i = threadIdx.x; i < p.R; i += 256) s += v[i] * dw[i];
    s = sat_block_sum_256(s, red);
    const float nrm = p.norm[d];
    const float g = p.g[d];
    const float dgv = s / nrm;
    const float c1 = g / nrm, c2 = g * s / (nrm * nrm * nrm);
    float* dv = p.w + (size_t)d * p.R;
    for (int i = threadIdx.x; i < p.R; i += 256) dv[i] = c1 * dw[i] - c2 * v[i];
    if (threadIdx.x == 0) p.dg[d] = dgv;
}

// Weight-norm gradient straight from the weight-gradient kernels' split slabs (round 5): dW[d][n][k] = sum_z partial[z * count +
// d * so_m + n * so_n + k * so_k] is summed here, row by row, instead of by sat_reduce_splits (+ a torch permute for the tap-major
// slabs of the k = 7 kernels) in front of sat_wn_grad — one launch per conv instead of two or three, and dW itself never touches HBM.
// One workgroup per row d of v (D0, R = N * K, torch layout r = n * K + k).  The row of dW is staged in LDS (R <= SAT_WN_ROW_CAP
// floats; longer rows are summed twice), walked in the slabs' own order so that the slab reads stay contiguous: tap-major slabs
// (so_n == 1) as (k, n), everything else as (n, k).  Slabs are added in index order: deterministic.
#define SAT_WN_ROW_CAP 16384
struct SatWnSplitParams {
    const float* partial;
    const float* v;
    const float* g;
    const float* norm;
    float* dv;
    float* dg;
    long long count, so_m, so_n, so_k;
    int nsplit, D0, N, K;
    const float* bias_partial;      // (D0, bias_cols) per-split sums of dy (the weight-gradient kernel's, or sat_rowsum's first pass) | NULL
    float* dbias;                   // (D0): their sums — the bias gradient's last reduction rides along
    int bias_cols;
};
#define SAT_WN_THREADS 1024
// sum over the slabs z = z0, z0 + zstep, ... of one element / of four consecutive elements
SAT_DEVICE float sat_wn_slab_sum(const float* base, long long count, int z0, int zstep, int nsplit) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = z0;
    for (; z + 3 * zstep < nsplit; z += 4 * zstep) {
        s0 += base[(size_t)z * count];
        s1 += base[(size_t)(z + zstep) * count];
        s2 += base[(size_t)(z + 2 * zstep) * count];
        s3 += base[(size_t)(z + 3 * zstep) * count];
    }
    for (; z < nsplit; z += zstep) s0 += base[(size_t)z * count];
    return (s0 + s1) + (s2 + s3);
}
SAT_DEVICE f32x4 sat_wn_slab_sum4(const float* base, long long count, int z0, int zstep, int nsplit) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 s[8] = {zero, zero, zero, zero, zero, zero, zero, zero};        // eight 16-byte loads in flight per thread
    int z = z0;
    for (; z + 7 * zstep < nsplit; z += 8 * zstep) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] += *reinterpret_cast<const f32x4*>(base + (size_t)(z + u * zstep) * count);
    }
    for (; z < nsplit; z += zstep) s[0] += *reinterpret_cast<const f32x4*>(base + (size_t)z * count);
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}
SAT_DEVICE float sat_block_sum_1024(float s, float* red) {
    s = sat_wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < SAT_WN_THREADS / 64; ++w) t += red[w];
    return t;
}
// 16 waves per row: the narrow levels have few rows (128) and MANY slabs (256), so a row's workgroup must keep enough loads in flight
// by itself.  The slab sum is spread over the threads as (z part, item): `nparts` groups of threads each add every nparts-th slab of
// all the row's items (item = 4 consecutive elements when the layout allows 16-byte loads), park their partial rows in LDS
// [part][R], and the parts are added in index order — deterministic.
__global__ void __launch_bounds__(SAT_WN_THREADS) sat_wn_grad_splits_kernel(SatWnSplitParams p) {
    __shared__ float row[SAT_WN_ROW_CAP];
    __shared__ float red[SAT_WN_THREADS / 64];
    const int d = blockIdx.x;
    const int R = p.N * p.K;
    const float* base = p.partial + (size_t)d * p.so_m;
    const float* v = p.v + (size_t)d * R;
    const bool staged = R <= SAT_WN_ROW_CAP;
    const bool tap_major = p.so_n == 1 && p.K > 1;
    const bool torch_order = p.so_k == 1 && p.so_n == p.K;
    const bool vec = ((p.count | p.so_m) & 3) == 0 && (((uintptr_t)p.partial) & 15) == 0 &&
                     (tap_major ? ((p.N | p.so_k) & 3) == 0 : (torch_order && (R & 3) == 0));
    float s = 0.f;
    if (staged) {
        const int items = vec ? (R >> 2) : R;
        int nparts = SAT_WN_THREADS / items;
        if (nparts > 8) nparts = 8;
        if (nparts > p.nsplit) nparts = p.nsplit;
        if (nparts < 1) nparts = 1;
        while (nparts > 1 && (long long)nparts * R > SAT_WN_ROW_CAP) --nparts;
        for (int idx = threadIdx.x; idx < items * nparts; idx += SAT_WN_THREADS) {
            const int zp = idx / items, it = idx - zp * items;
            float* dst = row + (size_t)zp * R;
            if (vec) {
                if (tap_major) {                                   // four consecutive n of one tap
                    const int nq = p.N >> 2;
                    const int k = it / nq, n = (it - k * nq) << 2;
                    const f32x4 a = sat_wn_slab_sum4(base + (size_t)k * p.so_k + n, p.count, zp, nparts, p.nsplit);
                    dst[(n + 0) * p.K + k] = a[0];
                    dst[(n + 1) * p.K + k] = a[1];
                    dst[(n + 2) * p.K + k] = a[2];
                    dst[(n + 3) * p.K + k] = a[3];
                } else {                                           // torch order: four consecutive r
                    const f32x4 a = sat_wn_slab_sum4(base + 4 * it, p.count, zp, nparts, p.nsplit);
                    *reinterpret_cast<f32x4*>(dst + 4 * it) = a;
                }
            } else {
                int n, k;
                if (tap_major) { k = it / p.N; n = it - k * p.N; }
                else           { n = it / p.K; k = it - n * p.K; }
                dst[n * p.K + k] = sat_wn_slab_sum(base + (size_t)n * p.so_n + (size_t)k * p.so_k, p.count, zp, nparts, p.nsplit);
            }
        }
        __syncthreads();
        for (int r = threadIdx.x; r < R; r += SAT_WN_THREADS) {
            float a = row[r];
            for (int zp = 1; zp < nparts; ++zp) a += row[(size_t)zp * R + r];
            row[r] = a;                                            // (every thread touches its own r of every part only)
            s += v[r] * a;
        }
    } else {
        for (int r = threadIdx.x; r < R; r += SAT_WN_THREADS) {
            const int n = r / p.K, k = r - n * p.K;
            s += v[r] * sat_wn_slab_sum(base + (size_t)n * p.so_n + (size_t)k * p.so_k, p.count, 0, 1, p.nsplit);
        }
    }
    s = sat_block_sum_1024(s, red);
    const float nrm = p.norm[d];
    const float g = p.g[d];
    const float c1 = g / nrm, c2 = g * s / (nrm * nrm * nrm);
    float* dv = p.dv + (size_t)d * R;
    for (int r = threadIdx.x; r < R; r += SAT_WN_THREADS) {
        float dw;
        if (staged) dw = row[r];
        else {
            const int n = r / p.K, k = r - n * p.K;
            dw = sat_wn_slab_sum(base + (size_t)n * p.so_n + (size_t)k * p.so_k, p.count, 0, 1, p.nsplit);
        }
        dv[r] = c1 * dw - c2 * v[r];
    }
    if (threadIdx.x == 0) p.dg[d] = s / nrm;
    if (p.bias_partial) {
        const float* bp = p.bias_partial + (size_t)d * p.bias_cols;
        float b = 0.f;
        for (int i = threadIdx.x; i < p.bias_cols; i += SAT_WN_THREADS) b += bp[i];
        b = sat_block_sum_1024(b, red);
        if (threadIdx.x == 0) p.dbias[d] = b;
    }
}
extern "C" int sat_wn_grad_splits(const float* partial, int nsplit, long long count, long long so_m, long long so_n, long long so_k,
                                  const float* v, const float* g, const float* norm, float* dv, float* dg, int D0, int N, int K,
                                  const float* bias_partial, int bias_cols, float* dbias, void* stream) {
    if (D0 <= 0 || N <= 0 || K <= 0 || nsplit <= 0 || count < (long long)D0 * N * K || (bias_partial && (bias_cols <= 0 || !dbias))) {
        sat_set_error("sat_wn_grad_splits: bad shape");
        return 1;
    }
    SatWnSplitParams p{partial, v, g, norm, dv, dg, count, so_m, so_n, so_k, nsplit, D0, N, K, bias_partial, dbias, bias_cols};
    SAT_LAUNCH(sat_wn_grad_splits_kernel, dim3(D0), dim3(SAT_WN_THREADS), stream, p);
    return sat_check_launch("sat_wn_grad_splits");
}

extern "C" int sat_wn_fold(const float* v, const float* g, float* w, float* norm, int D0, int R, void* stream) {
    if (D0 <= 0 || R <= 0) { sat_set_error("sat_wn_fold: empty shape"); return 1; }
    SatWnParams p{v, g, nullptr, w, norm, nullptr, D0, R};
    SAT_LAUNCH(sat_wn_fold_kernel, dim3(D0), dim3(256), stream, p);
    return sat_check_launch("sat_wn_fold");
}
extern "C" int sat_wn_grad(const float* v, const float* g, const float* norm, const float* dw, float* dv, float* dg,
                           int D0, int R, void* stream) {
    if (D0 <= 0 || R <= 0) { sat_set_error("sat_wn_grad: empty shape"); return 1; }
    SatWnParams p{v, g, dw, dv, const_cast<float*>(norm), dg, D0, R};
    SAT_LAUNCH(sat_wn_grad_kernel, dim3(D0), dim3(256), stream, p);
    return sat_check_launch("sat_wn_grad");
}

// ---------------------------------------------------------------------------------------------
// weight packing: torch layout w[d0][d1][K]  ->  GEMM-side layouts (see sat_amd.h)
//   mode 0: out[d1][k][d0]                       (conv fwd; convT dgrad)
//   mode 1: out[d0][K-1-k][d1]                   (stride-1 conv dgrad: flipped, transposed)
//   mode 2: out[r][j][d0][d1], k = r + j*S       (polyphase: convT fwd; down-conv dgrad)
// ---------------------------------------------------------------------------------------------
struct SatPackParams {
    const float* w;
    float* out;
    int D0, D1, K, S, mode;
    long long total;
};
__global__ void __launch_bounds__(256) sat_pack_kernel(SatPackParams p) {
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= p.total) return;
    int i0, i1, k;
    if (p.mode == 0) {
        i0 = (int)(o % p.D0);
        const long long q = o / p.D0;
        k = (int)(q % p.K);
        i1 = (int)(q / p.K);
    } else if (p.mode == 1) {
        i1 = (int)(o % p.D1);
        const long long q = o / p.D1;
        k = p.K - 1 - (int)(q % p.K);
        i0 = (int)(q / p.K);
    } else {
        i1 = (int)(o % p.D1);
        long long q = o / p.D1;
        i0 = (int)(q % p.D0);
        q /= p.D0;
        const int j = (int)(q % 2), r = (int)(q / 2);
        k = r + j * p.S;
    }
    p.out[o] = p.w[((size_t)i0 * p.D1 + i1) * p.K + k];
}
extern "C" int sat_pack_weights(const float* w, float* out, int D0, int D1, int K, int S, int mode, void* stream) {
    if (D0 <= 0 || D1 <= 0 || K <= 0) { sat_set_error("sat_pack_weights: empty shape"); return 1; }
    if (mode < 0 || mode > 2) { sat_set_error("sat_pack_weights: bad mode"); return 1; }
    if (mode == 2 && K != 2 * S) { sat_set_error("sat_pack_weights: polyphase packing needs K == 2*S"); return 1; }
    SatPackParams p{w, out, D0, D1, K, S, mode, (long long)D0 * D1 * K};
    SAT_LAUNCH(sat_pack_kernel, dim3((unsigned)sat_cdivll(p.total, 256)), dim3(256), stream, p);
    return sat_check_launch("sat_pack_weights");
}

// ---------------------------------------------------------------------------------------------
// VAE bottleneck (reference: models/bottleneck.py:105-113 vae_sample, :119-133 VAEBottleneck.encode)
//   mean, scale = chunk(pre, 2, dim=1); stdev = softplus(scale) + 1e-4; z = noise*stdev + mean
//   kl = (mean^2 + var - log var - 1).sum(1).mean()
// The N(0,1) draw is an INPUT (torch.randn_like on the caller side) so results are reproducible.
// ---------------------------------------------------------------------------------------------------------------------
// Row packing for the 2-D convs of the MS-STFT discriminator (models/encodec.py:37-106) run as 1-D convs over virtual channels
// (stable_audio_tools_amd/discriminators.py conv2d_virtual): one pass builds the (B, C*kh, T*pitch) sequence tensor — kh frame taps as
// channels (time-shifted copies), rows [pad_w zeros | W samples | zeros] of `pitch` floats — instead of torch pad + stack + pad +
// reshape (and their autograd adds / fills); one pass takes a conv output back to (B, C, T, W) and applies LeakyReLU.
// ---------------------------------------------------------------------------------------------------------------------
struct SatRowsParams {
    const float* src;
    const float* aux;     // unpack backward: the activated output (sign of the pre-activation)
    float* dst;
    int B, C, T, W, kh, dil_t, pad_t, pad_w, pitch, lead;
    float slope;
};

// buf[lead + ((b*C*kh + c*kh + kt)*T + t)*pitch + pad_w + w] = x[b][c][t + kt*dil_t - pad_t][w]; everything else (row padding,
// frames outside [0, T), the `lead` floats before and after) = 0.  One thread = 4 consecutive floats of buf (pitch % 4 == 0).
__global__ void __launch_bounds__(256) sat_rows_pack_kernel(SatRowsParams p) {
    const int q4 = p.pitch >> 2;
    const long long groups = (long long)p.B * p.C * p.kh * p.T * q4;
    const long long lead4 = p.lead >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < groups + 2 * lead4; i += (long long)gridDim.x * 256) {
        if (i >= groups) {                                  // the slack before / after the sequence
            const long long k = i - groups;
            float* d = (k < lead4) ? p.dst + 4 * k : p.dst + p.lead + 4 * groups + 4 * (k - lead4);
            *reinterpret_cast<f32x4*>(d) = f32x4{0.f, 0.f, 0.f, 0.f};
            continue;
        }
        const int j = (int)(i % q4);
        long long r = i / q4;
        const int t = (int)(r % p.T);
        r /= p.T;
        const int kt = (int)(r % p.kh);
        r /= p.kh;                                          // r = b * C + c
        const int ts = t + kt * p.dil_t - p.pad_t;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)ts < (unsigned)p.T) {
            const float* row = p.src + (r * p.T + ts) * p.W;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int w = 4 * j + e - p.pad_w;
                if ((unsigned)w < (unsigned)p.W) v[e] = row[w];
            }
        }
        *reinterpret_cast<f32x4*>(p.dst + p.lead + 4 * i) = v;
    }
}
// adjoint: dx[b][c][t][w] = sum_kt dbuf[...(c*kh + kt), t - kt*dil_t + pad_t, pad_w + w]
__global__ void __launch_bounds__(256) sat_rows_pack_bwd_kernel(SatRowsParams p) {
    const long long total = (long long)p.B * p.C * p.T * p.W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int w = (int)(i % p.W);
        long long r = i / p.W;
        const int t = (int)(r % p.T);
        r /= p.T;                                           // b * C + c
        float s = 0.0f;
        for (int kt = 0; kt < p.kh; ++kt) {
            const int tb = t - kt * p.dil_t + p.pad_t;
            if ((unsigned)tb < (unsigned)p.T) s += p.src[p.lead + ((r * p.kh + kt) * p.T + tb) * p.pitch + p.pad_w + w];
        }
        p.dst[i] = s;
    }
}
// out[b][c][t][w] = leaky_relu(y[b][c][t*pitch + pad_w + w], slope)   (slope 1: plain copy)
__global__ void __launch_bounds__(256) sat_rows_unpack_kernel(SatRowsParams p) {
    const long long total = (long long)p.B * p.C * p.T * p.W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int w = (int)(i % p.W);
        const long long r = i / p.W;                        // (b * C + c) * T + t
        const float v = p.src[r * p.pitch + p.pad_w + w];
        p.dst[i] = v > 0.0f ? v : v * p.slope;
    }
}
// dy[b][c][t*pitch + col] = dout[b][c][t][col - pad_w] * (out > 0 ? 1 : slope) inside the row, 0 on the padding columns
__global__ void __launch_bounds__(256) sat_rows_unpack_bwd_kernel(SatRowsParams p) {
    const int q4 = p.pitch >> 2;
    const long long groups = (long long)p.B * p.C * p.T * q4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < groups; i += (long long)gridDim.x * 256) {
        const int j = (int)(i % q4);
        const long long r = i / q4;                         // (b * C + c) * T + t
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int w = 4 * j + e - p.pad_w;
            if ((unsigned)w < (unsigned)p.W) {
                const float g = p.src[r * p.W + w];
                v[e] = (p.aux == nullptr || p.aux[r * p.W + w] > 0.0f) ? g : g * p.slope;
            }
        }
        *reinterpret_cast<f32x4*>(p.dst + 4 * i) = v;
    }
}

static int sat_rows_check(const char* who, int B, int C, int T, int W, int kh, int pad_w, int pitch, int lead) {
    if (B <= 0 || C <= 0 || T <= 0 || W <= 0 || kh < 1 || pad_w < 0 || (pitch & 3) || pitch < W + pad_w || (lead & 3) || lead < 0) {
        sat_set_error(who);
        return 1;
    }
    return 0;
}
static unsigned sat_rows_grid(long long n) {
    const long long g = sat_cdivll(n, 256);
    return (unsigned)(g < 65536 ? (g > 0 ? g : 1) : 65536);
}
extern "C" int sat_rows_pack(const float* x, float* buf, int B, int C, int T, int W, int kh, int dil_t, int pad_t, int pad_w, int pitch,
                             int lead, void* stream) {
    if (sat_rows_check("sat_rows_pack: bad shape (pitch % 4 == 0, pitch >= W + pad_w, lead % 4 == 0)", B, C, T, W, kh, pad_w, pitch, lead)) return 1;
    SatRowsParams p{x, nullptr, buf, B, C, T, W, kh, dil_t, pad_t, pad_w, pitch, lead, 1.0f};
    SAT_LAUNCH(sat_rows_pack_kernel, dim3(sat_rows_grid((long long)B * C * kh * T * (pitch >> 2) + (lead >> 1))), dim3(256), stream, p);
    return sat_check_launch("sat_rows_pack");
}
extern "C" int sat_rows_pack_bwd(const float* dbuf, float* dx, int B, int C, int T, int W, int kh, int dil_t, int pad_t, int pad_w,
                                 int pitch, int lead, void* stream) {
    if (sat_rows_check("sat_rows_pack_bwd: bad shape", B, C, T, W, kh, pad_w, pitch, lead)) return 1;
    SatRowsParams p{dbuf, nullptr, dx, B, C, T, W, kh, dil_t, pad_t, pad_w, pitch, lead, 1.0f};
    SAT_LAUNCH(sat_rows_pack_bwd_kernel, dim3(sat_rows_grid((long long)B * C * T * W)), dim3(256), stream, p);
    return sat_check_launch("sat_rows_pack_bwd");
}
extern "C" int sat_rows_unpack(const float* y, float* out, int B, int C, int T, int W, int pad_w, int pitch, float slope, void* stream) {
    if (sat_rows_check("sat_rows_unpack: bad shape", B, C, T, W, 1, pad_w, pitch, 0)) return 1;
    SatRowsParams p{y, nullptr, out, B, C, T, W, 1, 1, 0, pad_w, pitch, 0, slope};
    SAT_LAUNCH(sat_rows_unpack_kernel, dim3(sat_rows_grid((long long)B * C * T * W)), dim3(256), stream, p);
    return sat_check_launch("sat_rows_unpack");
}
extern "C" int sat_rows_unpack_bwd(const float* dout, const float* out, float* dy, int B, int C, int T, int W, int pad_w, int pitch,
                                   float slope, void* stream) {
    if (sat_rows_check("sat_rows_unpack_bwd: bad shape", B, C, T, W, 1, pad_w, pitch, 0)) return 1;
    SatRowsParams p{dout, out, dy, B, C, T, W, 1, 1, 0, pad_w, pitch, 0, slope};
    SAT_LAUNCH(sat_rows_unpack_bwd_kernel, dim3(sat_rows_grid((long long)B * C * T * (pitch >> 2))), dim3(256), stream, p);
    return sat_check_launch("sat_rows_unpack_bwd");
}

// ---------------------------------------------------------------------------------------------
struct SatVaeParams {
    const float* pre;    // (B, 2C, T)
    const float* noise;  // (B, C, T)
    const float* dz;     // (B, C, T) (bwd)
    float* z;            // (B, C, T) fwd out | d_pre (B, 2C, T) bwd out
    float* kl_partial;   // [gridDim.x] fwd
    const float* dkl;    // bwd: device scalar dL/dkl (or null)
    float inv_bt;        // bwd: 1 / (B*T)
    int B, C, T;
};
SAT_DEVICE float sat_softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }  // torch threshold = 20

__global__ void __launch_bounds__(256) sat_vae_fwd_kernel(SatVaeParams p) {
    __shared__ float red[4];
    const long long n = (long long)p.B * p.C * p.T;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int t = (int)(i % p.T);
        const long long q = i / p.T;
        const int c = (int)(q % p.C), b = (int)(q / p.C);
        const float mean = p.pre[((size_t)b * 2 * p.C + c) * p.T + t];
        const float scale = p.pre[((size_t)b * 2 * p.C + p.C + c) * p.T + t];
        const float stdev = sat_softplus(scale) + 1e-4f;
        const float var = stdev * stdev;
        p.z[i] = p.noise[i] * stdev + mean;
        s += mean * mean + var - logf(var) - 1.0f;
    }
    s = sat_block_sum_256(s, red);
    if (threadIdx.x == 0) p.kl_partial[blockIdx.x] = s;
}
__global__ void __launch_bounds__(256) sat_vae_bwd_kernel(SatVaeParams p) {
    const long long n = (long long)p.B * p.C * p.T;
    const float gkl = p.dkl ? p.dkl[0] * p.inv_bt : 0.0f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int t = (int)(i % p.T);
        const long long q = i / p.T;
        const int c = (int)(q % p.C), b = (int)(q / p.C);
        const size_t im = ((size_t)b * 2 * p.C + c) * p.T + t;
        const size_t is = ((size_t)b * 2 * p.C + p.C + c) * p.T + t;
        const float mean = p.pre[im], scale = p.pre[is];
        const float stdev = sat_softplus(scale) + 1e-4f;
        const float sig = scale > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-scale));
        const float dz = p.dz ? p.dz[i] : 0.0f;
        p.z[im] = dz + gkl * 2.0f * mean;
        p.z[is] = (dz * p.noise[i] + gkl * (2.0f * stdev - 2.0f / stdev)) * sig;
    }
}
extern "C" int sat_vae_nblocks(long long n) {
    long long b = sat_cdivll(n, 256);
    if (b > 1024) b = 1024;
    return (int)b;
}
extern "C" int sat_vae_sample_fwd(const float* pre, const float* noise, float* z, float* kl_partial, int B, int C,
                                  int T, void* stream) {
    if (B <= 0 || C <= 0 || T <= 0) { sat_set_error("sat_vae_sample_fwd: empty shape"); return 1; }
    SatVaeParams p{pre, noise, nullptr, z, kl_partial, nullptr, 0.f, B, C, T};
    SAT_LAUNCH(sat_vae_fwd_kernel, dim3(sat_vae_nblocks((long long)B * C * T)), dim3(256), stream, p);
    return sat_check_launch("sat_vae_sample_fwd");
}
extern "C" int sat_vae_sample_bwd(const float* pre, const float* noise, const float* dz, const float* dkl,
                                  float* dpre, int B, int C, int T, void* stream) {
    if (B <= 0 || C <= 0 || T <= 0) { sat_set_error("sat_vae_sample_bwd: empty shape"); return 1; }
    SatVaeParams p{pre, noise, dz, dpre, nullptr, dkl, 1.0f / ((float)B * (float)T), B, C, T};
    SAT_LAUNCH(sat_vae_bwd_kernel, dim3(sat_vae_nblocks((long long)B * C * T)), dim3(256), stream, p);
    return sat_check_launch("sat_vae_sample_bwd");
}

// ---------------------------------------------------------------------------------------------
// Fused AdamW over a flat fp32 parameter buffer (torch.optim.AdamW semantics, the optimizer the
// reference configures: configs/model_configs/autoencoders/stable_audio_2_0_vae.json:41-49;
// training/utils.py:60-79).  One launch per optimizer step instead of one per tensor.
// ---------------------------------------------------------------------------------------------
struct SatAdamParams {
    float* p;
    const float* g;
    float* m;
    float* v;
    long long n;
    float lr, b1, b2, eps, wd, bc1, bc2s;  // bc1 = 1 - b1^t, bc2s = sqrt(1 - b2^t)
    float gscale;
    float* ema;       // optional EMA shadow of the parameters (null = none)
    float ema_decay;  // ema = decay*ema + (1-decay)*p_before_step  (the wrapper calls ema.update()
                      // BEFORE optimizer.step(): training/autoencoders.py:504-515)
};
SAT_DEVICE void sat_adamw_one(const SatAdamParams& a, float g, float& p, float& m, float& v, float* ema) {
    g *= a.gscale;
    if (ema) *ema = a.ema_decay * *ema + (1.0f - a.ema_decay) * p;
    p *= (1.0f - a.lr * a.wd);
    m = a.b1 * m + (1.0f - a.b1) * g;
    v = a.b2 * v + (1.0f - a.b2) * g * g;
    const float denom = sqrtf(v) / a.bc2s + a.eps;
    p -= (a.lr / a.bc1) * (m / denom);
}
// streaming update, 16-byte accesses (the flat buffers are 16-byte aligned; the < 4 leftover elements go one by one)
__global__ void __launch_bounds__(256) sat_adamw_kernel(SatAdamParams a) {
    const long long n4 = a.n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 g = reinterpret_cast<const f32x4*>(a.g)[i];
        f32x4 p = reinterpret_cast<f32x4*>(a.p)[i], m = reinterpret_cast<f32x4*>(a.m)[i], v = reinterpret_cast<f32x4*>(a.v)[i];
        f32x4 e = {0.f, 0.f, 0.f, 0.f};
        if (a.ema) e = reinterpret_cast<f32x4*>(a.ema)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pj = p[j], mj = m[j], vj = v[j], ej = e[j];
            sat_adamw_one(a, g[j], pj, mj, vj, a.ema ? &ej : nullptr);
            p[j] = pj; m[j] = mj; v[j] = vj; e[j] = ej;
        }
        reinterpret_cast<f32x4*>(a.p)[i] = p;
        reinterpret_cast<f32x4*>(a.m)[i] = m;
        reinterpret_cast<f32x4*>(a.v)[i] = v;
        if (a.ema) reinterpret_cast<f32x4*>(a.ema)[i] = e;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        float p = a.p[i], m = a.m[i], v = a.v[i];
        sat_adamw_one(a, a.g[i], p, m, v, a.ema ? a.ema + i : nullptr);
        a.p[i] = p; a.m[i] = m; a.v[i] = v;
    }
}
// The same update with the per-step scalars read from DEVICE memory — hyper[5] = {lr, 1 - beta1^t, sqrt(1 - beta2^t), grad_scale,
// ema_decay} — so that a training step captured into a HIP graph (training.GraphedTrainStep) is replayed with the next step's learning
// rate, bias corrections and EMA decay by rewriting 20 bytes.
struct SatAdamDevParams { SatAdamParams a; const float* hyper; };
__global__ void __launch_bounds__(256) sat_adamw_dev_kernel(SatAdamDevParams d) {
    SatAdamParams a = d.a;
    a.lr = d.hyper[0]; a.bc1 = d.hyper[1]; a.bc2s = d.hyper[2]; a.gscale = d.hyper[3]; a.ema_decay = d.hyper[4];
    const long long n4 = a.n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 g = reinterpret_cast<const f32x4*>(a.g)[i];
        f32x4 p = reinterpret_cast<f32x4*>(a.p)[i], m = reinterpret_cast<f32x4*>(a.m)[i], v = reinterpret_cast<f32x4*>(a.v)[i];
        f32x4 e = {0.f, 0.f, 0.f, 0.f};
        if (a.ema) e = reinterpret_cast<f32x4*>(a.ema)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pj = p[j], mj = m[j], vj = v[j], ej = e[j];
            sat_adamw_one(a, g[j], pj, mj, vj, a.ema ? &ej : nullptr);
            p[j] = pj; m[j] = mj; v[j] = vj; e[j] = ej;
        }
        reinterpret_cast<f32x4*>(a.p)[i] = p;
        reinterpret_cast<f32x4*>(a.m)[i] = m;
        reinterpret_cast<f32x4*>(a.v)[i] = v;
        if (a.ema) reinterpret_cast<f32x4*>(a.ema)[i] = e;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        float p = a.p[i], m = a.m[i], v = a.v[i];
        sat_adamw_one(a, a.g[i], p, m, v, a.ema ? a.ema + i : nullptr);
        a.p[i] = p; a.m[i] = m; a.v[i] = v;
    }
}
extern "C" int sat_adamw_step_dev(float* p, const float* g, float* m, float* v, long long n, const float* hyper, float beta1,
                                  float beta2, float eps, float weight_decay, float* ema, void* stream) {
    if (n <= 0 || !hyper) { sat_set_error("sat_adamw_step_dev: empty buffer or no hyper-parameter buffer"); return 1; }
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema) & 15) != 0) {
        sat_set_error("sat_adamw_step_dev: buffers must be 16-byte aligned");
        return 1;
    }
    SatAdamDevParams d{{p, g, m, v, n, 0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f, 1.f, ema, 0.f}, hyper};
    long long nb = sat_cdivll(sat_cdivll(n, 4), 256);
    if (nb > 4096) nb = 4096;
    SAT_LAUNCH(sat_adamw_dev_kernel, dim3((unsigned)nb), dim3(256), stream, d);
    return sat_check_launch("sat_adamw_step_dev");
}
extern "C" int sat_adamw_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float grad_scale, float* ema,
                              float ema_decay, void* stream) {
    if (n <= 0 || step < 1) { sat_set_error("sat_adamw_step: empty buffer or step < 1"); return 1; }
    SatAdamParams a{p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
                    1.0f - powf(beta1, (float)step), sqrtf(1.0f - powf(beta2, (float)step)), grad_scale, ema, ema_decay};
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema) & 15) != 0) {
        sat_set_error("sat_adamw_step: buffers must be 16-byte aligned");
        return 1;
    }
    long long nb = sat_cdivll(sat_cdivll(n, 4), 256);
    if (nb > 4096) nb = 4096;
    SAT_LAUNCH(sat_adamw_kernel, dim3((unsigned)nb), dim3(256), stream, a);
    return sat_check_launch("sat_adamw_step");
}

// ---------------------------------------------------------------------------------------------------------------------
// sat_multi_copy — many small device-to-device copies in a few launches: the gradients autograd produced for the ~380 parameters of
// the Oobleck VAE (1 000+ of the DiT) gathered into the flat gradient buffer (training.FlatParameters.gather_grads).  Until round 5
// every parameter's gradient cost its own 5-us `add` launch of torch's AccumulateGrad (379 launches, 2.0 ms of the 148-ms generator step).
// The table travels IN THE KERNEL ARGUMENTS (up to SAT_MC_ARGS entries of 24 bytes per launch: under the 4-KiB argument limit), not
// through a device buffer: no pinned staging, no host-to-device copy in front of the launch, and a HIP-graph capture simply records
// the entries with the launch (the gradients of a captured step live at fixed addresses of the graph's pool).  Entries: {src, dst,
// numel, first block}, `first block` = prefix sum of ceil(numel / 16384) — a block finds its entry by binary search (uniform: scalar
// loads from the argument segment).  fp32 only; 16-byte accesses when both pointers of an entry allow them.
// ---------------------------------------------------------------------------------------------------------------------
struct SatCopyEntry {          // host table of sat_multi_copy (C-ABI)
    const float* src;
    float* dst;
    long long n;
};
struct SatMcArg {
    const float* src;
    float* dst;
    unsigned n;
    unsigned block0;
};
#define SAT_MC_ARGS 160
struct SatMultiCopyParams {
    SatMcArg e[SAT_MC_ARGS];
    int nent;
};
static_assert(sizeof(SatMultiCopyParams) <= 4096, "kernel arguments are limited to 4 KiB");
#define SAT_MC_CHUNK 16384
__global__ void __launch_bounds__(256) sat_multi_copy_kernel(SatMultiCopyParams p) {
    const unsigned b = blockIdx.x;
    int lo = 0, hi = p.nent - 1;
    while (lo < hi) {                                         // largest e with block0[e] <= b
        const int mid = (lo + hi + 1) >> 1;
        if (p.e[mid].block0 <= b) lo = mid;
        else hi = mid - 1;
    }
    const float* src = p.e[lo].src;
    float* dst = p.e[lo].dst;
    const long long n = p.e[lo].n;
    const long long beg = (long long)(b - p.e[lo].block0) * SAT_MC_CHUNK;
    long long end = beg + SAT_MC_CHUNK;
    if (end > n) end = n;
    const float* s = src + beg;
    float* d = dst + beg;
    const long long cnt = end - beg;
    if (((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0) {
        const long long n4 = cnt >> 2;
        for (long long i = threadIdx.x; i < n4; i += 256) reinterpret_cast<f32x4*>(d)[i] = reinterpret_cast<const f32x4*>(s)[i];
        for (long long i = (n4 << 2) + threadIdx.x; i < cnt; i += 256) d[i] = s[i];
    } else {
        for (long long i = threadIdx.x; i < cnt; i += 256) d[i] = s[i];
    }
}
extern "C" int sat_multi_copy(const void* entries, int nent, void* stream) {
    if (nent <= 0) return 0;
    if (!entries) { sat_set_error("sat_multi_copy: no table"); return 1; }
    const SatCopyEntry* tab = (const SatCopyEntry*)entries;
    for (int base = 0; base < nent; base += SAT_MC_ARGS) {
        SatMultiCopyParams p;
        const int cnt = nent - base < SAT_MC_ARGS ? nent - base : SAT_MC_ARGS;
        unsigned blocks = 0;
        int used = 0;
        for (int i = 0; i < cnt; ++i) {
            const SatCopyEntry& e = tab[base + i];
            if (e.n <= 0) continue;
            if (!e.src || !e.dst || e.n > 0x7fffffffLL) { sat_set_error("sat_multi_copy: bad entry"); return 1; }
            p.e[used] = SatMcArg{e.src, e.dst, (unsigned)e.n, blocks};
            blocks += (unsigned)((e.n + SAT_MC_CHUNK - 1) / SAT_MC_CHUNK);
            ++used;
        }
        if (!used) continue;
        for (int i = used; i < SAT_MC_ARGS; ++i) p.e[i] = SatMcArg{nullptr, nullptr, 0u, 0u};
        p.nent = used;
        SAT_LAUNCH(sat_multi_copy_kernel, dim3(blocks), dim3(256), stream, p);
        if (int rc = sat_check_launch("sat_multi_copy")) return rc;
    }
    return 0;
}
