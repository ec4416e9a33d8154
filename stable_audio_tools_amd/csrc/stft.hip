// stft.hip — auraloss multi-resolution STFT loss, forward and backward, without ever materialising
// a spectrogram (reference: stable_audio_tools/training/losses/auraloss.py — FIRFilter :76-169,
// STFTLoss.stft :368-395, SpectralConvergenceLoss :171-181, STFTMagnitudeLoss :183-223,
// STFTLoss.forward :397-449; stereo assembly training/autoencoders.py:142-146,:186-194).
//
//   sat_fir         A-weighting FIR (101-tap cross-correlation, zero pad) and its adjoint.
//   sat_stft_fwd    per resolution: frames (reflect pad, periodic Hann) -> LDS radix-2 FFT ->
//                   |X|,|Y| -> per-(item, view) sums  S1 = sum (|Y|-|X|)^2, S2 = sum |Y|^2,
//                   S3 = sum |log|X| - log|Y||.
//   sat_stft_bwd    recomputes the FFT, forms dL/dY per bin from the three sums' coefficients,
//                   runs the adjoint DFT in LDS, applies the window, overlap-adds a block's frames in
//                   LDS (gathered per output sample) and writes dL/d(filtered y) as four planes (even /
//                   odd workgroups, direct / reflected samples) with plain stores: no atomics anywhere,
//                   the gradient is bit-reproducible.
//
// Two real signals ride in one complex FFT (z = x + i y), so one transform yields both spectra.
// A "view" is a linear combination of an item's channels (sum / difference / left / right of a
// stereo pair — SumAndDifference, auraloss.py:43-73); the FIR is linear, so it is applied once per
// channel and the views are formed while frames are loaded.
#include "sat_device.h"

#define SAT_FFT_MAX 2048
// tuning knobs (defaults = the shipped configuration; tools/stft_sweep.sh builds variants of this file alone with -D...)
#ifndef SAT_STFT_NG
#define SAT_STFT_NG 4  // frame groups per workgroup (consecutive frames share the LDS overlap-add buffer); 8 until the sweep of round 6 (profiles/r06_experiments/stft_sweep/: 2.45 -> 2.25 ms over the seven resolutions)
#endif
#ifndef SAT_STFT_FBPTS
#define SAT_STFT_FBPTS 512  // points per channel transformed concurrently for n below this (fb = FBPTS / n frames)
#endif
#ifndef SAT_STFT_CCMAX
#define SAT_STFT_CCMAX 1    // channels transformed concurrently by the n = 2048 instance
#endif
#define SAT_STFT_OBUF (7 * 512 + 2048)
#define SAT_STFT_SMALL 512             // n <= 512: fb * n == 512
#define SAT_STFT_OBUF_SMALL 1408       // (8 fb - 1) hop + n at hop = n / 4 for n <= 512 (largest at n = 512)
#define SAT_STFT_MID 1024              // n == 1024
#define SAT_STFT_OBUF_MID 2816         // 7 * 256 + 1024

#if defined(SAT_HIPEMU)
static inline unsigned sat_brev(unsigned v) {
    unsigned r = 0;
    for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i);
    return r;
}
static inline void sat_sincospi(float x, float* s, float* c) {
    *s = (float)sin(3.14159265358979323846 * (double)x);
    *c = (float)cos(3.14159265358979323846 * (double)x);
}
#else
SAT_DEVICE unsigned sat_brev(unsigned v) { return __brev(v); }
SAT_DEVICE void sat_sincospi(float x, float* s, float* c) { sincospif(x, s, c); }
#endif

// ---------------------------------------------------------------------------------------------
// FIR
// ---------------------------------------------------------------------------------------------
struct SatFirParams {
    const float* x;     // (N, T)
    const float* taps;  // (ntaps)
    float* y;           // (N, T)
    int N, T, ntaps, adjoint;
};
#define SAT_FIR_TILE 1024
#define SAT_FIR_MAXTAPS 257
__global__ void __launch_bounds__(256) sat_fir_kernel(SatFirParams p) {
    __shared__ float xs[SAT_FIR_TILE + SAT_FIR_MAXTAPS];
    __shared__ float ts[SAT_FIR_MAXTAPS];
    const int row = blockIdx.y;
    const int t0 = blockIdx.x * SAT_FIR_TILE;
    const int P = p.ntaps / 2;
    const float* xr = p.x + (size_t)row * p.T;
    for (int i = threadIdx.x; i < p.ntaps; i += 256) ts[i] = p.adjoint ? p.taps[p.ntaps - 1 - i] : p.taps[i];
    for (int i = threadIdx.x; i < SAT_FIR_TILE + p.ntaps - 1; i += 256) {
        const int t = t0 - P + i;
        xs[i] = (t >= 0 && t < p.T) ? xr[t] : 0.0f;
    }
    __syncthreads();
    for (int u = 0; u < SAT_FIR_TILE / 256; ++u) {
        const int o = threadIdx.x + u * 256;
        const int t = t0 + o;
        if (t < p.T) {
            float acc = 0.f;
            for (int k = 0; k < p.ntaps; ++k) acc = fmaf(ts[k], xs[o + k], acc);
            p.y[(size_t)row * p.T + t] = acc;
        }
    }
}
extern "C" int sat_fir(const float* x, const float* taps, float* y, int N, int T, int ntaps, int adjoint, void* stream) {
    if (N <= 0 || T <= 0) { sat_set_error("sat_fir: empty shape"); return 1; }
    if (ntaps < 1 || ntaps > SAT_FIR_MAXTAPS || (ntaps & 1) == 0) { sat_set_error("sat_fir: ntaps must be odd and <= 257"); return 1; }
    SatFirParams p{x, taps, y, N, T, ntaps, adjoint};
    SAT_LAUNCH(sat_fir_kernel, dim3(sat_cdiv(T, SAT_FIR_TILE), N), dim3(256), stream, p);
    return sat_check_launch("sat_fir");
}

// ---------------------------------------------------------------------------------------------
// STFT loss
// ---------------------------------------------------------------------------------------------
struct SatStftParams {
    const float* x;       // (NI, C, T) first loss argument  ("input" in auraloss naming)
    const float* y;       // (NI, C, T) second argument ("target"); gradients flow to this one
    const float* views;   // (NV, 2) channel weights
    float* partial;       // fwd: [NI][NV][3][tiles]
    const float* coef;    // bwd: [NI][NV][3]  (c1, c2, c3)
    float* dy;            // bwd: (4, NI, C, T) planes [direct-even | direct-odd | mirror-even | mirror-odd], zero-filled by the caller; plain stores
    int NI, C, T, NV;
    int n, log2n, hop, nframes;
    int fb;               // frames transformed concurrently
    int wrt_x;            // bwd: 0 = gradient w.r.t. y (second argument), 1 = w.r.t. x (first argument)
};

SAT_DEVICE int sat_reflect(int t, int T) {
    if (t < 0) t = -t;
    if (t >= T) t = 2 * (T - 1) - t;
    return t;
}

struct SatFftLds {
    float* re;
    float* im;
    float* twr;
    float* twi;
};

// twiddles tw[j] = exp(-2 pi i j / n), j < n/2
SAT_DEVICE void sat_fft_init_twiddles(const SatFftLds& L, int n) {
    for (int j = threadIdx.x; j < n / 2; j += 256) {
        float s, c;
        sat_sincospi(-2.0f * (float)j / (float)n, &s, &c);
        L.twr[j] = c;
        L.twi[j] = s;
    }
}
SAT_DEVICE float sat_hann(const SatFftLds& L, int j, int n) {
    const int h = n >> 1;
    const float c = (j < h) ? L.twr[j] : -L.twr[j - h];
    return 0.5f - 0.5f * c;
}
// K fused radix-2 stages (s .. s+K-1, half = 2^(s-1)) on 2^K points held in registers: the butterflies, their order and their
// twiddles are exactly those of K single radix-2 DIT stages (bit-identical results), with one LDS round trip and one barrier instead of K.
template <int K>
SAT_DEVICE void sat_fft_pass(const SatFftLds& L, int n, int log2n, int fb, int s) {
    constexpr int R = 1 << K;
    const int half = 1 << (s - 1);
    const int per_frame = n >> K;                         // items (groups of R points) per frame
    const int lpf = log2n - K;
    for (int i = threadIdx.x; i < fb * per_frame; i += 256) {
        const int fi = i >> lpf;
        const int bi = i & (per_frame - 1);
        const int grp = bi >> (s - 1), pos = bi & (half - 1);
        const int base = fi * n + (grp << (s - 1 + K)) + pos;
        float xr[R], xi[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            xr[j] = L.re[base + j * half];
            xi[j] = L.im[base + j * half];
        }
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const int d = 1 << q;                          // butterfly distance in units of `half`
#pragma unroll
            for (int m = 0; m < d; ++m) {                  // position inside the stage's block: pos + m * half
                const int tw = (pos + m * half) << (log2n - s - q);
                const float wr = L.twr[tw], wi = L.twi[tw];
#pragma unroll
                for (int j0 = 0; j0 < R; j0 += 2 * d) {
                    const int a = j0 + m, b = a + d;
                    const float tr = xr[b] * wr - xi[b] * wi, ti = xr[b] * wi + xi[b] * wr;
                    const float ar = xr[a], ai = xi[a];
                    xr[a] = ar + tr;
                    xi[a] = ai + ti;
                    xr[b] = ar - tr;
                    xi[b] = ai - ti;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            L.re[base + j * half] = xr[j];
            L.im[base + j * half] = xi[j];
        }
    }
    __syncthreads();
}
// in-place radix-2 DIT over `fb` frames of length n stored bit-reversed, run as fused radix-8 / radix-4 passes; ends with a barrier
SAT_DEVICE void sat_fft_run(const SatFftLds& L, int n, int log2n, int fb) {
    int s = 1;
    while (log2n - s + 1 >= 3) {
        sat_fft_pass<3>(L, n, log2n, fb, s);
        s += 3;
    }
    if (log2n - s + 1 == 2) sat_fft_pass<2>(L, n, log2n, fb, s);
    else if (log2n - s + 1 == 1) sat_fft_pass<1>(L, n, log2n, fb, s);
}

// frames f0 .. f0+fb-1 of ONE channel into the frame slots slot0 .. slot0+fb-1: x (first loss argument) in the real part,
// y in the imaginary part, windowed, stored bit-reversed for the in-place DIT
SAT_DEVICE void sat_stft_load_channel(const SatStftParams& p, const SatFftLds& L, int item, int ch, int f0, int slot0) {
    const int n = p.n, log2n = p.log2n;
    const float* x0 = p.x + ((size_t)item * p.C + ch) * p.T;
    const float* y0 = p.y + ((size_t)item * p.C + ch) * p.T;
    for (int i = threadIdx.x; i < p.fb * n; i += 256) {
        const int fi = i >> log2n, j = i & (n - 1);
        const int f = f0 + fi;
        float xv = 0.f, yv = 0.f;
        if (f < p.nframes) {
            const int t = sat_reflect(f * p.hop + j - (n >> 1), p.T);
            const float w = sat_hann(L, j, n);
            xv = w * x0[t];
            yv = w * y0[t];
        }
        const int jr = (int)(sat_brev((unsigned)j) >> (32 - log2n));
        L.re[(slot0 + fi) * n + jr] = xv;
        L.im[(slot0 + fi) * n + jr] = yv;
    }
}

// spectra of the two packed real signals at bin k of frame slot fi
SAT_DEVICE void sat_unpack_bins(const SatFftLds& L, int n, int fi, int k, float* xr, float* xi, float* yr, float* yi) {
    const int k2 = (n - k) & (n - 1);
    const float ar = L.re[fi * n + k], ai = L.im[fi * n + k];
    const float br = L.re[fi * n + k2], bi = L.im[fi * n + k2];
    *xr = 0.5f * (ar + br);
    *xi = 0.5f * (ai - bi);
    *yr = 0.5f * (ai + bi);
    *yi = -0.5f * (ar - br);
}

// The DFT is linear and a view is a linear combination of an item's channels, so the kernels transform each CHANNEL once
// (x and y of a channel packed in one complex FFT) and form every view's spectrum from the channels' bins in registers:
// two transforms per frame group instead of one per view (round 6; four views = the sum / difference / left / right of
// training/autoencoders.py:186-194).  Template parameters: NMAX = the largest fb * n an instance holds (2048, 1024, or 512 for
// the five resolutions n <= 512: a fraction of the LDS, so that several workgroups share a CU — these kernels are chains of
// dependent LDS round trips); U = bins a thread owns (fb * (n/2+1) <= 256 U); CC = channels transformed concurrently (2: both
// channels of a stereo item go through the FFT passes side by side — twice the butterflies per barrier).
//
// Loads the channels' frames, transforms them and leaves each thread with the bins it owns: sx / sy [channel][u][re, im].
// Ends with a barrier: the frame buffer may be overwritten afterwards.
template <int U, int CC>
SAT_DEVICE void sat_stft_channel_spectra(const SatStftParams& p, const SatFftLds& L, int item, int f0, float (&sx)[2][U][2], float (&sy)[2][U][2]) {
    const int n = p.n, nb = (n >> 1) + 1;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int u = 0; u < U; ++u) sx[c][u][0] = sx[c][u][1] = sy[c][u][0] = sy[c][u][1] = 0.f;
#pragma unroll
    for (int c0 = 0; c0 < 2; c0 += CC) {
        if (c0 < p.C) {  // block-uniform
            const int nc = (p.C - c0 < CC) ? p.C - c0 : CC;
#pragma unroll
            for (int cc = 0; cc < CC; ++cc)
                if (cc < nc) sat_stft_load_channel(p, L, item, c0 + cc, f0, cc * p.fb);
            __syncthreads();
            sat_fft_run(L, n, p.log2n, nc * p.fb);
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) {
                if (cc < nc) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = threadIdx.x + u * 256;
                        if (i < p.fb * nb) {
                            const int fi = i / nb, k = i - fi * nb;
                            sat_unpack_bins(L, n, cc * p.fb + fi, k, &sx[c0 + cc][u][0], &sx[c0 + cc][u][1], &sy[c0 + cc][u][0], &sy[c0 + cc][u][1]);
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
}

#define SAT_STFT_VB 4  // views per forward workgroup (grid z walks chunks of views)
template <int NMAX, int U, int CC>
__global__ void __launch_bounds__(256) sat_stft_fwd_kernel(SatStftParams p) {
    __shared__ float re[NMAX * CC], im[NMAX * CC], twr[NMAX / 2], twi[NMAX / 2];
    __shared__ float red[SAT_STFT_VB][3][4];
    const SatFftLds L{re, im, twr, twi};
    const int n = p.n, nb = (n >> 1) + 1;
    const int item = blockIdx.y, v0 = blockIdx.z * SAT_STFT_VB;
    const int nv = (p.NV - v0 < SAT_STFT_VB) ? p.NV - v0 : SAT_STFT_VB;
    sat_fft_init_twiddles(L, n);
    float va[SAT_STFT_VB], vb[SAT_STFT_VB], s[SAT_STFT_VB][3];
#pragma unroll
    for (int v = 0; v < SAT_STFT_VB; ++v) {
        va[v] = (v < nv) ? p.views[(v0 + v) * 2 + 0] : 0.f;
        vb[v] = (v < nv && p.C > 1) ? p.views[(v0 + v) * 2 + 1] : 0.f;
        s[v][0] = s[v][1] = s[v][2] = 0.f;
    }
    __syncthreads();
    for (int g = 0; g < SAT_STFT_NG; ++g) {
        const int f0 = (blockIdx.x * SAT_STFT_NG + g) * p.fb;
        if (f0 >= p.nframes) break;  // block-uniform
        float sx[2][U][2], sy[2][U][2];
        sat_stft_channel_spectra<U, CC>(p, L, item, f0, sx, sy);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = threadIdx.x + u * 256;
            if (i < p.fb * nb && f0 + i / nb < p.nframes) {
#pragma unroll
                for (int v = 0; v < SAT_STFT_VB; ++v) {
                    if (v < nv) {
                        const float xr = va[v] * sx[0][u][0] + vb[v] * sx[1][u][0], xi = va[v] * sx[0][u][1] + vb[v] * sx[1][u][1];
                        const float yr = va[v] * sy[0][u][0] + vb[v] * sy[1][u][0], yi = va[v] * sy[0][u][1] + vb[v] * sy[1][u][1];
                        const float xm = sqrtf(fmaxf(xr * xr + xi * xi, 1e-8f));
                        const float ym = sqrtf(fmaxf(yr * yr + yi * yi, 1e-8f));
                        const float d = ym - xm;
                        s[v][0] += d * d;
                        s[v][1] += ym * ym;
                        s[v][2] += fabsf(logf(xm) - logf(ym));
                    }
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < SAT_STFT_VB; ++v)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float t = sat_wave_sum(s[v][q]);
            if ((threadIdx.x & 63) == 0) red[v][q][threadIdx.x >> 6] = t;
        }
    __syncthreads();
    if ((int)threadIdx.x < 3 * nv) {
        // layout [(item, view, q)][tile]: reduced over tiles by sat_rowsum
        const int v = threadIdx.x / 3, q = threadIdx.x - 3 * v;
        float* o = p.partial + ((((size_t)item * p.NV + v0 + v) * 3 + q) * gridDim.x + blockIdx.x);
        *o = red[v][q][0] + red[v][q][1] + red[v][q][2] + red[v][q][3];
    }
}

// One workgroup handles its SAT_STFT_NG frame groups for ALL views: the views' bin gradients are accumulated per CHANNEL in the
// frequency domain (dL/dY_c = sum_v w_vc dL/dY_v), the two channels' half-spectra are extended to Hermitian ones and packed as
// Z = H_a + i H_b, and ONE complex FFT per frame group returns both channels' time-domain gradients (real / imaginary part),
// which are overlap-added into two per-channel LDS buffers.
template <int NMAX, int OBUF, int U, int CC>
__global__ void __launch_bounds__(256) sat_stft_bwd_kernel(SatStftParams p) {
    __shared__ float re[NMAX * CC], im[NMAX * CC], twr[NMAX / 2], twi[NMAX / 2];
    __shared__ float obuf_a[OBUF], obuf_b[OBUF];
    const SatFftLds L{re, im, twr, twi};
    const int n = p.n, nb = (n >> 1) + 1, log2n = p.log2n;
    const int item = blockIdx.y;
    sat_fft_init_twiddles(L, n);
    const int fpb = SAT_STFT_NG * p.fb;
    const int olen = (fpb - 1) * p.hop + n;
    for (int i = threadIdx.x; i < olen; i += 256) {
        obuf_a[i] = 0.f;
        obuf_b[i] = 0.f;
    }
    __syncthreads();
    const int fbase = blockIdx.x * fpb;
    for (int g = 0; g < SAT_STFT_NG; ++g) {
        const int f0 = fbase + g * p.fb;
        if (f0 >= p.nframes) break;
        float sx[2][U][2], sy[2][U][2];
        sat_stft_channel_spectra<U, CC>(p, L, item, f0, sx, sy);
        // dL/dY_a, dL/dY_b per owned bin (G = gr + i gi: the derivatives w.r.t. the real and the imaginary part)
        float ga[U][2], gb[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) ga[u][0] = ga[u][1] = gb[u][0] = gb[u][1] = 0.f;
        for (int view = 0; view < p.NV; ++view) {
            const float c1 = p.coef[(item * p.NV + view) * 3 + 0];
            const float c2 = p.coef[(item * p.NV + view) * 3 + 1];
            const float c3 = p.coef[(item * p.NV + view) * 3 + 2];
            const float va = p.views[view * 2 + 0], vb = (p.C > 1) ? p.views[view * 2 + 1] : 0.0f;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = threadIdx.x + u * 256;
                if (i < p.fb * nb && f0 + i / nb < p.nframes) {
                    const float xr = va * sx[0][u][0] + vb * sx[1][u][0], xi = va * sx[0][u][1] + vb * sx[1][u][1];
                    const float yr = va * sy[0][u][0] + vb * sy[1][u][0], yi = va * sy[0][u][1] + vb * sy[1][u][1];
                    const float px = xr * xr + xi * xi, py = yr * yr + yi * yi;
                    const float xm = sqrtf(fmaxf(px, 1e-8f)), ym = sqrtf(fmaxf(py, 1e-8f));
                    const float dl = logf(ym) - logf(xm);
                    const float sg = (dl > 0.f) ? 1.f : ((dl < 0.f) ? -1.f : 0.f);
                    float gr = 0.f, gi = 0.f;
                    // clamp(min=eps) passes no gradient below eps (auraloss.py:385-387)
                    if (!p.wrt_x) {
                        if (py > 1e-8f) {
                            const float gm = c1 * ((ym - xm) - c2 * ym) + c3 * sg / ym;
                            gr = gm * yr / ym;
                            gi = gm * yi / ym;
                        }
                    } else if (px > 1e-8f) {
                        const float gm = -c1 * (ym - xm) - c3 * sg / xm;
                        gr = gm * xr / xm;
                        gi = gm * xi / xm;
                    }
                    ga[u][0] += va * gr;
                    ga[u][1] += va * gi;
                    gb[u][0] += vb * gr;
                    gb[u][1] += vb * gi;
                }
            }
        }
        // adjoint DFT of both channels in one transform: a_j = Re(sum_{k <= n/2} Ga_k e^{+2 pi i jk/n}) = sum_{k < n} Ha_k e^{+2 pi i jk/n}
        // with Ha_k = Ga_k / 2, Ha_{n-k} = conj(Ga_k) / 2 (0 < k < n/2), Ha_0 = Re Ga_0, Ha_{n/2} = Re Ga_{n/2}; Z = Ha + i Hb;
        // sum_k Z_k e^{+...jk} = forward DFT of W_m = Z_{(n-m) mod n}: real part a_j, imaginary part b_j.  Every position m is written
        // by the owner of bin k = m or n - m (zeros for frames past the end), so the buffer needs no clearing.
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = threadIdx.x + u * 256;
            if (i < p.fb * nb) {
                const int fi = i / nb, k = i - fi * nb;
                if (k == 0 || k == (n >> 1)) {
                    const int kr = (int)(sat_brev((unsigned)k) >> (32 - log2n));
                    re[fi * n + kr] = ga[u][0];
                    im[fi * n + kr] = gb[u][0];
                } else {
                    const int kr = (int)(sat_brev((unsigned)k) >> (32 - log2n));
                    const int mr = (int)(sat_brev((unsigned)(n - k)) >> (32 - log2n));
                    re[fi * n + mr] = 0.5f * (ga[u][0] - gb[u][1]);   // W_{n-k} = Z_k
                    im[fi * n + mr] = 0.5f * (ga[u][1] + gb[u][0]);
                    re[fi * n + kr] = 0.5f * (ga[u][0] + gb[u][1]);   // W_k = Z_{n-k}
                    im[fi * n + kr] = 0.5f * (gb[u][0] - ga[u][1]);
                }
            }
        }
        __syncthreads();
        sat_fft_run(L, n, log2n, p.fb);
        // overlap-add of this group's fb frames into the per-channel LDS buffers, GATHERED: a thread owns output sample o and
        // sums the (<= n/hop) frames that cover it in frame order — no LDS atomics, so the result does not depend on wave timing
        {
            const int gspan = (p.fb - 1) * p.hop + n;
            const int obase = g * p.fb * p.hop;
            for (int orel = threadIdx.x; orel < gspan; orel += 256) {
                int f_lo = (orel - n + p.hop) / p.hop;            // first frame with f*hop + n > orel  (ceil((orel - n + 1) / hop))
                if (orel - n + 1 <= 0) f_lo = 0;
                int f_hi = orel / p.hop;
                if (f_hi > p.fb - 1) f_hi = p.fb - 1;
                float v = 0.f, w = 0.f;
                for (int fi = f_lo; fi <= f_hi; ++fi) {
                    const int j = orel - fi * p.hop;
                    if (f0 + fi < p.nframes) {
                        const float h = sat_hann(L, j, n);
                        v += re[fi * n + j] * h;
                        w += im[fi * n + j] * h;
                    }
                }
                obuf_a[obase + orel] += v;
                if (p.C > 1) obuf_b[obase + orel] += w;
            }
        }
        __syncthreads();
    }
    // write-out without atomics: obuf[i] belongs to padded-signal index fbase*hop + i -> sample (.. - n/2), reflected at the ends.
    // Consecutive workgroups overlap by n - hop < their own stride, so a sample is touched by at most two NEIGHBOURING workgroups:
    // even and odd workgroups write different planes; reflected ("mirror") samples get planes of their own (the reflection is
    // one-to-one).  dy = 4 planes [direct-even | direct-odd | mirror-even | mirror-odd] of (NI, C, T), zero-filled by the caller,
    // written with plain stores; the gradient is their sum (formed by the caller in a fixed order): bit-reproducible.
    const size_t plane = (size_t)p.NI * p.C * p.T;
    const int last = (p.nframes - 1) * p.hop + n;  // one past the last padded index any frame touches
    for (int i = threadIdx.x; i < olen; i += 256) {
        const int pidx = fbase * p.hop + i;
        if (pidx < last) {
            const int traw = pidx - (n >> 1);
            const int t = sat_reflect(traw, p.T);
            const bool mirror = (traw < 0) || (traw >= p.T);
            float* d0 = p.dy + (size_t)((mirror ? 2 : 0) + (blockIdx.x & 1)) * plane + (size_t)item * p.C * p.T;
            d0[t] = obuf_a[i];
            if (p.C > 1) d0[p.T + t] = obuf_b[i];
        }
    }
}

static int sat_stft_plan(int n, int hop, int T, SatStftParams* p) {
    int l = 0;
    while ((1 << l) < n) ++l;
    if ((1 << l) != n || n < 8 || n > SAT_FFT_MAX) return 1;
    if (hop < 1 || hop > n) return 1;
    if (T <= n / 2) return 1;  // reflect padding needs n/2 < T (torch.stft raises as well)
    p->n = n;
    p->log2n = l;
    p->hop = hop;
    p->nframes = 1 + T / hop;
    p->fb = (n >= SAT_STFT_FBPTS) ? 1 : SAT_STFT_FBPTS / n;
    const int fpb = SAT_STFT_NG * p->fb;
    if ((fpb - 1) * hop + n > SAT_STFT_OBUF) return 1;
    if (p->fb * (n / 2 + 1) > 5 * 256) return 1;
    return 0;
}

extern "C" int sat_stft_tiles(int n, int hop, int T) {
    SatStftParams p;
    if (sat_stft_plan(n, hop, T, &p)) return -1;
    return sat_cdiv(p.nframes, SAT_STFT_NG * p.fb);
}

extern "C" int sat_stft_fwd(const float* x, const float* y, const float* views, float* partial, int NI, int C, int T,
                            int NV, int n_fft, int hop, void* stream) {
    SatStftParams p;
    if (NI <= 0 || T <= 0 || NV <= 0 || (C != 1 && C != 2)) { sat_set_error("sat_stft_fwd: bad shape (C must be 1 or 2)"); return 1; }
    if (sat_stft_plan(n_fft, hop, T, &p)) { sat_set_error("sat_stft_fwd: unsupported n_fft/hop/T (n_fft power of two in [8, 2048], hop <= n_fft, T > n_fft/2)"); return 1; }
    p.x = x; p.y = y; p.views = views; p.partial = partial; p.coef = nullptr; p.dy = nullptr;
    p.NI = NI; p.C = C; p.T = T; p.NV = NV; p.wrt_x = 0;
    dim3 grid(sat_cdiv(p.nframes, SAT_STFT_NG * p.fb), NI, sat_cdiv(NV, SAT_STFT_VB));
    const int bins = p.fb * (p.n / 2 + 1);
    if (p.fb * p.n <= SAT_STFT_SMALL && bins <= 2 * 256) SAT_LAUNCH((sat_stft_fwd_kernel<SAT_STFT_SMALL, 2, 2>), grid, dim3(256), stream, p);
    else if (p.fb * p.n <= SAT_STFT_MID && bins <= 3 * 256) SAT_LAUNCH((sat_stft_fwd_kernel<SAT_STFT_MID, 3, 2>), grid, dim3(256), stream, p);
    else SAT_LAUNCH((sat_stft_fwd_kernel<SAT_FFT_MAX, 5, SAT_STFT_CCMAX>), grid, dim3(256), stream, p);
    return sat_check_launch("sat_stft_fwd");
}

extern "C" int sat_stft_bwd(const float* x, const float* y, const float* views, const float* coef, float* dy, int NI,
                            int C, int T, int NV, int n_fft, int hop, int wrt_x, void* stream) {
    SatStftParams p;
    if (NI <= 0 || T <= 0 || NV <= 0 || (C != 1 && C != 2)) { sat_set_error("sat_stft_bwd: bad shape (C must be 1 or 2)"); return 1; }
    if (sat_stft_plan(n_fft, hop, T, &p)) { sat_set_error("sat_stft_bwd: unsupported n_fft/hop/T"); return 1; }
    p.x = x; p.y = y; p.views = views; p.partial = nullptr; p.coef = coef; p.dy = dy;
    p.NI = NI; p.C = C; p.T = T; p.NV = NV; p.wrt_x = wrt_x;
    dim3 grid(sat_cdiv(p.nframes, SAT_STFT_NG * p.fb), NI, 1);       // the views are looped inside the workgroup
    const int olen = (SAT_STFT_NG * p.fb - 1) * p.hop + p.n;
    // the write-out's even / odd planes assume a sample is touched by at most two NEIGHBOURING workgroups: a workgroup's span (olen) must not
    // exceed twice its stride (fpb * hop), i.e. n <= (fpb + 1) * hop — true for every hop >= n / 4 (the reference's configurations use n / 4)
    if (olen > 2 * SAT_STFT_NG * p.fb * p.hop) { sat_set_error("sat_stft_bwd: hop too small for this n_fft (needs n_fft <= (frames per workgroup + 1) * hop)"); return 1; }
    const int bins = p.fb * (p.n / 2 + 1);
    if (p.fb * p.n <= SAT_STFT_SMALL && olen <= SAT_STFT_OBUF_SMALL && bins <= 2 * 256) SAT_LAUNCH((sat_stft_bwd_kernel<SAT_STFT_SMALL, SAT_STFT_OBUF_SMALL, 2, 2>), grid, dim3(256), stream, p);
    else if (p.fb * p.n <= SAT_STFT_MID && olen <= SAT_STFT_OBUF_MID && bins <= 3 * 256) SAT_LAUNCH((sat_stft_bwd_kernel<SAT_STFT_MID, SAT_STFT_OBUF_MID, 3, 2>), grid, dim3(256), stream, p);
    else SAT_LAUNCH((sat_stft_bwd_kernel<SAT_FFT_MAX, SAT_STFT_OBUF, 5, SAT_STFT_CCMAX>), grid, dim3(256), stream, p);
    return sat_check_launch("sat_stft_bwd");
}

// ---------------------------------------------------------------------------------------------
// Complex spectrogram for the MS-STFT discriminator (reference: stable_audio_tools/models/encodec.py:73-76, :97-102 —
// torchaudio.transforms.Spectrogram(n_fft, hop, win_length = n_fft, hann window (periodic), normalized = True (by
// sqrt(sum w^2)), center = False, power = None), real and imaginary parts concatenated on the channel axis and the
// (freq, frame) axes swapped: z (NI, 2C, frames, n/2+1)).  torchaudio itself is not in the reference tree; its published
// algorithm is torch.stft(..., center=False, onesided) / ||w||_2.
//   sat_spec_fwd   x (NI, C, T), C in {1, 2}  ->  z[ni][c][f][k] = Re X_c, z[ni][C + c][f][k] = Im X_c,  X_c[f][k] =
//                  sum_j w_j x_c[f hop + j] e^{-2 pi i jk/n} / ||w||;  frames = 1 + (T - n) / hop.  The two channels of a
//                  stereo item ride in one complex FFT (as the loss kernels do).
//   sat_spec_bwd   dz -> dx, adjoint of the above: one complex FFT per channel and frame, frames overlap-added by gathering per
//                  output sample in LDS; even / odd workgroups write different planes dx[2][NI][C][T] (plain stores, the caller
//                  zero-fills and sums them): no atomics.
// ---------------------------------------------------------------------------------------------
struct SatSpecParams {
    const float* x;     // fwd in (NI, C, T)
    float* z;           // fwd out / bwd in (NI, 2C, frames, nb)
    float* dx;          // bwd out (2, NI, C, T)
    int NI, C, T, n, log2n, hop, nframes, fb;
};

__global__ void __launch_bounds__(256) sat_spec_fwd_kernel(SatSpecParams p) {
    __shared__ float re[SAT_FFT_MAX], im[SAT_FFT_MAX], twr[SAT_FFT_MAX / 2], twi[SAT_FFT_MAX / 2];
    __shared__ float wsum[4];
    const SatFftLds L{re, im, twr, twi};
    const int n = p.n, nb = (n >> 1) + 1, log2n = p.log2n;
    const int item = blockIdx.y;
    sat_fft_init_twiddles(L, n);
    __syncthreads();
    // ||w||^2 of the periodic Hann window = 3n/8 for n >= 4; computed from the table so that it matches the window used
    float ws = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) { const float w = sat_hann(L, j, n); ws += w * w; }
    ws = sat_wave_sum(ws);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = ws;
    __syncthreads();
    const float inv_norm = 1.0f / sqrtf(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
    const float* x0 = p.x + (size_t)item * p.C * p.T;
    for (int g = 0; g < SAT_STFT_NG; ++g) {
        const int f0 = (blockIdx.x * SAT_STFT_NG + g) * p.fb;
        if (f0 >= p.nframes) break;
        for (int i = threadIdx.x; i < p.fb * n; i += 256) {
            const int fi = i >> log2n, j = i & (n - 1);
            const int f = f0 + fi;
            float a = 0.f, b = 0.f;
            if (f < p.nframes) {
                const int t = f * p.hop + j;
                const float w = sat_hann(L, j, n) * inv_norm;
                a = w * x0[t];
                if (p.C > 1) b = w * x0[p.T + t];
            }
            const int jr = (int)(sat_brev((unsigned)j) >> (32 - log2n));
            re[fi * n + jr] = a;
            im[fi * n + jr] = b;
        }
        __syncthreads();
        sat_fft_run(L, n, log2n, p.fb);
        for (int i = threadIdx.x; i < p.fb * nb; i += 256) {
            const int fi = i / nb, k = i - fi * nb;
            const int f = f0 + fi;
            if (f < p.nframes) {
                float ar, ai, br, bi;
                sat_unpack_bins(L, n, fi, k, &ar, &ai, &br, &bi);
                float* zo = p.z + (((size_t)item * 2 * p.C) * p.nframes + f) * nb + k;
                const size_t cs = (size_t)p.nframes * nb;
                zo[0] = ar;
                if (p.C > 1) {
                    zo[cs] = br;
                    zo[2 * cs] = ai;
                    zo[3 * cs] = bi;
                } else {
                    zo[cs] = ai;
                }
            }
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) sat_spec_bwd_kernel(SatSpecParams p) {
    __shared__ float re[SAT_FFT_MAX], im[SAT_FFT_MAX], twr[SAT_FFT_MAX / 2], twi[SAT_FFT_MAX / 2];
    __shared__ float obuf[SAT_STFT_OBUF];
    __shared__ float wsum[4];
    const SatFftLds L{re, im, twr, twi};
    const int n = p.n, nb = (n >> 1) + 1, log2n = p.log2n;
    const int item = blockIdx.y;
    sat_fft_init_twiddles(L, n);
    __syncthreads();
    float ws = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) { const float w = sat_hann(L, j, n); ws += w * w; }
    ws = sat_wave_sum(ws);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = ws;
    __syncthreads();
    const float inv_norm = 1.0f / sqrtf(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
    const int fpb = SAT_STFT_NG * p.fb;
    const int olen = (fpb - 1) * p.hop + n;
    const int fbase = blockIdx.x * fpb;
    const size_t cs = (size_t)p.nframes * nb;
    for (int c = 0; c < p.C; ++c) {
        for (int i = threadIdx.x; i < olen; i += 256) obuf[i] = 0.f;
        __syncthreads();
        const float* gre = p.z + ((size_t)item * 2 * p.C + c) * cs;
        const float* gim = p.z + ((size_t)item * 2 * p.C + p.C + c) * cs;
        for (int g = 0; g < SAT_STFT_NG; ++g) {
            const int f0 = fbase + g * p.fb;
            if (f0 >= p.nframes) break;
            for (int i = threadIdx.x; i < p.fb * n; i += 256) {
                re[i] = 0.f;
                im[i] = 0.f;
            }
            __syncthreads();
            // dx_j = w_j Re( sum_{k <= n/2} G_k e^{+2 pi i jk/n} ) = w_j Re( DFT(conj G)[j] )
            for (int i = threadIdx.x; i < p.fb * nb; i += 256) {
                const int fi = i / nb, k = i - fi * nb;
                const int f = f0 + fi;
                if (f < p.nframes) {
                    const int kr = (int)(sat_brev((unsigned)k) >> (32 - log2n));
                    re[fi * n + kr] = gre[(size_t)f * nb + k];
                    im[fi * n + kr] = -gim[(size_t)f * nb + k];
                }
            }
            __syncthreads();
            sat_fft_run(L, n, log2n, p.fb);
            const int gspan = (p.fb - 1) * p.hop + n;
            const int obase = g * p.fb * p.hop;
            for (int orel = threadIdx.x; orel < gspan; orel += 256) {
                int f_lo = (orel - n + p.hop) / p.hop;
                if (orel - n + 1 <= 0) f_lo = 0;
                int f_hi = orel / p.hop;
                if (f_hi > p.fb - 1) f_hi = p.fb - 1;
                float v = 0.f;
                for (int fi = f_lo; fi <= f_hi; ++fi) {
                    const int j = orel - fi * p.hop;
                    if (f0 + fi < p.nframes) v += re[fi * n + j] * sat_hann(L, j, n);
                }
                obuf[obase + orel] += v * inv_norm;
            }
            __syncthreads();
        }
        const int last = (p.nframes - 1) * p.hop + n;     // one past the last sample any frame touches (<= T)
        float* d0 = p.dx + (size_t)(blockIdx.x & 1) * p.NI * p.C * p.T + ((size_t)item * p.C + c) * p.T;
        for (int i = threadIdx.x; i < olen; i += 256) {
            const int t = fbase * p.hop + i;
            if (t < last) d0[t] = obuf[i];
        }
        __syncthreads();
    }
}

static int sat_spec_plan(int n, int hop, int T, SatSpecParams* p) {
    SatStftParams q;
    if (T < n) return 1;
    if (sat_stft_plan(n, hop, T > n / 2 ? T : n, &q)) return 1;
    p->n = q.n; p->log2n = q.log2n; p->hop = hop; p->fb = q.fb;
    p->nframes = 1 + (T - n) / hop;
    return 0;
}
extern "C" int sat_spec_frames(int n_fft, int hop, int T) {
    SatSpecParams p;
    if (sat_spec_plan(n_fft, hop, T, &p)) return -1;
    return p.nframes;
}
extern "C" int sat_spec_fwd(const float* x, float* z, int NI, int C, int T, int n_fft, int hop, void* stream) {
    SatSpecParams p{};
    if (NI <= 0 || (C != 1 && C != 2)) { sat_set_error("sat_spec_fwd: bad shape (C must be 1 or 2)"); return 1; }
    if (sat_spec_plan(n_fft, hop, T, &p)) { sat_set_error("sat_spec_fwd: unsupported n_fft/hop/T (n_fft power of two in [8, 2048], hop <= n_fft <= T)"); return 1; }
    p.x = x; p.z = z; p.NI = NI; p.C = C; p.T = T;
    dim3 grid(sat_cdiv(p.nframes, SAT_STFT_NG * p.fb), NI);
    SAT_LAUNCH(sat_spec_fwd_kernel, grid, dim3(256), stream, p);
    return sat_check_launch("sat_spec_fwd");
}
extern "C" int sat_spec_bwd(const float* dz, float* dx, int NI, int C, int T, int n_fft, int hop, void* stream) {
    SatSpecParams p{};
    if (NI <= 0 || (C != 1 && C != 2)) { sat_set_error("sat_spec_bwd: bad shape (C must be 1 or 2)"); return 1; }
    if (sat_spec_plan(n_fft, hop, T, &p)) { sat_set_error("sat_spec_bwd: unsupported n_fft/hop/T"); return 1; }
    p.z = const_cast<float*>(dz); p.dx = dx; p.NI = NI; p.C = C; p.T = T;
    // (the even / odd planes of the write-out: a workgroup's span must not exceed twice its stride — see sat_stft_bwd)
    if ((SAT_STFT_NG * p.fb - 1) * p.hop + p.n > 2 * SAT_STFT_NG * p.fb * p.hop) { sat_set_error("sat_spec_bwd: hop too small for this n_fft"); return 1; }
    dim3 grid(sat_cdiv(p.nframes, SAT_STFT_NG * p.fb), NI);
    SAT_LAUNCH(sat_spec_bwd_kernel, grid, dim3(256), stream, p);
    return sat_check_launch("sat_spec_bwd");
}
