// disc_conv.hip — the Conv2d layers of the MS-STFT discriminator (stable_audio_tools/models/encodec.py:37-106: NormConv2d 3 x 9 /
// dilated 3 x 9 / 3 x 3, stride (1, 1), 'same' padding, LeakyReLU 0.2) on the bf16 matrix cores at fp32 accuracy (hi / lo split,
// three MFMAs per product: conv1d_bf16x3.hip), forward, data-gradient and weight-gradient.
//
// Layout ("pitched rows").  A (B, C, frames, freq) activation lives as (B, C, L): frame r occupies [r P, (r + 1) P) as
// [4 zeros | W samples | >= 4 zeros], P = ceil4(W + 8), L = frames * P.  A 'same' 1-D conv along that sequence never mixes two
// frames, so the frequency taps are a 1-D conv; the kh frame taps are "virtual channels": virtual channel (tap_t, c) is channel c
// read (tap_t - (kh-1)/2) * dil_t * P positions further along the SAME buffer — no shifted copies (round 2 materialised them:
// sat_rows_pack, 3 x the activation, and un-pitched every output again).  The kernels' matrix operands are the bf16 hi / lo PLANES of
// that sequence, [B][ceil(C/8)][rows][8 channels] with `lead` zero rows before position 0 and zero rows after L (they are the frame
// padding), written by the producing layer's epilogue (activation + pad mask applied) or by sat_disc_planes.
//
//   sat_disc_conv_kernel   64(co) x 512(positions) per workgroup, 8 waves of 64 x 64 (2 x 2 accumulators of 32 x 32), K-chunk = 16
//       virtual channels x kw taps: MFMA k-step tau = frequency tap tau, k-slots 0-7 / 8-15 = the chunk's two 8-channel groups —
//       the structure of conv1d_bf16x3_k7q.h (LDS-DMA of lane-linear 1-KiB pieces, two 72-KiB stages, the two wave rows ONE BARRIER
//       APART, fragment reads + next chunk's DMA in one wave row under the other's MFMAs), three phases of three taps per chunk.
//       The data-gradient is the same kernel on the planes of dL/d(pre-activation) with the weights packed transposed / flipped.
//   sat_disc_wgrad_kernel  dW = sum over positions of dy (x) shifted x: conv_wgrad7_bf16x3_pipe.h's register-staged pipeline
//       (global loads of stage c+2 | MFMAs of stage c | hi/lo split of stage c+1 into LDS) re-tiled for 64 output channels:
//       64(co) x 64(virtual ci) x kw taps per workgroup, the two halves of a 128-position stage on different waves.
#include "sat_device.h"

#define SAT_DC_CO 64
#define SAT_DC_T 512
#define SAT_DC_NT 512
#define SAT_DC_TAPS 9
#define SAT_DC_AROWS 576                                   // 512 positions + 8 taps of halo, in 64-row pieces
#define SAT_DC_WBYTES (SAT_DC_TAPS * 2 * 2 * 1024)         // [tap][plane][group][64 co][16 B]
#define SAT_DC_ABYTES (2 * 2 * SAT_DC_AROWS * 16)          // [plane][group][576 rows][16 B]
#define SAT_DC_STAGE (SAT_DC_WBYTES + SAT_DC_ABYTES)       // 73728
#define SAT_DC_APIECES (2 * 2 * (SAT_DC_AROWS / 64))       // 36
#define SAT_DC_MAXSHIFT 4                                  // max |frame shift| = dil_t * (kh - 1) / 2 frames

SAT_DEVICE void sat_dc_split2(float x, short* hi, short* lo) {
    const short h = sat_f32_to_bf16(x);
    *hi = h;
    *lo = sat_f32_to_bf16(x - sat_bf16_to_f32(h));
}

// ---- geometry of the pitched sequence and of its planes ----
extern "C" int sat_disc_geom(int frames, int W, int* P, int* L, int* lead, int* rows) {
    if (frames <= 0 || W <= 0) { sat_set_error("sat_disc_geom: empty shape"); return 1; }
    const int p = (W + 8 + 3) / 4 * 4;
    const long long l = (long long)frames * p;
    const int ld = sat_cdiv(SAT_DC_MAXSHIFT * p + 8, 64) * 64;
    const long long r = ld + sat_cdivll(l, SAT_DC_T) * SAT_DC_T + ld + 64;
    if (r * 16 >= (1ll << 40) || l >= (1ll << 31) - 1024) { sat_set_error("sat_disc_geom: sequence too long"); return 1; }
    *P = p; *L = (int)l; *lead = ld; *rows = (int)r;
    return 0;
}

// ---- weights: w (Cout, Cin, kh, kw) fp32 -> ONE bf16 buffer [co tile][chunk][tap][plane hi | lo][group g][64 m][e] (a K-chunk's
//      4 kw one-KiB pieces are contiguous, taps in the order the kernel's phases consume them), element = W'[m][v][tap] with
//      virtual channel v: group gv = chunk * 2 + g = tap_t * c8 + cg, channel c = cg * 8 + e (c8 = groups of the conv INPUT):
//   mode 0 (conv):          m = co, c = ci:  W' = w[m][c][tap_t][tap]
//   mode 1 (data-gradient): m = ci, c = co:  W' = w[c][m][kh-1-tap_t][kw-1-tap]
struct SatDiscPackParams {
    const float* w;
    short* wq;
    int Cout, Cin, kh, kw, mode, c8, nchunks, m_v, c_v;
    long long total;
};
__global__ void __launch_bounds__(256) sat_disc_pack_kernel(SatDiscPackParams p) {
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= p.total) return;
    const int e = (int)(o & 7), ml = (int)((o >> 3) & 63), g = (int)((o >> 9) & 1);
    long long q = o >> 10;
    const int tap = (int)(q % p.kw);
    q /= p.kw;
    const int chunk = (int)(q % p.nchunks), tile = (int)(q / p.nchunks);
    const int m = tile * SAT_DC_CO + ml;
    const int gv = chunk * 2 + g;
    float val = 0.0f;
    if (gv < p.kh * p.c8 && m < p.m_v) {
        const int tap_t = gv / p.c8, c = (gv - tap_t * p.c8) * 8 + e;
        if (c < p.c_v) {
            if (p.mode == 0) val = p.w[(((size_t)m * p.Cin + c) * p.kh + tap_t) * p.kw + tap];
            else val = p.w[(((size_t)c * p.Cin + m) * p.kh + (p.kh - 1 - tap_t)) * p.kw + (p.kw - 1 - tap)];
        }
    }
    short h, l;
    sat_dc_split2(val, &h, &l);
    const long long oq = (o >> 10) * 2048 + (o & 1023);    // (tile, chunk, tap) block of [plane][group][64][8]
    p.wq[oq] = h;
    p.wq[oq + 1024] = l;
}
static bool sat_disc_pack_geometry(int Cout, int Cin, int kh, int kw, int mode, SatDiscPackParams* p) {
    if (mode < 0 || mode > 1 || Cout <= 0 || Cin <= 0 || kh < 1 || !(kh & 1) || kw < 1 || kw > SAT_DC_TAPS || !(kw & 1)) return false;
    p->Cout = Cout; p->Cin = Cin; p->kh = kh; p->kw = kw; p->mode = mode;
    p->m_v = mode == 0 ? Cout : Cin;
    p->c_v = mode == 0 ? Cin : Cout;
    p->c8 = sat_cdiv(p->c_v, 8);
    p->nchunks = sat_cdiv(kh * p->c8, 2);
    p->total = (long long)sat_cdiv(p->m_v, SAT_DC_CO) * p->nchunks * kw * 2 * SAT_DC_CO * 8;
    return true;
}
extern "C" long long sat_disc_pack_size(int Cout, int Cin, int kh, int kw, int mode) {
    SatDiscPackParams p{};
    return sat_disc_pack_geometry(Cout, Cin, kh, kw, mode, &p) ? 2 * p.total : -1;
}
extern "C" int sat_disc_pack_weights(const float* w, short* wq, int Cout, int Cin, int kh, int kw, int mode, void* stream) {
    SatDiscPackParams p{};
    if (!sat_disc_pack_geometry(Cout, Cin, kh, kw, mode, &p)) { sat_set_error("sat_disc_pack_weights: odd kh, odd kw <= 9, mode 0|1"); return 1; }
    p.w = w; p.wq = wq;
    SAT_LAUNCH(sat_disc_pack_kernel, dim3((unsigned)sat_cdivll(p.total, 256)), dim3(256), stream, p);
    return sat_check_launch("sat_disc_pack_weights");
}

// ---- planes / pitched fp32 from a tensor: one thread = one position of one 8-channel group ----
//   src: (B, C, frames, W) (pitched == 0) or the pitched (B, C, L); `out` (pitched, or null): the values are multiplied by
//   LeakyReLU'(out) = out > 0 ? 1 : slope (the gradient w.r.t. a layer's pre-activation from the gradient w.r.t. its output);
//   pad positions are written as zeros.  dst (pitched fp32, or null) and hi / lo (planes, or null) receive the result.
//   fm_sign / fm_coef (with `out`): the L1 feature-matching term of this layer's output rides along — the gradient w.r.t. the output
//   is src + fm_coef[0] * fm_sign (fm_sign = sign(out - ref) as int8, written by sat_disc_l1_sum in the forward pass — the other
//   signal's feature map itself need not stay alive; fm_coef: a device scalar, dL/d(sum |out - ref|)) before the LeakyReLU' factor.
struct SatDiscPlanesParams {
    const float* src;
    const float* out;
    const signed char* fm_sign;
    const float* fm_coef;
    float* dst;
    short* hi;
    short* lo;
    int B, C, c8, frames, W, P, L, lead, rows, pitched;
    float slope;
};
__global__ void __launch_bounds__(256) sat_disc_planes_kernel(SatDiscPlanesParams p) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y, b = blockIdx.z;
    if (t >= p.L) return;
    const int r = t / p.P, f = t - r * p.P - 4;
    const bool valid = (unsigned)f < (unsigned)p.W;
    const float fmc = p.fm_sign ? p.fm_coef[0] : 0.0f;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = g * 8 + e;
        float o = 0.0f;
        if (valid && ch < p.C) {
            const size_t ip = ((size_t)b * p.C + ch) * p.L + t;
            o = p.pitched ? p.src[ip] : p.src[(((size_t)b * p.C + ch) * p.frames + r) * p.W + f];
            if (p.out) {
                const float ov = p.out[ip];
                if (p.fm_sign) o += fmc * (float)p.fm_sign[ip];
                o *= (ov > 0.0f ? 1.0f : p.slope);
            }
        }
        if (p.dst && ch < p.C) p.dst[((size_t)b * p.C + ch) * p.L + t] = o;
        v[e] = o;
    }
    if (p.hi) {
        uint32_t h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sat_split2_pk(v[2 * j], v[2 * j + 1], &h[j], &l[j]);
        const size_t o = (((size_t)b * p.c8 + g) * p.rows + p.lead + t) * 8;
        *reinterpret_cast<u32x4*>(p.hi + o) = u32x4{h[0], h[1], h[2], h[3]};
        *reinterpret_cast<u32x4*>(p.lo + o) = u32x4{l[0], l[1], l[2], l[3]};
    }
}
extern "C" int sat_disc_planes(const float* src, const float* out, const signed char* fm_sign, const float* fm_coef, float* dst, void* hi, void* lo,
                               int B, int C, int frames, int W, int pitched, float slope, void* stream) {
    int P, L, lead, rows;
    if (B <= 0 || C <= 0 || sat_disc_geom(frames, W, &P, &L, &lead, &rows)) { sat_set_error("sat_disc_planes: bad shape"); return 1; }
    if (!src || (!dst && !hi) || ((hi == nullptr) != (lo == nullptr))) { sat_set_error("sat_disc_planes: missing operand"); return 1; }
    if (fm_sign && (!out || !fm_coef || !pitched)) { sat_set_error("sat_disc_planes: the feature-matching term needs out, fm_coef and a pitched src"); return 1; }
    SatDiscPlanesParams p{src, out, fm_sign, fm_coef, dst, (short*)hi, (short*)lo, B, C, sat_cdiv(C, 8), frames, W, P, L, lead, rows, pitched, slope};
    SAT_LAUNCH(sat_disc_planes_kernel, dim3(sat_cdiv(L, 256), p.c8, B), dim3(256), stream, p);
    return sat_check_launch("sat_disc_planes");
}

// sum |a - b| over n floats (n % 4 == 0, 16-byte aligned): partial[block], 1024 blocks of grid-stride float4 loads — the L1 feature-
// matching distance of two feature maps in the pitched layout (pad positions are zero in both); sign (or null): sign(a - b) as int8,
// all the backward needs of b (sat_disc_planes fm_sign)
struct SatDiscL1Params { const float* a; const float* b; float* partial; signed char* sign; long long n4; };
__global__ void __launch_bounds__(256) sat_disc_l1_kernel(SatDiscL1Params p) {
    __shared__ float red[4];
    float s = 0.0f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.n4; i += (long long)gridDim.x * 256) {
        const f32x4 x = reinterpret_cast<const f32x4*>(p.a)[i], y = reinterpret_cast<const f32x4*>(p.b)[i];
        s += (fabsf(x[0] - y[0]) + fabsf(x[1] - y[1])) + (fabsf(x[2] - y[2]) + fabsf(x[3] - y[3]));
        if (p.sign) {                                      // sign(a - b) per element, four int8 per store
            uint32_t w = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = x[e] - y[e];
                const uint32_t sg = d > 0.0f ? 0x01u : (d < 0.0f ? 0xffu : 0x00u);
                w |= sg << (8 * e);
            }
            reinterpret_cast<uint32_t*>(p.sign)[i] = w;
        }
    }
    s = sat_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) p.partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
extern "C" int sat_disc_l1_blocks(void) { return 1024; }
extern "C" int sat_disc_l1_sum(const float* a, const float* b, float* partial, signed char* sign, long long n, void* stream) {
    if (!a || !b || !partial || n <= 0 || (n & 3) || (((uintptr_t)a | (uintptr_t)b) & 15) || ((uintptr_t)sign & 3)) { sat_set_error("sat_disc_l1_sum: n % 4 == 0, 16-byte aligned operands"); return 1; }
    SatDiscL1Params p{a, b, partial, sign, n >> 2};
    SAT_LAUNCH(sat_disc_l1_kernel, dim3(1024), dim3(256), stream, p);
    return sat_check_launch("sat_disc_l1_sum");
}

// =====================================================================================================================
struct SatDiscConvParams {
    const short* xp_hi;   // input planes [B][c8][rows][8]
    const short* xp_lo;
    const short* wq;      // sat_disc_pack_weights
    const float* bias;    // (Cout) or null
    float* y;             // (B, Cout, L) pitched
    short* em_hi;         // planes of y [B][em_c8][rows][8] for the next layer, or null
    short* em_lo;
    int B, c8, Cout, em_c8;
    int rows, lead, P, W, L;
    int kh, kw, shift;    // shift = dil_t * P: positions between frame taps
    int nchunks, t_tiles, co_tiles;
    float slope;          // LeakyReLU slope of the epilogue (1: none)
    const float* lk_src;  // (B, Cout, L) or null: the result is multiplied by LeakyReLU'(lk_src) = lk_src > 0 ? 1 : lk_slope — a
    float lk_slope;       // data-gradient that leaves as dL/d(pre-activation) of the layer that produced lk_src (its activated output)
};

template <int KW>
__global__ void __launch_bounds__(SAT_DC_NT) sat_disc_conv_kernel(SatDiscConvParams p) {
    constexpr int CO_T = SAT_DC_CO, T_T = SAT_DC_T, AROWS = SAT_DC_AROWS, STAGE = SAT_DC_STAGE, WB = SAT_DC_WBYTES;
    // ONE LDS object (a second one makes hipcc drain the LDS-DMA queue in front of every fragment read)
    __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE + CO_T * 4];
    float* bias_lds = reinterpret_cast<float*>(lds + 2 * STAGE);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = SAT_UNIFORM(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 2;                              // wave row: the one-barrier-apart halves of the workgroup
    const int t_w = wave * 64;
    constexpr int kw = KW;
    const int kh = p.kh;
    const int pad_w = (kw - 1) >> 1, pad_t = (kh - 1) >> 1;
    constexpr int nph = (kw + 2) / 3;                      // phases of three taps
    constexpr int ipp = (9 + nph - 1) / nph;               // DMA issue slots (of 9) per phase
    // kw <= 3 (the 3 x 3 layers): ONE phase per chunk — with two stages the chunk's only read section would have to request the next
    // chunk AND wait for it (round 3: vmcnt(0) in the phase of the request: 0.24 of the peak).  Its stage is 48 KB (12 weight + 36
    // activation pieces), so the same 144 KB hold a THREE-stage ring: chunk c's read section requests chunk c + 2 and waits for chunk
    // c + 1 only (counted: the six pieces just issued stay in flight) — gemm.hip's sat_gemm8_kernel schedule.
    constexpr bool RING3 = (nph == 1);
    constexpr int WBK = RING3 ? KW * 4096 : WB;            // bytes of a stage's weight part = offset of its activation part
    constexpr int STG = RING3 ? KW * 4096 + SAT_DC_ABYTES : STAGE;
    constexpr int NSTG = RING3 ? 3 : 2;
    constexpr int NPW = 5 + (KW >= 3 ? 1 : 0);             // RING3: LDS-DMA instructions per wave and chunk (slots 0-4, slot 5 for kw = 3)
    static_assert(NSTG * STG <= 2 * STAGE, "the ring lives in the two-stage allocation");
    // tile order: the tiles are dealt round-robin to the 8 XCDs; give every XCD a CONTIGUOUS range of the sequence so that the frame
    // taps (the same plane rows, read again by the tiles dil_t * P positions earlier / later) and the halos hit in its L2
    int L = (int)blockIdx.x;
    {
        const int n = (int)gridDim.x, per = n >> 3;
        if (L < per * 8) L = (L & 7) * per + (L >> 3);
    }
    const int co_tile = L % p.co_tiles;
    const int win = L / p.co_tiles;
    const int b = win / p.t_tiles, t_tile = win - b * p.t_tiles;
    const int co0 = co_tile * CO_T, t0 = t_tile * T_T;
    const int row_in0 = p.lead + t0 - pad_w;

    if (tid < CO_T) bias_lds[tid] = (co0 + tid < p.Cout && p.bias) ? p.bias[co0 + tid] : 0.0f;

    // ---- LDS-DMA of a chunk: 36 activation pieces ((plane, group) x 9 x 64 rows) + 4 kw weight pieces (tap, plane, group: 64 co x 16 B)
    //      in nine issue slots per wave: slots 0-3 = (plane, group) s, rows 64 wave ..; slot 4 = the ninth row piece of (plane, group)
    //      `wave` (waves 0-3) | weight pieces 0-3 (waves 4-7); slots 5-8 = weight pieces 8 (s - 5) + wave + 4.  Every source address is
    //      a wave-uniform base + lane * 16 bytes, and the bases are a handful of scalar adds per piece: the per-chunk part (which
    //      virtual groups, i.e. which frame tap / channel group / row shift) is computed once per chunk (set_chunk) ----
    const unsigned lane16 = (unsigned)lane * 16u;
    const int nvg = kh * p.c8;
    const char* wq_tile = (const char*)p.wq + (size_t)co_tile * p.nchunks * (kw * 4096);
    long long aoff[2];                                     // byte offset (within a plane) of row row_in0 of the chunk's two groups
    auto set_chunk = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            int gv = c * 2 + g;
            gv = gv < nvg ? gv : nvg - 1;                  // past the end: any finite rows (their weights are zero)
            const int tap_t = gv / p.c8, cg = gv - tap_t * p.c8;
            aoff[g] = (((long long)b * p.c8 + cg) * p.rows + row_in0 + (long long)(tap_t - pad_t) * p.shift) * 16;
        }
    };
    auto issue_w = [&](int c, char* stage, int r) __attribute__((always_inline)) {
        if (r < 4 * kw) sat_glds16(wq_tile + ((size_t)c * (4 * kw) + r) * 1024 + lane16, stage + r * 1024);
    };
    auto issue = [&](int c, int st, auto slot_c) __attribute__((always_inline)) {
        constexpr int s = decltype(slot_c)::value;
        char* stage = lds + st * STG;
        if constexpr (s < 4) {
            const char* src = (const char*)((s >> 1) ? p.xp_lo : p.xp_hi) + aoff[s & 1] + wave * 1024;
            sat_glds16(src + lane16, stage + WBK + (s * AROWS + wave * 64) * 16);
        } else if constexpr (s == 4) {
            if (wave < 4) {
                const char* src = (const char*)((wave >> 1) ? p.xp_lo : p.xp_hi) + aoff[wave & 1] + 8 * 1024;
                sat_glds16(src + lane16, stage + WBK + (wave * AROWS + 512) * 16);
            } else {
                issue_w(c, stage, wave - 4);
            }
        } else {
            issue_w(c, stage, (s - 5) * 8 + wave + 4);
        }
    };
    auto issue_range = [&](int c, int st, auto lo_c, auto hi_c) __attribute__((always_inline)) {
        constexpr int lo = decltype(lo_c)::value, hi_ = decltype(hi_c)::value;
        if constexpr (lo < hi_ && lo < 9) {
            issue(c, st, std::integral_constant<int, lo>{});
            if constexpr (lo + 1 < hi_ && lo + 1 < 9) issue(c, st, std::integral_constant<int, lo + 1>{});
            if constexpr (lo + 2 < hi_ && lo + 2 < 9) issue(c, st, std::integral_constant<int, lo + 2>{});
            if constexpr (lo + 3 < hi_ && lo + 3 < 9) issue(c, st, std::integral_constant<int, lo + 3>{});
            if constexpr (lo + 4 < hi_ && lo + 4 < 9) issue(c, st, std::integral_constant<int, lo + 4>{});
            if constexpr (lo + 5 < hi_ && lo + 5 < 9) issue(c, st, std::integral_constant<int, lo + 5>{});
            if constexpr (lo + 6 < hi_ && lo + 6 < 9) issue(c, st, std::integral_constant<int, lo + 6>{});
            if constexpr (lo + 7 < hi_ && lo + 7 < 9) issue(c, st, std::integral_constant<int, lo + 7>{});
            if constexpr (lo + 8 < hi_ && lo + 8 < 9) issue(c, st, std::integral_constant<int, lo + 8>{});
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    struct Frags { bf16x8 wa[2][2], xa[2][2]; };          // [mi | ni][plane]
    Frags fr[3];
    auto load_frags = [&](Frags& f, const char* sb, int tap) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const char* wb = sb + ((tap * 2 + pl) * 2 + hi) * 1024;
            f.wa[0][pl] = *reinterpret_cast<const bf16x8*>(wb + l31 * 16);
            f.wa[1][pl] = *reinterpret_cast<const bf16x8*>(wb + (32 + l31) * 16);
            const char* ab = sb + WBK + (pl * 2 + hi) * (AROWS * 16);
            f.xa[0][pl] = *reinterpret_cast<const bf16x8*>(ab + (t_w + l31 + tap) * 16);
            f.xa[1][pl] = *reinterpret_cast<const bf16x8*>(ab + (t_w + 32 + l31 + tap) * 16);
        }
    };
    auto mfma_frags = [&](const Frags& f) __attribute__((always_inline)) {
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

    // prologue: chunk 0 complete in stage 0 (RING3: chunk 1 requested as well)
    set_chunk(0);
    issue_range(0, 0, std::integral_constant<int, 0>{}, std::integral_constant<int, 9>{});
    if constexpr (RING3) {
        if (p.nchunks > 1) {
            set_chunk(1);
            issue_range(1, 1, std::integral_constant<int, 0>{}, std::integral_constant<int, 9>{});
            SAT_WAIT_VMCNT(NPW);
        } else {
            SAT_WAIT_VMCNT(0);
        }
    } else {
        SAT_WAIT_VMCNT(0);
    }
    SAT_RAW_BARRIER();
    if (wr == 1) SAT_RAW_BARRIER();                        // the second wave row runs one barrier behind the first

    if constexpr (RING3) {
        // wave row 0 reads chunk c in interval 2c, row 1 in 2c + 1.  Chunk c + 2 goes to the stage of chunk c - 1, whose last fragment read
        // retired (lgkmcnt(0)) before the barrier that closes interval 2c - 1; a wave's pieces of chunk c + 1 are waited for here, in
        // front of a barrier every reader of chunk c + 1 (intervals 2c + 2 / 2c + 3) has passed.
        int st = 0;
        for (int c = 0; c < p.nchunks; ++c) {
            const char* sb = lds + st * STG;
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (u < kw) load_frags(fr[u], sb, u);
            if (c + 2 < p.nchunks) {
                set_chunk(c + 2);
                issue_range(c + 2, st == 0 ? 2 : st - 1, std::integral_constant<int, 0>{}, std::integral_constant<int, 9>{});
                SAT_WAIT_VMCNT(NPW);
            } else {
                SAT_WAIT_VMCNT(0);
            }
            SAT_WAIT_LGKM0();
            SAT_RAW_BARRIER();
            SAT_SCHED_FENCE();
            SAT_SETPRIO(1);
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (u < kw) mfma_frags(fr[u]);
            SAT_SETPRIO(0);
            SAT_SCHED_FENCE();
            SAT_RAW_BARRIER();
            st = st == 2 ? 0 : st + 1;
        }
    } else {
    // kw == 9: COUNTED waits — the last phase's three pieces per wave (weights of taps 3-8, L2-resident) stay in flight across the
    // chunk boundary and are retired by the next chunk's phase 0, one phase before they are read (as conv1d_bf16x3_k7q.h, VARIANT 1)
    constexpr bool COUNTED = (KW == 9);
    for (int c = 0; c < p.nchunks; ++c) {
        const char* sb = lds + (c & 1) * STAGE;
        const bool more = c + 1 < p.nchunks;
        if (more) set_chunk(c + 1);
#pragma unroll
        for (int ph = 0; ph < nph; ++ph) {
            // read section: this phase's fragments + this phase's share of the next chunk's DMA
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (3 * ph + u < kw) load_frags(fr[u], sb, 3 * ph + u);
            if (more) {
                if (ph == 0) issue_range(c + 1, (c + 1) & 1, std::integral_constant<int, 0>{}, std::integral_constant<int, ipp>{});
                if (ph == 1) issue_range(c + 1, (c + 1) & 1, std::integral_constant<int, ipp>{}, std::integral_constant<int, 2 * ipp>{});
                if (ph == 2) issue_range(c + 1, (c + 1) & 1, std::integral_constant<int, 2 * ipp>{}, std::integral_constant<int, 3 * ipp>{});
            }
            if constexpr (COUNTED) {
                if (ph == 0) { if (more) { SAT_WAIT_VMCNT(3); } else { SAT_WAIT_VMCNT(0); } }
                if (ph == 2 && more) { SAT_WAIT_VMCNT(3); }
            } else {
                if (ph == nph - 1) { SAT_WAIT_VMCNT(0); }  // chunk c+1 has landed (this wave's pieces; the barriers publish the others')
            }
            SAT_WAIT_LGKM0();
            SAT_RAW_BARRIER();
            SAT_SCHED_FENCE();
            SAT_SETPRIO(1);
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (3 * ph + u < kw) mfma_frags(fr[u]);
            SAT_SETPRIO(0);
            SAT_SCHED_FENCE();
            SAT_RAW_BARRIER();
        }
    }
    }
    if (wr == 0) SAT_RAW_BARRIER();                        // pairs with the second wave row's last barrier
    __syncthreads();                                       // every wave is done with the stages: their memory serves the epilogue

    // ---- epilogue: bias, LeakyReLU, pad mask; 16-byte stores through a per-wave LDS transposition (32 rows x 68 floats in the drained
    //      stage memory); plane emission = the same tile read column-wise (8 consecutive channels of a position = one plane row) ----
    float (*tile)[68] = reinterpret_cast<float (*)[68]>(lds) + wave * 32;
    const int lr = lane >> 4, t4 = (lane & 15) * 4;
    const int tg = t0 + t_w + t4;                          // this lane's four positions (L % 4 == 0, P % 4 == 0: same frame, all in / out)
    const int rem = tg % p.P;
    bool valid[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) valid[e] = tg < p.L && (unsigned)(rem + e - 4) < (unsigned)p.W;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        if (mi == 1) __syncthreads();                      // every wave is done reading its first half
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) tile[(r & 3) + 8 * (r >> 2) + 4 * hi][ni * 32 + l31] = acc[mi][ni][r];
        __syncthreads();                                    // (a wave only reads its own tile: this orders its own lanes)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = j * 4 + lr;
            const int co = co0 + mi * 32 + row;
            const bool ok = co < p.Cout && tg < p.L;
            const f32x4 av = *reinterpret_cast<const f32x4*>(&tile[row][t4]);
            const float bias = bias_lds[mi * 32 + row];
            f32x4 lk = {1.f, 1.f, 1.f, 1.f};
            if (p.lk_src && ok) lk = *reinterpret_cast<const f32x4*>(p.lk_src + ((size_t)b * p.Cout + co) * p.L + tg);
            f32x4 ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = av[e] + bias;
                v = v > 0.0f ? v : v * p.slope;
                if (p.lk_src) v *= (lk[e] > 0.0f ? 1.0f : p.lk_slope);
                ov[e] = valid[e] ? v : 0.0f;
            }
            if (ok) *reinterpret_cast<f32x4*>(p.y + ((size_t)b * p.Cout + co) * p.L + tg) = ov;
            if (p.em_hi) {
                if (!ok) ov = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(&tile[row][t4]) = ov;
            }
        }
        if (p.em_hi) {
            __syncthreads();
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c8i = ((co0 + mi * 32) >> 3) + g;
                const int tq = t0 + t_w + lane;
                float v8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = tile[g * 8 + e][lane];
                uint32_t eh[4], el[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) sat_split2_pk(v8[2 * e], v8[2 * e + 1], &eh[e], &el[e]);
                if (c8i < p.em_c8 && tq < p.L) {
                    const size_t o = (((size_t)b * p.em_c8 + c8i) * p.rows + p.lead + tq) * 8;
                    *reinterpret_cast<u32x4*>(p.em_hi + o) = u32x4{eh[0], eh[1], eh[2], eh[3]};
                    *reinterpret_cast<u32x4*>(p.em_lo + o) = u32x4{el[0], el[1], el[2], el[3]};
                }
            }
        }
    }
}

// y = LeakyReLU_slope(conv2d(x, w) + bias) on the pitched layout (slope 1: no activation), pad positions written as zeros:
//   xp_hi / xp_lo: planes of x (Cin channels; sat_disc_planes or a previous layer's emission); wq: sat_disc_pack_weights
//   (mode 0; mode 1 with Cin / Cout swapped: the data-gradient, xp = planes of dL/d(pre-activation)); y (B, Cout, L) fp32;
//   em_hi / em_lo (or null): the planes of y for the layer that consumes it.  kh frame taps dil_t frames apart, kw <= 9 frequency taps.
//   lk_src (or null): y *= LeakyReLU'(lk_src) with slope lk_slope — the data-gradient of a layer whose input was the activated output
//   lk_src of the previous layer, leaving directly as that layer's dL/d(pre-activation) (fp32 + planes), when nothing else consumes lk_src.
extern "C" int sat_disc_conv(const void* xp_hi, const void* xp_lo, const void* wq, const float* bias, float* y,
                             void* em_hi, void* em_lo, int B, int Cin, int Cout, int frames, int W, int kh, int kw, int dil_t,
                             float slope, const float* lk_src, float lk_slope, void* stream) {
    int P, L, lead, rows;
    if (B <= 0 || Cin <= 0 || Cout <= 0 || sat_disc_geom(frames, W, &P, &L, &lead, &rows)) { sat_set_error("sat_disc_conv: bad shape"); return 1; }
    if (kh < 1 || !(kh & 1) || kw < 1 || kw > SAT_DC_TAPS || !(kw & 1) || dil_t < 1 || dil_t * ((kh - 1) / 2) > SAT_DC_MAXSHIFT) {
        sat_set_error("sat_disc_conv: odd kh, odd kw <= 9, dil_t * (kh - 1) / 2 <= 4");
        return 1;
    }
    if (!xp_hi || !xp_lo || !wq || !y || ((em_hi == nullptr) != (em_lo == nullptr)) || (((uintptr_t)y | (uintptr_t)lk_src) & 15)) {
        sat_set_error("sat_disc_conv: missing / misaligned operand");
        return 1;
    }
    SatDiscConvParams p{};
    p.xp_hi = (const short*)xp_hi; p.xp_lo = (const short*)xp_lo; p.wq = (const short*)wq;
    p.bias = bias; p.y = y; p.em_hi = (short*)em_hi; p.em_lo = (short*)em_lo;
    p.B = B; p.c8 = sat_cdiv(Cin, 8); p.Cout = Cout; p.em_c8 = sat_cdiv(Cout, 8);
    p.rows = rows; p.lead = lead; p.P = P; p.W = W; p.L = L;
    p.kh = kh; p.kw = kw; p.shift = dil_t * P;
    p.nchunks = sat_cdiv(kh * p.c8, 2);
    p.t_tiles = sat_cdiv(L, SAT_DC_T);
    p.co_tiles = sat_cdiv(Cout, SAT_DC_CO);
    p.slope = slope;
    p.lk_src = lk_src; p.lk_slope = lk_slope;
    const long long total = (long long)p.t_tiles * B * p.co_tiles;
    const dim3 grid((unsigned)total), block(SAT_DC_NT);
    switch (kw) {
        case 1: SAT_LAUNCH((sat_disc_conv_kernel<1>), grid, block, stream, p); break;
        case 3: SAT_LAUNCH((sat_disc_conv_kernel<3>), grid, block, stream, p); break;
        case 5: SAT_LAUNCH((sat_disc_conv_kernel<5>), grid, block, stream, p); break;
        case 7: SAT_LAUNCH((sat_disc_conv_kernel<7>), grid, block, stream, p); break;
        default: SAT_LAUNCH((sat_disc_conv_kernel<9>), grid, block, stream, p); break;
    }
    return sat_check_launch("sat_disc_conv");
}

// =====================================================================================================================
// weight gradient.  dW[m][(tap_t, c)][tap] = sum_b sum_t dy[b][m][t] * x[b][c][t + (tap_t - pad_t) * shift + tap - pad_w]
//
// One workgroup = 8 waves = 64(co) x 64(virtual ci) x kw taps over a range of 128-position stages: wave w owns the 32 x 32 tile
// (co half w & 1, ci half (w >> 1) & 1) of taps 0..4 (w < 4) or 5..8 (w >= 4) — five / four accumulators, 120 / 96 MFMAs per stage;
// waves w and w + 4 share a SIMD: the first multiplies while the second converts the next stage, then they swap.
#define SAT_DW_NT 512
#define SAT_DW_TT 128                // positions per stage (8 MFMA k-steps)
#define SAT_DW_ROW 136               // 128 + 8 taps of halo / pad: 272-byte rows (68 dwords = 4 x odd: conflict-free b128)
#define SAT_DW_NI 64                 // virtual input channels per workgroup

struct SatDiscWgParams {
    const float* dy;     // (B, M, L)
    const float* x;      // (B, Cr, L)
    float* out;          // [nsplit][kw][m_pad][n_pad]
    int B, M, Cr, N, L, kh, shift, pad_w;
    int m_pad, n_pad;
    int chunks_per_split, nchunks, nT;
};

#if defined(SAT_HIPEMU)
static inline unsigned sat_dw_alignbit(unsigned hi, unsigned lo, unsigned s) { return (unsigned)((((unsigned long long)hi << 32) | lo) >> s); }
#else
SAT_DEVICE unsigned sat_dw_alignbit(unsigned hi, unsigned lo, unsigned s) { return __builtin_amdgcn_alignbit(hi, lo, s); }
#endif

template <int KW>
__global__ void __launch_bounds__(SAT_DW_NT) sat_disc_wgrad_kernel(SatDiscWgParams p) {
    constexpr int NXU = 5;                                           // x staging: 16 threads per row walk its 68 pairs 16 at a time
    constexpr bool PAIR = (((KW - 1) / 2) & 1) == 0;                 // pad_w even: the pairs are 8-byte aligned
    constexpr int K0 = (KW + 1) / 2;                                 // taps of the first wave group; the second takes K0 .. KW-1
    __shared__ __attribute__((aligned(16))) short y_lds0[2][SAT_DC_CO][SAT_DW_ROW], y_lds1[2][SAT_DC_CO][SAT_DW_ROW];   // dy [plane][co][t]
    __shared__ __attribute__((aligned(16))) short x_lds0[2][SAT_DW_NI][SAT_DW_ROW], x_lds1[2][SAT_DW_NI][SAT_DW_ROW];   // x  [plane][n][t]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int m_w = (wave & 1) * 32, n_w = ((wave >> 1) & 1) * 32;
    const bool first = wave < 4;                                     // tap group 0: MFMAs first; tap group 1: conversion first
    // grid = (m tiles, n tiles, splits): the (m, n) tiles of one split read the same dy / x positions -> same XCD (sat_xcd_tile)
    int mn_tile, split;
    sat_xcd_tile(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y, gridDim.z, &mn_tile, &split);
    const int m0 = (mn_tile % (int)gridDim.x) * SAT_DC_CO, n0 = (mn_tile / (int)gridDim.x) * SAT_DW_NI;
    const int pad_t = (p.kh - 1) >> 1;

    f32x16 acc[K0];
#pragma unroll
    for (int k = 0; k < K0; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;

    // this thread's two x rows: n = n0 + (tid >> 4) + 32 v -> (tap_t, c)
    long long xoff[2];
    int xshift[2];
    bool xon[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const int n = n0 + (tid >> 4) + 32 * v;
        xon[v] = n < p.N;
        const int tap_t = xon[v] ? n / p.Cr : 0, c = xon[v] ? n - tap_t * p.Cr : 0;
        xoff[v] = (long long)c * p.L;
        xshift[v] = (tap_t - pad_t) * p.shift - p.pad_w;
    }

    const int c_begin = split * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > p.nchunks) c_end = p.nchunks;
    const int nst = c_end - c_begin;                                 // stages of this workgroup (>= 1)
    auto clampc = [&](int c) { return c < nst - 1 ? c : nst - 1; };  // (redundant reloads past the end keep phases branch-free)

    f32x4 dyv[4];
    float xv[2][NXU][2];
    auto issue_loads = [&](int c) __attribute__((always_inline)) {
        const int ch = c_begin + c;
        const int b = ch / p.nT;
        const int tt0 = (ch - b * p.nT) * SAT_DW_TT;
        // dy: 64 rows x 128 positions = 2048 float4, four per thread: 32 threads per row (L % 4 == 0: a float4 is all in or all out)
        const float* sdy = p.dy + (size_t)b * p.M * p.L;
        const int c4 = (tid & 31) * 4;
        const bool t_ok = tt0 + c4 < p.L;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int m = m0 + (tid >> 5) + u * 16;
            const bool ok = t_ok && m < p.M;
            const f32x4 q = *reinterpret_cast<const f32x4*>(sdy + (size_t)(ok ? m : 0) * p.L + (ok ? tt0 + c4 : 0));
            dyv[u] = ok ? q : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const float* sx = p.x + (size_t)b * p.Cr * p.L;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const float* s = sx + xoff[v];
            const int th0 = tt0 + xshift[v];
#pragma unroll
            for (int u = 0; u < NXU; ++u) {
                const int pi = (tid & 15) + 16 * u;
                const int t = th0 + 2 * pi;
                const bool ok = pi < SAT_DW_ROW / 2 && xon[v];
                if constexpr (PAIR) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    const bool in = ok && t >= 0 && t + 1 < p.L;    // t even, L even: a pair is all in or all out
                    const f2 q = *reinterpret_cast<const f2*>(s + (in ? t : 0));
                    xv[v][u][0] = in ? q[0] : 0.0f;
                    xv[v][u][1] = in ? q[1] : 0.0f;
                } else {
                    const bool in0 = ok && t >= 0 && t < p.L, in1 = ok && t + 1 >= 0 && t + 1 < p.L;
                    xv[v][u][0] = in0 ? s[in0 ? t : 0] : 0.0f;
                    xv[v][u][1] = in1 ? s[in1 ? t + 1 : 0] : 0.0f;
                }
            }
        }
    };
    auto write_lds = [&](auto buf_c) __attribute__((always_inline)) {
        auto& y_lds = sat_pick<decltype(buf_c)::value>(y_lds0, y_lds1);
        auto& x_lds = sat_pick<decltype(buf_c)::value>(x_lds0, x_lds1);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = (tid >> 5) + u * 16, col = (tid & 31) * 4;
            uint32_t h0, h1, l0, l1;
            sat_split2_pk(dyv[u][0], dyv[u][1], &h0, &l0);
            sat_split2_pk(dyv[u][2], dyv[u][3], &h1, &l1);
            typedef uint32_t u2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u2*>(&y_lds[0][row][col]) = u2{h0, h1};
            *reinterpret_cast<u2*>(&y_lds[1][row][col]) = u2{l0, l1};
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int row = (tid >> 4) + 32 * v;
#pragma unroll
            for (int u = 0; u < NXU; ++u) {
                const int pi = (tid & 15) + 16 * u;
                if (pi < SAT_DW_ROW / 2) {
                    uint32_t h, l;
                    sat_split2_pk(xv[v][u][0], xv[v][u][1], &h, &l);
                    *reinterpret_cast<uint32_t*>(&x_lds[0][row][2 * pi]) = h;
                    *reinterpret_cast<uint32_t*>(&x_lds[1][row][2 * pi]) = l;
                }
            }
        }
    };
    const bool wave_on = (m0 + m_w < p.M) && (n0 + n_w < p.N);
    // the MFMAs of one stage for taps KB .. KB + NK - 1 (compile-time: each tap's fragment is a register selection + v_alignbit of the
    // two aligned 16-byte chunks covering positions [t, t + 16))
    auto mfma_taps = [&](auto buf_c, auto kb_c, auto nk_c) __attribute__((always_inline)) {
        constexpr int KB = decltype(kb_c)::value, NK = decltype(nk_c)::value;
        auto& y_lds = sat_pick<decltype(buf_c)::value>(y_lds0, y_lds1);
        auto& x_lds = sat_pick<decltype(buf_c)::value>(x_lds0, x_lds1);
#pragma unroll 2
        for (int ks = 0; ks < SAT_DW_TT / 16; ++ks) {
            const int tb = 16 * ks + 8 * hi;
            bf16x8 af[2];
            af[0] = *reinterpret_cast<const bf16x8*>(&y_lds[0][m_w + l31][tb]);
            af[1] = *reinterpret_cast<const bf16x8*>(&y_lds[1][m_w + l31][tb]);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                u32x4 cw[2];
#if !defined(SAT_HIPEMU)
                asm volatile("" ::: "memory");
#endif
#pragma unroll
                for (int j = 0; j < 2; ++j) cw[j] = *reinterpret_cast<const u32x4*>(&x_lds[pl][n_w + l31][tb + 8 * j]);
#pragma unroll
                for (int kk = 0; kk < NK; ++kk) {
                    constexpr int dummy = 0; (void)dummy;
                    const int k = KB + kk;
                    const int wbase = (k >> 3) * 4 + ((k & 7) >> 1);
                    const bool odd = (k & 1) != 0;
                    u32x4 r;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int w0 = wbase + i, w1 = wbase + i + 1;
                        const unsigned a0 = cw[w0 >> 2][w0 & 3];
                        if (odd) r[i] = sat_dw_alignbit(cw[(w1 >> 2) & 1][w1 & 3], a0, 16);
                        else r[i] = a0;
                    }
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, r);
                    acc[kk] = sat_mfma_32x32x16_bf16(af[0], bf, acc[kk]);
                    if (pl == 0) acc[kk] = sat_mfma_32x32x16_bf16(af[1], bf, acc[kk]);
                }
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    auto mfma_phase = [&](auto buf_c) __attribute__((always_inline)) {
        if (!wave_on) return;
        if (first) mfma_taps(buf_c, std::integral_constant<int, 0>{}, std::integral_constant<int, K0>{});
        else if constexpr (KW > K0) mfma_taps(buf_c, std::integral_constant<int, K0>{}, std::integral_constant<int, KW - K0>{});
    };
    auto phase = [&](auto bc, auto bn, int next) __attribute__((always_inline)) {
        if (!first) {
            write_lds(bn);
            issue_loads(next);
        }
        mfma_phase(bc);
        if (first) {
            write_lds(bn);
            issue_loads(next);
        }
    };

    // prologue: stage 0 into buffer 0, stage 1's data in flight in the registers
    issue_loads(0);
    write_lds(I0{});
    issue_loads(clampc(1));
    __syncthreads();
    int c = 0;
    for (; c + 2 < nst; c += 2) {
        phase(I0{}, I1{}, c + 2);                          // stage c out of buffer 0; stage c+1 -> buffer 1
        __syncthreads();
        phase(I1{}, I0{}, clampc(c + 3));                  // stage c+1 out of buffer 1; stage c+2 -> buffer 0
        __syncthreads();
    }
    if (c + 1 < nst) {                                     // stage c in buffer 0, stage c+1 in the registers
        if (!first) write_lds(I1{});
        mfma_phase(I0{});
        if (first) write_lds(I1{});
        __syncthreads();
        mfma_phase(I1{});
    } else {
        mfma_phase(I0{});
    }

    // partial slab of this split: [kw][m_pad][n_pad], n contiguous (an accumulator row stores 128 contiguous bytes)
    float* ob = p.out + (size_t)split * KW * p.m_pad * p.n_pad;
    const int n = n0 + n_w + l31;
    const int kb = first ? 0 : K0, nk = first ? K0 : KW - K0;
#pragma unroll
    for (int kk = 0; kk < K0; ++kk)
        if (kk < nk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + m_w + (r & 3) + 8 * (r >> 2) + 4 * hi;
                ob[((size_t)(kb + kk) * p.m_pad + m) * p.n_pad + n] = acc[kk][r];
            }
        }
}

struct SatDiscWgPlan { int nsplit, cps, nchunks, nT; };
static void sat_disc_wg_plan(int B, int M, int N, int L, SatDiscWgPlan* pl) {
    pl->nT = sat_cdiv(L, SAT_DW_TT);
    pl->nchunks = B * pl->nT;
    const int tiles = sat_cdiv(M, SAT_DC_CO) * sat_cdiv(N, SAT_DW_NI);
    int want = 512 / tiles;                        // one workgroup per CU (LDS): at most two FULL rounds of the 256 CUs (a 513th
    if (want > pl->nchunks) want = pl->nchunks;    // workgroup would be a third round on its own)
    if (want < 1) want = 1;
    pl->cps = sat_cdiv(pl->nchunks, want);
    pl->nsplit = sat_cdiv(pl->nchunks, pl->cps);
}
extern "C" int sat_disc_wgrad_nsplit(int B, int M, int Cin, int kh, int frames, int W) {
    int P, L, lead, rows;
    if (B <= 0 || M <= 0 || Cin <= 0 || kh < 1 || sat_disc_geom(frames, W, &P, &L, &lead, &rows)) return -1;
    SatDiscWgPlan pl;
    sat_disc_wg_plan(B, M, kh * Cin, L, &pl);
    return pl.nsplit;
}
// dW of sat_disc_conv: dy (B, M, L) = dL/d(pre-activation) (pitched, zeros at pad positions), x (B, Cin, L) the layer's input (pitched).
// Writes nsplit slabs [kw][ceil64(M)][ceil64(kh * Cin)] (virtual channel n = tap_t * Cin + c); sum them with sat_reduce_splits.
extern "C" int sat_disc_wgrad(const float* dy, const float* x, float* partial, int B, int M, int Cin, int frames, int W, int kh, int kw,
                              int dil_t, void* stream) {
    int P, L, lead, rows;
    if (B <= 0 || M <= 0 || Cin <= 0 || sat_disc_geom(frames, W, &P, &L, &lead, &rows)) { sat_set_error("sat_disc_wgrad: bad shape"); return 1; }
    if (kh < 1 || !(kh & 1) || !(kw == 9 || kw == 3) || dil_t < 1) { sat_set_error("sat_disc_wgrad: odd kh, kw 9 or 3"); return 1; }
    if ((((uintptr_t)dy | (uintptr_t)x) & 15) || !partial) { sat_set_error("sat_disc_wgrad: missing / misaligned operand"); return 1; }
    SatDiscWgPlan pl;
    const int N = kh * Cin;
    sat_disc_wg_plan(B, M, N, L, &pl);
    SatDiscWgParams p{dy, x, partial, B, M, Cin, N, L, kh, dil_t * P, (kw - 1) / 2,
                      sat_cdiv(M, SAT_DC_CO) * SAT_DC_CO, sat_cdiv(N, SAT_DW_NI) * SAT_DW_NI, pl.cps, pl.nchunks, pl.nT};
    dim3 grid(sat_cdiv(M, SAT_DC_CO), sat_cdiv(N, SAT_DW_NI), pl.nsplit);
    if (kw == 9) SAT_LAUNCH(sat_disc_wgrad_kernel<9>, grid, dim3(SAT_DW_NT), stream, p);
    else SAT_LAUNCH(sat_disc_wgrad_kernel<3>, grid, dim3(SAT_DW_NT), stream, p);
    return sat_check_launch("sat_disc_wgrad");
}
