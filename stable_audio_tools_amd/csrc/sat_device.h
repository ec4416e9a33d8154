// sat_device.h — common device-side vocabulary for the gfx950 (MI355X / CDNA4) kernels.
//
// The kernels are written for wave64 + MFMA + LDS directly; there is no other backend.  The only
// alternative consumer of these sources is the host-side *simulator* used by the CPU test-suite
// (tests/emu/, -DSAT_HIPEMU), which executes the same kernel bodies with fibers so that indexing
// and fragment layouts can be checked against the oracle without a GPU.  It is never shipped.
#pragma once

#if defined(SAT_HIPEMU)
#include "hipemu.h"
#define SAT_DEVICE static inline
#else
#include <hip/hip_runtime.h>
#define SAT_DEVICE __device__ __forceinline__
#endif

#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

#define SAT_WAVE 64

// ---------------------------------------------------------------------------------------------
// MFMA wrappers.  Fragment layouts (cdna_hip_programming.md §3):
//   32x32x2 f32 : A lane l -> A[i=l&31][k=l>>5]; B lane l -> B[k=l>>5][j=l&31];
//                 D reg r  -> D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]
//   32x32x16 bf16: A lane l -> A[i=l&31][k=8*(l>>5)+e], e=0..7; B lane l -> B[k=8*(l>>5)+e][j=l&31];
//                 D as above.
//   16x16x32 bf16: A lane l -> A[i=l&15][k=8*(l>>4)+e]; B lane l -> B[k=8*(l>>4)+e][j=l&15];
//                 D reg r -> D[row=4*(l>>4)+r][col=l&15]
// ---------------------------------------------------------------------------------------------
#if defined(SAT_HIPEMU)
static inline float hipemu_bf16_to_f32(short s) {
    uint32_t u = ((uint32_t)(uint16_t)s) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline f32x16 sat_mfma_32x32x2_f32(float a, float b, f32x16 c) {
    struct { float a, b; } mine = {a, b};
    const char* all = hipemu::wave_exchange(&mine, sizeof(mine));
    const int l = hipemu::lane_id();
    const int col = l & 31, hi = l >> 5;
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            memcpy(&av, all + (size_t)(row + 32 * k) * hipemu::kSlotBytes, 4);
            memcpy(&bv, all + (size_t)(col + 32 * k) * hipemu::kSlotBytes + 4, 4);
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}
static inline f32x16 sat_mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    struct { bf16x8 a, b; } mine = {a, b};
    const char* all = hipemu::wave_exchange(&mine, sizeof(mine));
    const int l = hipemu::lane_id();
    const int col = l & 31, hi = l >> 5;
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int g = 0; g < 2; ++g) {
            bf16x8 av, bv;
            memcpy(&av, all + (size_t)(row + 32 * g) * hipemu::kSlotBytes, 16);
            memcpy(&bv, all + (size_t)(col + 32 * g) * hipemu::kSlotBytes + 16, 16);
            for (int e = 0; e < 8; ++e) acc += hipemu_bf16_to_f32(av[e]) * hipemu_bf16_to_f32(bv[e]);
        }
        d[r] = acc;
    }
    return d;
}
static inline f32x4 sat_mfma_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    struct { bf16x8 a, b; } mine = {a, b};
    const char* all = hipemu::wave_exchange(&mine, sizeof(mine));
    const int l = hipemu::lane_id();
    const int col = l & 15, grp = l >> 4;
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * grp + r;
        float acc = c[r];
        for (int g = 0; g < 4; ++g) {
            bf16x8 av, bv;
            memcpy(&av, all + (size_t)(row + 16 * g) * hipemu::kSlotBytes, 16);
            memcpy(&bv, all + (size_t)(col + 16 * g) * hipemu::kSlotBytes + 16, 16);
            for (int e = 0; e < 8; ++e) acc += hipemu_bf16_to_f32(av[e]) * hipemu_bf16_to_f32(bv[e]);
        }
        d[r] = acc;
    }
    return d;
}
// OCP fp8 e4m3fn (gfx950's fp8; no infinities, 0x7f / 0xff = NaN, max 448)
static inline float hipemu_e4m3_to_f32(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f;
    if (e == 15 && m == 7) f = NAN;
    else if (e == 0) f = ldexpf((float)m, -9);
    else f = ldexpf((float)(8 + m), e - 10);
    return s ? -f : f;
}
// 32x32x64 fp8 x fp8 with unit block scales: lane l holds A[i = l & 31][k = 32 (l >> 5) + e], e = 0..31 (byte e of the 8 dwords),
// B[k = 32 (l >> 5) + e][j = l & 31]; D as the other 32x32 shapes.
static inline f32x16 sat_mfma_32x32x64_fp8(i32x8 a, i32x8 b, f32x16 c) {
    struct { i32x8 a, b; } mine = {a, b};
    const char* all = hipemu::wave_exchange(&mine, sizeof(mine));
    const int l = hipemu::lane_id();
    const int col = l & 31, hi = l >> 5;
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        for (int g = 0; g < 2; ++g) {
            uint8_t av[32], bv[32];
            memcpy(av, all + (size_t)(row + 32 * g) * hipemu::kSlotBytes, 32);
            memcpy(bv, all + (size_t)(col + 32 * g) * hipemu::kSlotBytes + 32, 32);
            for (int e = 0; e < 32; ++e) acc += hipemu_e4m3_to_f32(av[e]) * hipemu_e4m3_to_f32(bv[e]);
        }
        d[r] = acc;
    }
    return d;
}
#else
// v_mfma_scale_f32_32x32x64_f8f6f4 with both formats e4m3 (cbsz = blgp = 0) and unit E8M0 block scales (127 = 2^0): the MX
// instruction is the only fp8 MFMA that runs at twice the bf16 rate on gfx950 (cdna_hip_programming.md section 3)
SAT_DEVICE f32x16 sat_mfma_32x32x64_fp8(i32x8 a, i32x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
}
SAT_DEVICE f32x16 sat_mfma_32x32x2_f32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
SAT_DEVICE f32x16 sat_mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
SAT_DEVICE f32x4 sat_mfma_16x16x32_bf16(bf16x8 a, bf16x8 b, f32x4 c) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
#endif

// four fp32 -> four OCP e4m3 bytes (saturating at 448, round to nearest even): gemm.hip's quantisers, dit_ops.hip's LayerNorm -> fp8
SAT_DEVICE uint32_t sat_f32x4_to_fp8(float a, float b, float c, float d) {
#if defined(SAT_HIPEMU)
    auto enc = [](float x) -> uint32_t {
        if (x != x) return 0x7fu;
        const uint32_t sign = x < 0.f ? 0x80u : 0u;
        float ax = fabsf(x);
        if (ax > 448.f) ax = 448.f;
        if (ax < ldexpf(1.0f, -10)) return sign;                                  // below half the smallest subnormal (2^-9)
        int e;
        frexpf(ax, &e);                                                           // ax = f * 2^e, f in [0.5, 1)
        int ue = e - 1;                                                           // unbiased exponent: ax in [2^ue, 2^(ue+1))
        if (ue < -6) ue = -6;                                                     // subnormal range shares the exponent of 2^-6
        const float q = ldexpf(1.0f, ue - 3);                                     // spacing of representable values
        float m = nearbyintf(ax / q);                                             // RNE (default rounding mode)
        float v = m * q;
        if (v > 448.f) v = 448.f;
        if (v == 0.f) return sign;
        frexpf(v, &e);
        ue = e - 1;
        uint32_t bits;
        if (ue < -6) bits = (uint32_t)nearbyintf(v / ldexpf(1.0f, -9));           // subnormal: mantissa only
        else bits = ((uint32_t)(ue + 7) << 3) | ((uint32_t)nearbyintf(v / ldexpf(1.0f, ue - 3)) - 8u);
        return sign | bits;
    };
    return enc(a) | (enc(b) << 8) | (enc(c) << 16) | (enc(d) << 24);
#else
    const float lim = 448.0f;
    a = fminf(fmaxf(a, -lim), lim); b = fminf(fmaxf(b, -lim), lim); c = fminf(fmaxf(c, -lim), lim); d = fminf(fmaxf(d, -lim), lim);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
#endif
}

// round-to-nearest-even fp32 -> bf16 bits
SAT_DEVICE short sat_f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (short)0x7fc0;  // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (short)(u >> 16);
}
SAT_DEVICE float sat_bf16_to_f32(short s) {
    uint32_t u = ((uint32_t)(uint16_t)s) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// hi/lo bf16 split of two fp32 values: x = hi + lo with |x - hi - lo| <= 2^-17 |x|.  Returns the two hi (resp. lo)
// bf16 patterns packed as (b << 16) | a.  gfx950 has a packed RNE convert (v_cvt_pk_bf16_f32): 5 VALU ops per pair.
SAT_DEVICE void sat_split2_pk(float a, float b, uint32_t* hi, uint32_t* lo) {
#if defined(SAT_HIPEMU)
    const short ha = sat_f32_to_bf16(a), hb = sat_f32_to_bf16(b);
    const short la = sat_f32_to_bf16(a - sat_bf16_to_f32(ha)), lb = sat_f32_to_bf16(b - sat_bf16_to_f32(hb));
    *hi = ((uint32_t)(uint16_t)hb << 16) | (uint16_t)ha;
    *lo = ((uint32_t)(uint16_t)lb << 16) | (uint16_t)la;
#else
    typedef __bf16 sat_bf2 __attribute__((ext_vector_type(2)));
    typedef float sat_f2 __attribute__((ext_vector_type(2)));
    const sat_f2 v = {a, b};
    const sat_bf2 h = __builtin_convertvector(v, sat_bf2);
    const sat_f2 d = v - __builtin_convertvector(h, sat_f2);
    const sat_bf2 l = __builtin_convertvector(d, sat_bf2);
    *hi = __builtin_bit_cast(uint32_t, h);
    *lo = __builtin_bit_cast(uint32_t, l);
#endif
}

// two fp32 -> packed bf16 (RNE), (b << 16) | a: one v_cvt_pk_bf16_f32
SAT_DEVICE uint32_t sat_cvt2_pk(float a, float b) {
#if defined(SAT_HIPEMU)
    return ((uint32_t)(uint16_t)sat_f32_to_bf16(b) << 16) | (uint16_t)sat_f32_to_bf16(a);
#else
    typedef __bf16 sat_bf2 __attribute__((ext_vector_type(2)));
    typedef float sat_f2 __attribute__((ext_vector_type(2)));
    const sat_f2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, sat_bf2));
#endif
}

// a value known to be identical in every lane of the wave -> scalar register (lets the compiler use s_load for
// addresses derived from it)
#if defined(SAT_HIPEMU)
#define SAT_UNIFORM(x) (x)
#else
#define SAT_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

// XCD-aware tile order.  Workgroups are dealt round-robin (in linear id order) to the 8 XCDs, each with its own L2, so the
// `nshare` tiles that read the same operand slab (the channel tiles of one activation window, the (co, ci) tiles of one
// split-K time range) get linear ids that differ by multiples of 8: same XCD, dispatched together, one HBM fetch.
// L = linear workgroup id, nouter = number of slabs; returns the tile index within the slab's group and the slab index.
// (Placement is a speed heuristic only — nothing depends on it for correctness.)
SAT_DEVICE void sat_xcd_tile(int L, int nshare, int nouter, int* share_idx, int* outer_idx) {
    const int ngroups = nouter >> 3;                  // full groups of 8 slabs
    const int per_group = 8 * nshare;
    if (L < ngroups * per_group) {
        const int grp = L / per_group, rem = L - grp * per_group;
        *share_idx = rem >> 3;
        *outer_idx = grp * 8 + (rem & 7);
    } else {                                          // the last < 8 slabs: natural order
        const int R = L - ngroups * per_group;
        *share_idx = R % nshare;
        *outer_idx = ngroups * 8 + R / nshare;
    }
}

// compile-time pick of one of two objects (the two halves of a double buffer are separate __shared__ arrays so that the
// compiler knows reads of one and writes of the other never alias)
template <int B, class T>
SAT_DEVICE T& sat_pick(T& a, T& b) {
    if constexpr (B == 0) return a;
    else return b;
}

// wave64 all-lane sum
SAT_DEVICE float sat_wave_sum(float v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
// sum inside each 32-lane half
SAT_DEVICE float sat_half_sum(float v) {
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// sin(2y) and cos(2y) on the hardware transcendental unit: v_sin_f32 / v_cos_f32 take their argument in revolutions,
// so the angle is y/pi reduced to its fraction.  The product y * (1/pi) is carried in two pieces (the fma recovers its
// rounding error, plus the second word of 1/pi) so that the reduction stays exact to ~2^-24 of a revolution for
// |y| < 1e4 — measured error of the pair on gfx950: < 4e-7 absolute.  7 VALU issues (2 of them transcendental) against
// ~25 for the polynomial sat_sincos; this is the SnakeBeta prologue / gradient epilogue of every conv.
SAT_DEVICE void sat_sincos2(float y, float* s2y, float* c2y) {
    const float hi = y * 0.31830987334251404f;                       // fl(1/pi)
    float lo = fmaf(y, 0.31830987334251404f, -hi);
    lo = fmaf(y, 1.2841276486e-08f, lo);                             // 1/pi - fl(1/pi)
#if defined(SAT_HIPEMU)
    const float f = (hi - floorf(hi)) + lo;
    *s2y = sinf(6.283185307179586f * f);
    *c2y = cosf(6.283185307179586f * f);
#else
    const float f = __builtin_amdgcn_fractf(hi) + lo;
    *s2y = __builtin_amdgcn_sinf(f);
    *c2y = __builtin_amdgcn_cosf(f);
#endif
}
// sin^2(y) = (1 - cos 2y) / 2
SAT_DEVICE float sat_sin2(float y) {
    float s, c;
    sat_sincos2(y, &s, &c);
    return fmaf(-0.5f, c, 0.5f);
}

// SnakeBeta activation (reference: stable_audio_tools/models/blocks.py:291-292, :321-329).
// a = exp(alpha_log), ib = 1/(exp(beta_log) + 1e-9) are prepared once per channel by the caller.
SAT_DEVICE float sat_snake(float x, float a, float ib) {
    return fmaf(ib, sat_sin2(x * a), x);
}

// ---------------------------------------------------------------------------------------------
// Launch + status plumbing shared by every C-ABI entry point.
// ---- LDS-DMA staging, counted waits, scheduling hints (gemm.hip, attention_fwd64.h) ----
#if defined(SAT_HIPEMU)
static inline void sat_glds16(const void* g, void* lds_wave_base) { memcpy((char*)lds_wave_base + 16 * hipemu::lane_id(), g, 16); }
static inline void sat_glds4(const void* g, void* lds_wave_base) { memcpy((char*)lds_wave_base + 4 * hipemu::lane_id(), g, 4); }
#define SAT_WAIT_VMCNT(n)
#define SAT_RAW_BARRIER() hipemu::block_barrier()
#define SAT_WAIT_LGKM0()
#define SAT_SCHED_FENCE()
static inline void sat_wave_sync() { int z = 0; (void)hipemu::wave_exchange(&z, sizeof(z)); }
#define SAT_SETPRIO(x)
#define SAT_SCHED_GROUP(mask, n)
static inline bool sat_wave_any(bool v) {
    int x = v ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) x |= __shfl_xor(x, m);
    return x != 0;
}
#else
// LDS destination = wave-uniform base + lane * 16 (cdna_hip_programming.md §5)
SAT_DEVICE void sat_glds16(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// 4 bytes per lane (LDS destination = wave-uniform base + lane * 4): the L2 "touch" prefetch of gemm.hip
SAT_DEVICE void sat_glds4(const void* g, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}
// counted wait on this wave's LDS-DMA queue + a bare s_barrier: tiles further down the ring stay in flight across the barrier
// (__syncthreads() would drain them: an LDS-DMA is a pending LDS write on the VM counter)
#define SAT_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define SAT_RAW_BARRIER() __builtin_amdgcn_s_barrier()
#define SAT_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define SAT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
SAT_DEVICE void sat_wave_sync() { __builtin_amdgcn_wave_barrier(); }
#define SAT_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
// scheduling group: the next `n` instructions of class `mask` (0x8 MFMA, 0x2 VALU, 0x100 DS read, ...) in program order
#define SAT_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
SAT_DEVICE bool sat_wave_any(bool v) { return __builtin_amdgcn_ballot_w64(v) != 0; }
#endif

// s_setprio around the MFMA sections of the generic conv / weight-gradient kernels (conv1d_bf16x3.hip, conv_wgrad*.hip): an A/B switch.
// The attention forward gains 13 % from raising its MFMA groups' priority; these kernels do NOT: with it their own event times are
// unchanged (138.6 vs 138.2 ms of conv kernels per 4 steps) and the generator step is 8 ms SLOWER (166.1 / 166.1 vs 158.0 ms, A / B / A
// on one box, profiles/EXPERIMENTS.md round 4) — off unless built with -DSAT_WITH_MFMA_PRIO.
#if defined(SAT_WITH_MFMA_PRIO)
#define SAT_MFMA_PRIO(x) SAT_SETPRIO(x)
#else
#define SAT_MFMA_PRIO(x)
#endif

// ---------------------------------------------------------------------------------------------
void sat_set_error(const char* msg);
int sat_check_launch(const char* what);
int sat_cu_count();      // compute units of the current device (elementwise.hip)

#if defined(SAT_HIPEMU)
#define SAT_LAUNCH(kernel, grid, block, stream, params) \
    hipemu::launch((grid), (block), [=]() { kernel(params); })
#else
#define SAT_LAUNCH(kernel, grid, block, stream, params) \
    hipLaunchKernelGGL(kernel, (grid), (block), 0, (hipStream_t)(stream), params)
#endif

static inline int sat_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long sat_cdivll(long long a, long long b) { return (a + b - 1) / b; }
