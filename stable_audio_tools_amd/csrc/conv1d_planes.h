// conv1d_planes.h — the activation PLANES the k = 7 conv kernel (conv1d_bf16x3_k7q.h) and the fused ResidualUnit read.  Included by
// conv1d_bf16x3.hip.
//
// The direct kernel (conv1d_bf16x3_k7.h) applies SnakeBeta and the bf16 hi / lo split while staging, per workgroup: at C = 128 that VALU +
// ds_write work costs as much as the chunk's MFMAs, and every one of the Cout / 128 channel tiles repeats it on the same input.  Here the
// activation is converted ONCE into two bf16 planes laid out [B][Cin/8][rows][8 channels] (row = 32 + t, zero rows around the sequence:
// snake(0) = 0) — by sat_k7_planes_kernel below, or (53 of the 70 k7 inputs of the stack) by the producing conv's epilogue ("plane
// emission", conv1d_bf16x3.hip) — and the consumer's K loop is LDS-DMA + matrix work only.  (Round 2's first consumer of this layout,
// the persistent 8-tap-group kernel conv1d_bf16x3_k7p.h, was retired in round 5: k7q serves every plan it served.)
#pragma once

#define SAT_K7P_LEAD 32           // zero rows before t = 0 in a plane (>= pad)
static_assert(SAT_K7P_LEAD == SAT_K7P_LEAD_ROWS, "plane emission (conv1d_bf16x3.hip) writes row 32 + t");

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
