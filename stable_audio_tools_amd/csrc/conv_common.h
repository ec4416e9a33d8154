// conv_common.h — parameter blocks shared by the Oobleck conv kernels (conv1d.hip, convtr1d.hip,
// conv_wgrad.hip).  All tensors are fp32, (B, C, T) with T contiguous — the reference's layout
// (stable_audio_tools/models/autoencoders.py:285-362 operates on (B, C, T) nn.Conv1d tensors).
#pragma once
#include "sat_device.h"

// Implicit-GEMM tile: 128 output channels x 128 output time steps per 256-thread workgroup;
// each of the 4 waves owns a 64(co) x 64(t) sub-tile = 2x2 MFMA 32x32x2 f32 accumulators.
#define SAT_CO_T 128
#define SAT_T_T 128
#define SAT_W_ROWS 64      // max (ci, tap) rows of the weight slab staged per K-chunk
#define SAT_A_FLOATS 4352  // activation slab capacity (floats)

struct SatConvParams {
    const float* x;       // (B, Cin, Tin)   conv input (pre-activation)
    const float* w;       // packed weights, see sat_amd.h (layout depends on kernel)
    const float* bias;    // (Cout) or null
    const float* alpha;   // (Cin) SnakeBeta log-alpha applied to x while staging, or null
    const float* beta;    // (Cin)
    const float* res;     // (B, Cout, Tout) added in the epilogue, or null
    float* y;             // (B, Cout, Tout)
    // backward epilogue (dgrad of a conv whose *input* was snake(x2)):
    //   y = acc * dsnake(x2) + res ; per-tile partial sums of d/dlog-alpha, d/dlog-beta
    const float* x2;      // (B, Cout, Tout) or null
    const float* alpha2;  // (Cout)
    const float* beta2;   // (Cout)
    float* part_da;       // [B * gridDim.x][Cout]
    float* part_db;
    int B, Cin, Cout, Tin, Tout;
    int K, stride, dil, pad;
    int tanh_out;
};

// d snake / dx, d/dlog-alpha, d/dlog-beta for act = x + sin^2(a x) / (b + 1e-9), a = e^la, b = e^lb
// (reference forward: models/blocks.py:291-292, :321-329).
struct SatSnakeGrad {
    float dx, dla, dlb;
};
SAT_DEVICE SatSnakeGrad sat_snake_grad(float x, float a, float b) {
    const float ib = 1.0f / (b + 1e-9f);
    float s2, c2;
    sat_sincos2(a * x, &s2, &c2);                 // sin(2 a x), cos(2 a x)
    const float sq = fmaf(-0.5f, c2, 0.5f);       // sin^2(a x)
    SatSnakeGrad g;
    g.dx = 1.0f + a * ib * s2;
    g.dla = x * a * ib * s2;
    g.dlb = -sq * ib * ib * b;
    return g;
}
