// hipemu.h — host-side functional simulator for the restricted HIP dialect used in
// stable_audio_tools_amd/csrc/*.hip.
//
// TEST INFRASTRUCTURE ONLY.  The build container has no GPU; this lets the CPU test-suite execute
// the *same kernel source* (block/thread indexing, LDS tiling, barriers, wave64 shuffles, MFMA
// fragment layouts) against the oracle before a GPU box is spent on it.  It is compiled into
// tests/emu/libsat_emu.so, which only tests/ load.  The product package never loads it and has no
// CPU fallback: stable_audio_tools_amd/_lib.py loads csrc/libsat_amd.so (gfx950) or raises.
//
// Model: one OS thread; each GPU thread of a block is a fiber.  Fibers of a wave (64 lanes) run
// one after the other until they reach a wave-collective (shuffle / MFMA) or a block barrier.
// MFMA fragment layouts follow /opt/skills/guides/cdna_hip_programming.md §3 (verified on hardware
// by tests/test_gpu_mfma_layout.py).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

namespace hipemu {

struct Dim3 {
    unsigned x, y, z;
    Dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

extern Dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

// ---- scheduler entry points (hipemu.cpp) ----
void launch(Dim3 grid, Dim3 block, const std::function<void()>& body);
void block_barrier();
// Deposit `nbytes` for this lane, wait for the whole wave, return pointer to the wave's 64 slots
// (slot stride = kSlotBytes).  Valid until this lane's next-but-one collective.
constexpr int kSlotBytes = 160;
const char* wave_exchange(const void* mine, int nbytes);
int lane_id();

}  // namespace hipemu

typedef hipemu::Dim3 dim3;
typedef void* hipStream_t;
#define threadIdx hipemu::g_threadIdx
#define blockIdx hipemu::g_blockIdx
#define blockDim hipemu::g_blockDim
#define gridDim hipemu::g_gridDim

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __syncthreads() hipemu::block_barrier()

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

template <typename T>
static inline T hipemu_shfl_src(T v, int src_lane) {
    const char* all = hipemu::wave_exchange(&v, (int)sizeof(T));
    T r;
    memcpy(&r, all + (size_t)(src_lane & 63) * hipemu::kSlotBytes, sizeof(T));
    return r;
}
template <typename T> static inline T __shfl_xor(T v, int mask) { return hipemu_shfl_src(v, hipemu::lane_id() ^ mask); }
template <typename T> static inline T __shfl_down(T v, int d) {
    int s = hipemu::lane_id() + d;
    return hipemu_shfl_src(v, s > 63 ? hipemu::lane_id() : s);
}
template <typename T> static inline T __shfl(T v, int lane) { return hipemu_shfl_src(v, lane); }

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

#define hipSuccess 0
static inline int hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(int) { return "hipemu"; }
