// Shared device/host helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <time.h>
#include "../../../include/mi_convnet.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MCN_API extern "C" __attribute__((visibility("default")))

void mcn_set_error(const char* fmt, ...);
int mcn_check_hip(hipError_t e, const char* what);

#define MCN_REQUIRE(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            mcn_set_error(__VA_ARGS__);       \
            return MCN_EINVAL;                \
        }                                     \
    } while (0)

int mcn_launch_check(const char* what);  // hipGetLastError + this thread's launch counter (mcn_launch_count)
// every kernel launch of the library: the host time inside the runtime's launch call is accumulated per calling thread
// (mcn_launch_ns) so that a caller can tell submission cost from time it spent blocked (bench.py host_breakdown)
extern thread_local uint64_t t_mcnLaunchNs;
static inline uint64_t mcn_now_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}
#define MCN_LAUNCH(...)                              \
    do {                                             \
        const uint64_t mcnT0_ = mcn_now_ns();        \
        hipLaunchKernelGGL(__VA_ARGS__);             \
        t_mcnLaunchNs += mcn_now_ns() - mcnT0_;      \
    } while (0)
#define MCN_LAUNCH_CHECK(name) mcn_launch_check(name)

// A tick armed by mcn_arm_tick for this thread's next launch on `stream`: a kernel that supports it takes (flag, value) into
// its parameter block and stores value into *flag in its first instructions (MCN_TICK_PROLOGUE) -- a kernel only starts when
// everything queued before it on its stream has finished.  Any other launch leaves the tick armed and mcn_launch_check sends it
// with a one-thread kernel right behind that launch.
struct McnTick {
    int* flag;  // host-visible word (a submitting thread polls it)
    int* gate;  // optional device word (a stream gate on another stream spins on it, mcn_stream_gate)
    int value;
};
McnTick mcn_take_tick(hipStream_t stream);  // {nullptr, 0} when nothing is armed for that stream

// Workgroups are dealt to the eight XCDs round-robin in launch order, and every XCD has its own L2: two blocks that are
// neighbours in launch order NEVER share an L2.  Kernels whose blocks come in groups of T that read the same operands (the tiles
// of one K chunk, the image tiles of one module, the row / filter tiles of one module chunk) therefore decode their launch index
// L through this function: the T blocks of a group are spread over 8 T launch slots, 8 apart, so that they land on ONE XCD
// close together in time.  (Measured with FETCH_SIZE, round 3: fc6's backward-data product 26 -> 13 MB of L2 misses per launch.)
// The last nG % 8 groups keep the plain group-major order.  Bijective on [0, T * nG).
__device__ __forceinline__ void mcn_xcd_group(int L, int T, int nG, int& group, int& t) {
    const int G = 8 * T, full = (nG >> 3) * G;
    if (L < full) {
        const int g = L / G, r = L - g * G;
        group = g * 8 + (r & 7);
        t = r >> 3;
    } else {
        const int l = L - full, q = l / T;
        group = (nG >> 3) * 8 + q;
        t = l - q * T;
    }
}

#define MCN_TICK_PROLOGUE(p)                                                                          \
    if ((p).tick.flag && (blockIdx.x | blockIdx.y | blockIdx.z) == 0 && threadIdx.x == 0) {           \
        if ((p).tick.gate) *(volatile int*)(p).tick.gate = (p).tick.value;                            \
        *(volatile int*)(p).tick.flag = (p).tick.value;                                               \
    }

static inline int divup(int a, int b) { return (a + b - 1) / b; }
static inline size_t divup_sz(size_t a, size_t b) { return (a + b - 1) / b; }

// v_mfma_f32_16x16x4_f32: lane l holds A[row=l&15][k=l>>4], B[k=l>>4][col=l&15];
// acc[r] = D[row=(l>>4)*4+r][col=l&15].  Exact fp32 (k-ordered fma chain).
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Neuron functors (neuron.cuh:64-456), shared by the fused epilogues and the
// stand-alone neuron kernels.  Exp/log/div are IEEE-accurate library versions;
// the reference's fast intrinsics differ within the stated fp32 tolerance.
__device__ __forceinline__ float neuron_apply(int type, float x, float a, float b) {
    switch (type) {
        case MCN_NEURON_RELU: return x < 0.f ? 0.f : x;
        case MCN_NEURON_LOGISTIC: return 1.f / (1.f + expf(-x));
        case MCN_NEURON_BRELU: return x < 0.f ? 0.f : (x > a ? a : x);
        case MCN_NEURON_ABS: return fabsf(x);
        case MCN_NEURON_TANH: return a * (2.f / (1.f + expf(x * (-2.f * b))) - 1.f);
        case MCN_NEURON_SOFTRELU: return x > 4.f ? x : logf(1.f + expf(x));
        case MCN_NEURON_SQUARE: return x * x;
        case MCN_NEURON_SQRT: return sqrtf(x);
        case MCN_NEURON_LINEAR: return a * x + b;
        default: return x;
    }
}

// gradient wrt the neuron input given dE/dy (g), the output y and the input x
__device__ __forceinline__ float neuron_grad(int type, float g, float y, float x, float a, float b) {
    switch (type) {
        case MCN_NEURON_RELU: return g * (y > 0.f ? 1.f : 0.f);
        case MCN_NEURON_LOGISTIC: return g * y * (1.f - y);
        case MCN_NEURON_BRELU: return g * (y > 0.f ? 1.f : 0.f) * (y < a ? 1.f : 0.f);
        case MCN_NEURON_ABS: return g * (x > 0.f ? 1.f : -1.f);
        case MCN_NEURON_TANH: { const float t = (1.f - y / a) / 2.f; return g * (-4.f * a * b) * (t * (t - 1.f)); }
        case MCN_NEURON_SOFTRELU: { if (y > 4.f) return g; const float e = expf(y); return g * (e / (1.f + e)); }
        case MCN_NEURON_SQUARE: return g * 2.f * x;
        case MCN_NEURON_SQRT: return g / (2.f * y);
        case MCN_NEURON_LINEAR: return g * a;
        default: return g;
    }
}

static inline bool neuron_valid(int t) { return t >= 0 && t <= 9; }
