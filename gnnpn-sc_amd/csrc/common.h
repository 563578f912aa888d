// Shared helpers for the gfx950 kernels of libgnnpn_hip.so.  CDNA4 only: 64-wide wavefronts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "gnnpn_hip.h"

#define GNNPN_WAVE 64

// thread-local last-error text (gnnpn_last_error)
extern thread_local char g_gnnpn_err[256];

#define GNNPN_FAIL(code, ...)                                   \
    do {                                                        \
        snprintf(g_gnnpn_err, sizeof(g_gnnpn_err), __VA_ARGS__); \
        return (code);                                          \
    } while (0)

#define GNNPN_REQUIRE(cond, ...)                          \
    do {                                                  \
        if (!(cond)) GNNPN_FAIL(GNNPN_E_ARG, __VA_ARGS__); \
    } while (0)

#define GNNPN_CHECK_LAUNCH(name)                                                         \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) GNNPN_FAIL(GNNPN_E_LAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

static inline bool gnnpn_aligned(const void* p, size_t a) { return ((uintptr_t)p & (a - 1)) == 0; }

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// wave-wide sum with a fixed butterfly order (deterministic, the same for every launch)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = __fadd_rn(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffu), off, 64);
        unsigned hi = __shfl_xor((unsigned)(v >> 32), off, 64);
        unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

// monotone map fp32 -> u32 (larger float -> larger unsigned; -0 < +0; NaNs sort above +inf)
__device__ __forceinline__ unsigned float_order_key(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// logistic and tanh used by the LSTM cells.  Written out so that every kernel (and the
// arithmetic the tests document) uses the same formula: accurate expf/tanhf from the device
// library, IEEE division; no fast-math.
__device__ __forceinline__ float sigmoid_f32(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == GNNPN_ACT_RELU) return v < 0.0f ? 0.0f : v;   // NaN stays NaN, as torch.relu
    if (act == GNNPN_ACT_SIGMOID) return sigmoid_f32(v);
    return v;
}
