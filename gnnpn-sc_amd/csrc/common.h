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
unsigned gnnpn_front_lds_pad(const void* func);   // api.hip: dynamic LDS bytes up to the thread's gnnpn_lds_footprint_kb (0: none)

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

// draw `ctr` of the counter-based uniform stream of `seed` (splitmix64 finaliser, the stream of oracle/woa.py), as a
// 24-bit uniform in [0, 1) — exactly representable in fp32
__device__ __forceinline__ float stream_uniform24(unsigned long long seed, unsigned long long ctr) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (ctr + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(z >> 40) * 5.9604644775390625e-08f;
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

// ---- activations of the LSTM cells ---------------------------------------------------------------
// The recurrent kernels run one wave per SIMD, so every VALU instruction of the cell update is on
// the per-step critical path (the device-library expf/tanhf sequences cost ~1 us of a 5.6 us step).
// These forms use the hardware v_exp_f32 / v_rcp_f32 (1 ulp each) with the argument product carried
// in two parts and one Newton step on the reciprocal: |error| <= ~1.5e-7 absolute, i.e. the rounding
// of the result itself (tests/test_gpu_ops.py::test_cell_activations measures it against torch CPU).
// ALL recurrent kernels (streaming and cooperative, encoder and decoder) use exactly these, which
// is what keeps the implementations bit-identical to each other.
__device__ __forceinline__ float cell_exp(float x) {        // e^x for x in [-87, 87]
    const float L2E_HI = 1.44269502e+00f, L2E_LO = 1.92596303e-08f;
    const float t = __fmul_rn(x, L2E_HI);
    float r = fmaf(x, L2E_HI, -t);                           // exact low part of the product
    r = fmaf(x, L2E_LO, r);
    const float e = __builtin_amdgcn_exp2f(t);               // v_exp_f32
    return fmaf(e, __fmul_rn(r, 0.693147182f), e);           // e * 2^r
}
__device__ __forceinline__ float cell_rcp(float d) {        // 1/d, d finite and >= 1
    const float r = __builtin_amdgcn_rcpf(d);                // v_rcp_f32
    return fmaf(fmaf(-d, r, 1.0f), r, r);
}
// One activation, branch-free in `is_tanh` (per-lane selectable): the sigmoid and the tanh share the
// exp/reciprocal core, the tanh-only parts (odd polynomial below |x| = 0.25, 1 - 2r, sign) are a few
// extra VALU instructions selected by v_cndmask — no divergent branch when half a wave wants tanh
// and the other half sigmoid (the [g | o] MFMA tile of the cooperative kernels).
__device__ __forceinline__ float cell_act(float x, bool is_tanh) {
    const float ax = fminf(fabsf(x), 43.0f);
    float ut = __fmul_rn(2.0f, ax), us = fminf(fmaxf(-x, -87.0f), 87.0f);
    asm volatile("" : "+v"(ut), "+v"(us));                   // both sides computed, then ONE v_cndmask (no exec-mask dance)
    const float u = is_tanh ? ut : us;
    const float r = cell_rcp(__fadd_rn(1.0f, cell_exp(u)));   // sigmoid(x), or 1/(e^{2|x|}+1)
    // tanh: |x| >= 0.25: 1 - 2/(e^{2|x|}+1);  below: odd Taylor polynomial (truncation < 3e-9 relative)
    const float big = __fsub_rn(1.0f, __fmul_rn(2.0f, r));
    const float x2 = __fmul_rn(ax, ax);
    float p = fmaf(x2, 2.18694885e-02f, -5.39682540e-02f);   // 62/2835, -17/315
    p = fmaf(x2, p, 1.33333333e-01f);                        // 2/15
    p = fmaf(x2, p, -3.33333333e-01f);                       // -1/3
    const float small = fmaf(__fmul_rn(ax, x2), p, ax);
    const float th = copysignf(ax < 0.25f ? small : big, x);
    return is_tanh ? th : r;
}
// The same activation on TWO values at once: every fmul/fma/fadd of the scalar form becomes one packed
// instruction (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two results per issue slot); min/max/select and the
// two transcendentals stay per component.  Operation for operation the same roundings as cell_act, so the
// results are bit-identical (tests: cooperative == streaming kernels, which use the scalar form).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_set(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 cell_exp2v(f32x2 x) {
    const f32x2 t = x * pk_set(1.44269502e+00f);
    f32x2 r = pk_fma(x, pk_set(1.44269502e+00f), -t);
    r = pk_fma(x, pk_set(1.92596303e-08f), r);
    const f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    return pk_fma(e, r * pk_set(0.693147182f), e);
}
__device__ __forceinline__ f32x2 cell_rcp2v(f32x2 d) {
    const f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    return pk_fma(pk_fma(-d, r, pk_set(1.0f)), r, r);
}
__device__ __forceinline__ f32x2 cell_act2(f32x2 x, bool is_tanh) {
    const f32x2 ax = {fminf(fabsf(x.x), 43.0f), fminf(fabsf(x.y), 43.0f)};
    f32x2 ut = pk_set(2.0f) * ax;
    f32x2 us = {fminf(fmaxf(-x.x, -87.0f), 87.0f), fminf(fmaxf(-x.y, -87.0f), 87.0f)};
    asm volatile("" : "+v"(ut), "+v"(us));
    const f32x2 u = is_tanh ? ut : us;
    const f32x2 r = cell_rcp2v(pk_set(1.0f) + cell_exp2v(u));
    const f32x2 big = pk_set(1.0f) - pk_set(2.0f) * r;
    const f32x2 x2 = ax * ax;
    f32x2 p = pk_fma(x2, pk_set(2.18694885e-02f), pk_set(-5.39682540e-02f));
    p = pk_fma(x2, p, pk_set(1.33333333e-01f));
    p = pk_fma(x2, p, pk_set(-3.33333333e-01f));
    const f32x2 small = pk_fma(ax * x2, p, ax);
    const f32x2 th = {copysignf(ax.x < 0.25f ? small.x : big.x, x.x), copysignf(ax.y < 0.25f ? small.y : big.y, x.y)};
    return is_tanh ? th : r;
}
__device__ __forceinline__ float cell_sigmoid(float x) { return cell_act(x, false); }
__device__ __forceinline__ float cell_tanh(float x) { return cell_act(x, true); }

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == GNNPN_ACT_RELU) return v < 0.0f ? 0.0f : v;   // NaN stays NaN, as torch.relu
    if (act == GNNPN_ACT_SIGMOID) return sigmoid_f32(v);
    return v;
}
