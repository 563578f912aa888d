// Device helpers shared by the training kernels (train.hip: the shipped decoder; train_attn.hip: 'Bahdanau' attention / glimpses):
// one thread per hidden unit, weight matrices streamed from L2 with coalesced reads along the unit index.
#pragma once
#include "common.h"

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// acc[g] += sum_k W[g*H + j][k] * v[k]  (forward product) from the TRANSPOSED matrix Wt[k][g*H + j] (k-major [H,4H]): for a
// fixed k the threads j of a wave read consecutive floats; k-ascending fma chain per gate
template <int H>
__device__ __forceinline__ void matvec_rows(const float* __restrict__ Wt, const float* v, int j, float (&acc)[4]) {
#pragma unroll 4
    for (int k = 0; k < H; ++k) {
        const float vk = v[k];
        const float* row = Wt + (size_t)k * (4 * H) + j;
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = fmaf(row[g * H], vk, acc[g]);
    }
}
// sum_gu W[gu][j] * d[gu]  (transposed product: column j of W, coalesced across the threads of a wave)
template <int H>
__device__ __forceinline__ float matvec_cols(const float* __restrict__ W, const float* d, int j) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    for (int gu = 0; gu < 4 * H; gu += 4) {
        a0 = fmaf(W[(size_t)(gu + 0) * H + j], d[gu + 0], a0);
        a1 = fmaf(W[(size_t)(gu + 1) * H + j], d[gu + 1], a1);
        a2 = fmaf(W[(size_t)(gu + 2) * H + j], d[gu + 2], a2);
        a3 = fmaf(W[(size_t)(gu + 3) * H + j], d[gu + 3], a3);
    }
    return (a0 + a1) + (a2 + a3);
}

// LSTM cell backward for one unit: pre-activation gates (gi,gf,gg,go), c_prev, c; dh, dc (in: gradient wrt h_t, c_t incl. the
// recurrent parts; out: dc = gradient wrt c_{t-1}); returns the four pre-activation gate gradients
__device__ __forceinline__ void cell_backward(float gi, float gf, float gg, float go, float c_prev, float c, float dh, float& dc,
                                              float (&dg)[4]) {
    const float i = sigm(gi), f = sigm(gf), g = tanhf(gg), o = sigm(go), tc = tanhf(c);
    const float dct = dc + dh * o * (1.0f - tc * tc);
    dg[0] = dct * g * i * (1.0f - i);
    dg[1] = dct * c_prev * f * (1.0f - f);
    dg[2] = dct * i * (1.0f - g * g);
    dg[3] = dh * tc * o * (1.0f - o);
    dc = dct * f;
}
}  // namespace
