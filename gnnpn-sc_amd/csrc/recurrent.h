// Pieces shared by the encoder recurrence (lstm.hip) and the pointer decoder (decode.hip).
#pragma once
#include "common.h"

// one LSTM cell update from the four gate pre-activations (torch: cy = f*c + i*g ; hy = o*tanh(cy),
// products and sum rounded separately)
__device__ __forceinline__ void lstm_cell_update(float gi, float gf, float gg, float go, float& c, float& h) {
    const float i = cell_sigmoid(gi), f = cell_sigmoid(gf), g = cell_tanh(gg), o = cell_sigmoid(go);
    c = __fadd_rn(__fmul_rn(f, c), __fmul_rn(i, g));
    h = __fmul_rn(o, cell_tanh(c));
}

// acc[p][g] += sum_k W[g*H+j][k] * v[p][k], k ascending, one fmaf chain per (p,g)
template <int H, int BT>
__device__ __forceinline__ void gemv_chain(const float4* __restrict__ Wp, const float (*v)[H], int j,
                                           float (&acc)[BT][4]) {
#pragma unroll 4
    for (int k4 = 0; k4 < H / 4; ++k4) {
        float4 w[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) w[g] = Wp[(k4 * 4 + g) * H + j];
#pragma unroll
        for (int p = 0; p < BT; ++p) {
            const float4 hv = *reinterpret_cast<const float4*>(&v[p][k4 * 4]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float a = acc[p][g];
                a = fmaf(w[g].x, hv.x, a);
                a = fmaf(w[g].y, hv.y, a);
                a = fmaf(w[g].z, hv.z, a);
                a = fmaf(w[g].w, hv.w, a);
                acc[p][g] = a;
            }
        }
    }
}

