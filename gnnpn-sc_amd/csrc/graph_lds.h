// Device helpers shared by the LDS-staged aggregates (graph.hip: whole block in LDS; graph_tiled.hip: destination tile x
// source tile): the lanes of a destination row's lane group take each other's (LDS offset, weight) entries by DPP quad
// permutes folded into the address add, read their 16 bytes of the source row from LDS and add them in edge order with
// separately rounded (packed) multiply and add — the rounding of a materialised message followed by scatter_add.
#pragma once
#include "common.h"

// value of lane L of the own group of LPR (4, 2 or 1) consecutive lanes, as a DPP quad permute (no LDS round trip)
template <int LPR, int L>
__device__ __forceinline__ int quad_from(int v) {
    if constexpr (LPR == 4) return __builtin_amdgcn_update_dpp(0, v, L * 0x55, 0xF, 0xF, true);                 // [L,L,L,L]
    else if constexpr (LPR == 2) return __builtin_amdgcn_update_dpp(0, v, L | (L << 2) | ((2 + L) << 4) | ((2 + L) << 6), 0xF, 0xF, true);
    else return v;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(4))) i32x4_u { int32_t v[4]; };     // 16 bytes at 4-byte alignment: one global_load_dwordx4
struct __attribute__((packed, aligned(4))) f32x4_u { float v[4]; };

// the 4 edges held by lane P of every row's lane group: each lane of the group takes (offset, weight) from there by a DPP
// quad permute (folded into the address add) and requests its 16 bytes of the source row from LDS ...
template <int LPR, bool HAS_W, int P>
__device__ __forceinline__ void lds_agg_read4(const char* __restrict__ tile, const int (&cc)[4], const float (&ww)[4],
                                              int lane_off, float4 (&xv)[4], float (&wq)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        xv[k] = *reinterpret_cast<const float4*>(tile + (quad_from<LPR, P>(cc[k]) + lane_off));
        if (HAS_W) wq[k] = __int_as_float(quad_from<LPR, P>(__float_as_int(ww[k])));
    }
}

// ... and adds them in edge order
template <bool HAS_W, int LPR = 4, int P = 0>
__device__ __forceinline__ void lds_agg_add4(const float4 (&xv)[4], const float (&wq)[4], f32x2& a01, f32x2& a23) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x2 lo = {xv[k].x, xv[k].y}, hi = {xv[k].z, xv[k].w};
        if (HAS_W) {
            const f32x2 w2 = {wq[k], wq[k]};
            lo = lo * w2;                             // -ffp-contract=off: the product is rounded before the add
            hi = hi * w2;
        }
        a01 = a01 + lo;
        a23 = a23 + hi;
    }
}

// the first NP groups of 4 edges of a batch, as straight-line code: the reads of group p+1 are in flight under the adds of p
template <int LPR, bool HAS_W, int NP>
__device__ __forceinline__ void lds_agg_consume(const char* __restrict__ tile, const int (&cc)[4], const float (&ww)[4],
                                                int lane_off, f32x2& a01, f32x2& a23) {
    float4 xa[4], xb[4];
    float wa[4], wb[4];
    lds_agg_read4<LPR, HAS_W, 0>(tile, cc, ww, lane_off, xa, wa);
    if (NP > 1) lds_agg_read4<LPR, HAS_W, 1 % LPR>(tile, cc, ww, lane_off, xb, wb);
    lds_agg_add4<HAS_W>(xa, wa, a01, a23);
    if (NP > 2) lds_agg_read4<LPR, HAS_W, 2 % LPR>(tile, cc, ww, lane_off, xa, wa);
    if (NP > 1) lds_agg_add4<HAS_W>(xb, wb, a01, a23);
    if (NP > 3) lds_agg_read4<LPR, HAS_W, 3 % LPR>(tile, cc, ww, lane_off, xb, wb);
    if (NP > 2) lds_agg_add4<HAS_W>(xa, wa, a01, a23);
    if (NP > 3) lds_agg_add4<HAS_W>(xb, wb, a01, a23);
}

