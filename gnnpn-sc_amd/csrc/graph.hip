// Graph kernels of the GNN stage: CSR gather-aggregate (GIN / GCN message passing), GCN symmetric
// normalisation, contiguous-segment mean.  All HBM/L2-bound gathers: one 64-lane wavefront per
// destination row, the row's neighbour list (col, w) fetched 64 entries at a time with one
// coalesced load per array and broadcast lane-by-lane, feature rows read as whole 16 B/lane
// coalesced rows, 4 neighbour rows in flight per wave, accumulation strictly in CSR order with
// separate multiply and add (the rounding of a materialised message followed by scatter_add).
#include "common.h"
#include "graph_lds.h"

template <bool VEC4>
__global__ __launch_bounds__(256) void csr_aggregate_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ w,
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ self_coef,
    const float* __restrict__ bias, const float* __restrict__ scale, const float* __restrict__ shift, int act,
    float* __restrict__ y, int64_t ldy, int32_t n_rows, int32_t C) {
    const int lane = threadIdx.x & 63;
    // XCD-aware mapping (speed only): workgroups are dealt round-robin over the 8 XCDs, so give every
    // XCD one CONTIGUOUS range of destination rows — neighbouring rows share source rows (always true
    // for the block-diagonal batch of service graphs), which then hit that XCD's own L2.
    const int nb = gridDim.x, q = nb >> 3, rem = nb & 7, xcd = blockIdx.x & 7;
    const int blk = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (blockIdx.x >> 3);
    const int row = blk * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int e_begin = rowptr[row], e_end = rowptr[row + 1];
    constexpr int V = VEC4 ? 4 : 1;
    const float one_plus_eps = self_coef ? __fadd_rn(1.0f, *self_coef) : 0.0f;

    for (int c0 = 0; c0 < C; c0 += 64 * V) {   // one pass covers 256 (VEC4) or 64 channels
        const int c = c0 + lane * V;
        const bool live = c < C;
        float acc[V];
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = 0.0f;

        for (int e0 = e_begin; e0 < e_end; e0 += 64) {
            const int cnt = min(64, e_end - e0);
            int my_col = 0;
            float my_w = 1.0f;
            if (lane < cnt) {
                my_col = col[e0 + lane];
                if (w) my_w = w[e0 + lane];
            }
            int j = 0;
            for (; j + 4 <= cnt; j += 4) {   // 4 neighbour rows in flight, summed in order
                float xv[4][V];
                float wj[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int cj = __shfl(my_col, j + u, 64);
                    wj[u] = __shfl(my_w, j + u, 64);
                    const float* src = x + (int64_t)cj * ldx + c;
                    if (live) {
                        if (VEC4) {
                            const float4 t = *reinterpret_cast<const float4*>(src);
                            xv[u][0] = t.x; xv[u][V > 1 ? 1 : 0] = t.y; xv[u][V > 2 ? 2 : 0] = t.z; xv[u][V > 3 ? 3 : 0] = t.w;
                        } else {
                            xv[u][0] = src[0];
                        }
                    } else {
#pragma unroll
                        for (int v = 0; v < V; ++v) xv[u][v] = 0.0f;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < V; ++v)
                        acc[v] = __fadd_rn(acc[v], w ? __fmul_rn(wj[u], xv[u][v]) : xv[u][v]);
            }
            for (; j < cnt; ++j) {
                const int cj = __shfl(my_col, j, 64);
                const float wv = __shfl(my_w, j, 64);
                if (live) {
                    const float* src = x + (int64_t)cj * ldx + c;
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const float t = src[v];
                        acc[v] = __fadd_rn(acc[v], w ? __fmul_rn(wv, t) : t);
                    }
                }
            }
        }
        if (!live) continue;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float r = acc[v];
            if (self_coef) r = __fadd_rn(r, __fmul_rn(one_plus_eps, x[(int64_t)row * ldx + c + v]));
            if (bias) r = __fadd_rn(r, bias[c + v]);
            if (scale) r = __fadd_rn(__fmul_rn(r, scale[c + v]), shift[c + v]);
            acc[v] = apply_act(r, act);
        }
        float* dst = y + (int64_t)row * ldy + c;
        if (VEC4) {
            *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[V > 1 ? 1 : 0], acc[V > 2 ? 2 : 0], acc[V > 3 ? 3 : 0]);
        } else {
            dst[0] = acc[0];
        }
    }
}

extern "C" int gnnpn_csr_aggregate_f32(const int32_t* rowptr, const int32_t* col, const float* w, const float* x,
                                       int64_t ldx, const float* self_coef, const float* bias,
                                       const float* scale, const float* shift, int act, float* y, int64_t ldy,
                                       int32_t n_rows, int32_t C, void* stream) {
    GNNPN_REQUIRE(n_rows >= 0 && C > 0 && ldx >= C && ldy >= C, "csr_aggregate: bad shape");
    if (n_rows == 0) return GNNPN_OK;
    GNNPN_REQUIRE(rowptr && x && y, "csr_aggregate: null operand");
    GNNPN_REQUIRE((scale == nullptr) == (shift == nullptr), "csr_aggregate: scale and shift go together");
    GNNPN_REQUIRE(x != y, "csr_aggregate: in-place aggregation is not supported");
    if (n_rows == 0) return GNNPN_OK;
    const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && gnnpn_aligned(x, 16) && gnnpn_aligned(y, 16);
    dim3 grid((n_rows + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (vec)
        hipLaunchKernelGGL(csr_aggregate_kernel<true>, grid, block, 0, s, rowptr, col, w, x, ldx, self_coef, bias,
                           scale, shift, act, y, ldy, n_rows, C);
    else
        hipLaunchKernelGGL(csr_aggregate_kernel<false>, grid, block, 0, s, rowptr, col, w, x, ldx, self_coef, bias,
                           scale, shift, act, y, ldy, n_rows, C);
    GNNPN_CHECK_LAUNCH("csr_aggregate_f32");
    return GNNPN_OK;
}

// ---------------------------------------------------------------------------------------------
// LDS-staged form for BLOCK-LOCAL graphs (north star: "node features staged in LDS").  A batch of service graphs is block
// diagonal: destination rows [b*R, (b+1)*R) only gather source rows of the same block (R = the table size S).  The kernel
// above re-fetches every source row from L2 once per neighbour (33 x 1 KiB per output row at degree 32: L2-gather-bound,
// 14 % of the HBM roofline).  Here a workgroup owns (block b, channel slice s of SLICE channels): it copies the block's
// slice x[b*R .. (b+1)*R)[SLICE*s .. SLICE*(s+1)) into LDS ONCE ((R + 1) * SLICE * 4 B <= 160 KB: SLICE = 16 up to R = 2559,
// 8 up to 5119, 4 up to 10239) and serves every gather from there, so each source element crosses the L2 -> CU path once.
// LPR = SLICE/4 lanes per destination row, one float4 each; a wave works on 64/LPR rows at a time and walks their neighbour
// lists in step, 4*LPR edges per batch: every lane fetches 4 consecutive (col, w) entries with one 16-byte load per array
// (the next batch's while the current one is consumed) and the lanes of a row take each other's entries by DPP quad
// permutes; rows are dealt to the waves by descending degree (gnnpn_csr_block_row_order) so that rows walking in step have
// the same length.  Sums are strictly in CSR order with separately rounded (packed) multiply and add: bit-identical to
// csr_aggregate_kernel.  24.7 % of the HBM roofline at 2507 x 256 copies (gather form 15.4 %); what bounds it: DESIGN.md 8.
// Placement (speed only): the SLICE-WGs of one block get equal blockIdx % 8 (one XCD), so the two halves of every 128-B
// line of x — read by two different slices — meet in that XCD's L2 and the (col, w) lists are fetched from HBM once.
// the 4 consecutive (col, w) entries at idx of one lane of a row's lane group.  SAFE = no row of the wave ends within 4
// entries of the arrays' end: one unconditional 16-byte load per array (SGPR base + 32-bit lane offset; a lane beyond its
// row's end re-reads the line behind it and the values are ignored).
template <bool HAS_W, bool SAFE>
__device__ __forceinline__ void lds_agg_fetch(const int32_t* __restrict__ col, const float* __restrict__ w, int idx, int e1,
                                              int (&cb)[4], float (&wb)[4]) {
    if (SAFE) {
        const unsigned boff = 4u * (unsigned)min(idx, e1);
        const i32x4_u c4 = *reinterpret_cast<const i32x4_u*>(reinterpret_cast<const char*>(col) + boff);
#pragma unroll
        for (int k = 0; k < 4; ++k) cb[k] = c4.v[k];
        if (HAS_W) {
            const f32x4_u w4 = *reinterpret_cast<const f32x4_u*>(reinterpret_cast<const char*>(w) + boff);
#pragma unroll
            for (int k = 0; k < 4; ++k) wb[k] = w4.v[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (idx + k < e1) {
                cb[k] = col[idx + k];
                if (HAS_W) wb[k] = w[idx + k];
            }
    }
}

// the rows of one pass of a wave (one row per lane group of LPR lanes), all their edges, 4*LPR per batch: entry k of lane p
// of the group is edge 4p+k of the batch.  The next batch's (col, w) are requested before the current batch is consumed, so
// their latency lies under its LDS reads.  Entries beyond a row's end point at the all-zero row behind the block with
// weight 0: acc + 0*0 = acc exactly (acc starts at +0 and therefore is never -0), so the adds carry no masks; a batch that
// is full for every row of the wave skips even that replacement, the last ones stop after the last group of 4 any row needs.
template <int LPR, bool HAS_W, bool SAFE>
__device__ __forceinline__ void lds_agg_rows(const char* __restrict__ tile_b, const int32_t* __restrict__ col,
                                             const float* __restrict__ w, int e, int e1, int zoff, int sub, int lane_off,
                                             f32x2& a01, f32x2& a23) {
    constexpr int SLICE = 4 * LPR;
    int cb[4] = {0, 0, 0, 0};
    float wb[4] = {0.f, 0.f, 0.f, 0.f};
    lds_agg_fetch<HAS_W, SAFE>(col, w, e + 4 * sub, e1, cb, wb);
    for (;;) {
        const int rem = e1 - e;                      // edges this row still has (<= 0: done)
        if (!__any(rem > 0)) break;
        const bool full = __all(rem >= 4 * LPR);
        int cc[4];
        float ww[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cc[k] = cb[k] * (SLICE * 4);             // LDS byte offset of the source row (lane_off carries -r0 and the lane's 16 B)
            ww[k] = wb[k];
        }
        if (!full) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool ok = 4 * sub + k < rem;
                cc[k] = ok ? cc[k] : zoff;
                ww[k] = ok ? ww[k] : 0.0f;
            }
        }
        e += 4 * LPR;
        lds_agg_fetch<HAS_W, SAFE>(col, w, e + 4 * sub, e1, cb, wb);
        if (full) {
            lds_agg_consume<LPR, HAS_W, LPR>(tile_b, cc, ww, lane_off, a01, a23);
        } else {                                     // the groups of 4 any row of the wave still needs
            const int np = LPR == 1 ? 1 : !__any(rem > 4) ? 1 : (LPR == 2 || !__any(rem > 8)) ? 2 : !__any(rem > 12) ? 3 : 4;
            if (np == 1) lds_agg_consume<LPR, HAS_W, 1>(tile_b, cc, ww, lane_off, a01, a23);
            else if (np == 2) lds_agg_consume<LPR, HAS_W, (LPR > 1 ? 2 : 1)>(tile_b, cc, ww, lane_off, a01, a23);
            else if (np == 3) lds_agg_consume<LPR, HAS_W, (LPR > 2 ? 3 : 1)>(tile_b, cc, ww, lane_off, a01, a23);
            else lds_agg_consume<LPR, HAS_W, LPR>(tile_b, cc, ww, lane_off, a01, a23);
        }
    }
}

template <int LPR, bool HAS_W>
__global__ __launch_bounds__(1024) void csr_aggregate_lds_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ w,
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ self_coef, const float* __restrict__ bias,
    const float* __restrict__ scale, const float* __restrict__ shift, int act, float* __restrict__ y, int64_t ldy,
    int32_t n_rows, int32_t C, int32_t R, int32_t n_blocks, int32_t n_slices, const int32_t* __restrict__ row_order) {
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [R + 1][SLICE]: the block's slice and one all-zero row
    constexpr int SLICE = 4 * LPR, RPP = 1024 / LPR;                   // rows per pass of the workgroup
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int s = j % n_slices, b = (j / n_slices) * 8 + xcd;
    if (b >= n_blocks) return;
    const int r0 = b * R, rows = min(R, n_rows - r0);
    const int c0 = s * SLICE;
    const int sub = threadIdx.x % LPR, c = c0 + 4 * sub;               // this lane's 4 channels
    const int nnz = rowptr[n_rows];
    // ---- fill: the block's channel slice, 16 B per lane, 10 rows in flight per lane (a block of 2560 rows in ONE round trip)
    for (int rr = threadIdx.x / LPR; rr < rows; rr += 10 * RPP) {
        float4 v[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const int r = min(rr + u * RPP, rows - 1);                 // clamped: the loads need no branch
            v[u] = *reinterpret_cast<const float4*>(x + (int64_t)(r0 + r) * ldx + c);
        }
#pragma unroll
        for (int u = 0; u < 10; ++u) {                                 // (clamped duplicates rewrite row rows-1 with its own bytes)
            const int r = min(rr + u * RPP, rows - 1);
            *reinterpret_cast<float4*>(tile + r * SLICE + 4 * sub) = v[u];
        }
    }
    if (threadIdx.x < SLICE) tile[R * SLICE + threadIdx.x] = 0.0f;
    __syncthreads();
    const float one_plus_eps = self_coef ? __fadd_rn(1.0f, *self_coef) : 0.0f;
    float bv[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        if (bias) bv[v] = bias[c + v];
        if (scale) {
            sc[v] = scale[c + v];
            sh[v] = shift[c + v];
        }
    }
    const char* tile_b = reinterpret_cast<const char*>(tile);
    const int lane_off = 16 * sub - r0 * (SLICE * 4);                   // + col * SLICE * 4 = the lane's 16 bytes of source row col
    const int zoff = (R + r0) * (SLICE * 4);                            // the zero row, in the same terms
    // ---- gather: the block's rows are dealt to (wave, lane group) in passes of 1024/LPR — in the caller's row_order
    // (descending degree: the 64/LPR rows a wave walks in step then have nearly the same number of edges) or as they come.
    for (int rb = 0; rb < rows; rb += RPP) {
        const int i = rb + threadIdx.x / LPR;
        const bool live = i < rows;
        int r = i, e = 0, e1 = 0;
        if (live) {
            if (row_order) r = row_order[r0 + i] - r0;
            e = rowptr[r0 + r];
            e1 = rowptr[r0 + r + 1];
        }
        f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
        if (__all(e1 + 4 <= nnz)) lds_agg_rows<LPR, HAS_W, true>(tile_b, col, w, e, e1, zoff, sub, lane_off, a01, a23);
        else lds_agg_rows<LPR, HAS_W, false>(tile_b, col, w, e, e1, zoff, sub, lane_off, a01, a23);
        if (!live) continue;
        const float4 own = *reinterpret_cast<const float4*>(tile + r * SLICE + 4 * sub);
        const float o4[4] = {own.x, own.y, own.z, own.w};
        float acc[4] = {a01.x, a01.y, a23.x, a23.y};
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float t = acc[v];
            if (self_coef) t = __fadd_rn(t, __fmul_rn(one_plus_eps, o4[v]));
            if (bias) t = __fadd_rn(t, bv[v]);
            if (scale) t = __fadd_rn(__fmul_rn(t, sc[v]), sh[v]);
            acc[v] = apply_act(t, act);
        }
        *reinterpret_cast<float4*>(y + (int64_t)(r0 + r) * ldy + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

extern "C" int gnnpn_csr_aggregate_blocks_f32(const int32_t* rowptr, const int32_t* col, const float* w, const float* x,
                                              int64_t ldx, const float* self_coef, const float* bias, const float* scale,
                                              const float* shift, int act, float* y, int64_t ldy, int32_t n_rows, int32_t C,
                                              int32_t block_rows, const int32_t* row_order, void* stream) {
    GNNPN_REQUIRE(rowptr && x && y, "csr_aggregate_blocks: null operand");
    GNNPN_REQUIRE(n_rows >= 0 && C > 0 && ldx >= C && ldy >= C && block_rows > 0, "csr_aggregate_blocks: bad shape");
    GNNPN_REQUIRE((scale == nullptr) == (shift == nullptr), "csr_aggregate_blocks: scale and shift go together");
    GNNPN_REQUIRE(x != y, "csr_aggregate_blocks: in-place aggregation is not supported");
    if (n_rows == 0) return GNNPN_OK;
    const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && gnnpn_aligned(x, 16) && gnnpn_aligned(y, 16);
    constexpr int64_t LDS_BYTES = 160 * 1024;
    int lpr = 0;                                   // lanes per row = SLICE / 4: the widest slice whose block fits the LDS
    for (int cand = 4; cand >= 1; cand >>= 1)
        if (C % (4 * cand) == 0 && ((int64_t)block_rows + 1) * 16 * cand <= LDS_BYTES) {
            lpr = cand;
            break;
        }
    if (!vec || lpr == 0)
        GNNPN_FAIL(GNNPN_E_UNSUP, "csr_aggregate_blocks: blocks of %d rows x %d channels do not fit the LDS-staged form "
                   "(16-byte aligned rows, (block_rows + 1) * 16 B <= 160 KB): use gnnpn_csr_aggregate_f32", block_rows, C);
    const int n_blocks = (n_rows + block_rows - 1) / block_rows, n_slices = C / (4 * lpr);
    const unsigned lds = (unsigned)(((int64_t)block_rows + 1) * 16 * lpr);
    dim3 grid((unsigned)(((n_blocks + 7) / 8) * n_slices * 8)), block(1024);
    hipStream_t st = (hipStream_t)stream;
#define GNNPN_AGG_LDS(LPR_, W_)                                                                                          \
    do {                                                                                                                 \
        if (hipFuncSetAttribute((const void*)csr_aggregate_lds_kernel<LPR_, W_>,                                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)                      \
            GNNPN_FAIL(GNNPN_E_LAUNCH, "csr_aggregate_blocks: cannot reserve %u B of LDS", lds);                          \
        hipLaunchKernelGGL((csr_aggregate_lds_kernel<LPR_, W_>), grid, block, lds, st, rowptr, col, w, x, ldx, self_coef, \
                           bias, scale, shift, act, y, ldy, n_rows, C, block_rows, n_blocks, n_slices, row_order);       \
    } while (0)
    if (lpr == 4 && w) GNNPN_AGG_LDS(4, true);
    else if (lpr == 4) GNNPN_AGG_LDS(4, false);
    else if (lpr == 2 && w) GNNPN_AGG_LDS(2, true);
    else if (lpr == 2) GNNPN_AGG_LDS(2, false);
    else if (w) GNNPN_AGG_LDS(1, true);
    else GNNPN_AGG_LDS(1, false);
#undef GNNPN_AGG_LDS
    GNNPN_CHECK_LAUNCH("csr_aggregate_blocks_f32");
    return GNNPN_OK;
}

// The row order the LDS-staged aggregate walks a block in: per block the rows by descending number of edges (ties: ascending
// row), as global row numbers.  One workgroup per block: bitonic sort of 32-bit keys {0xFFFF - min(degree, 0xFFFF), local
// row} in LDS.  A property of the graph: computed once, reused by every layer and call.
__global__ __launch_bounds__(1024) void csr_block_order_kernel(const int32_t* __restrict__ rowptr, int32_t n_rows, int32_t R,
                                                               int32_t P, int32_t* __restrict__ order) {
    extern __shared__ uint32_t okeys[];                                // [P], P = the power of two >= R
    const int r0 = blockIdx.x * R, rows = min(R, n_rows - r0);
    for (int i = threadIdx.x; i < P; i += 1024) {
        uint32_t k = 0xFFFFFFFFu;
        if (i < rows) {
            const int deg = rowptr[r0 + i + 1] - rowptr[r0 + i];
            k = ((uint32_t)(0xFFFF - min(max(deg, 0), 0xFFFF)) << 16) | (uint32_t)i;
        }
        okeys[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += 1024) {
                const int p = i ^ j;
                if (p > i) {
                    const uint32_t a = okeys[i], b = okeys[p];
                    if ((a > b) == ((i & k) == 0)) {
                        okeys[i] = b;
                        okeys[p] = a;
                    }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < rows; i += 1024) order[r0 + i] = r0 + (int32_t)(okeys[i] & 0xFFFFu);
}

extern "C" int gnnpn_csr_block_row_order(const int32_t* rowptr, int32_t n_rows, int32_t block_rows, int32_t* row_order,
                                         void* stream) {
    GNNPN_REQUIRE(n_rows >= 0 && block_rows > 0, "csr_block_row_order: bad shape");
    if (n_rows == 0) return GNNPN_OK;
    GNNPN_REQUIRE(rowptr && row_order, "csr_block_row_order: null operand");
    if (block_rows > 16384)
        GNNPN_FAIL(GNNPN_E_UNSUP, "csr_block_row_order: blocks of %d rows (at most 16384)", block_rows);
    int P = 2;
    while (P < block_rows) P <<= 1;
    const unsigned lds = (unsigned)P * 4u;
    if (hipFuncSetAttribute((const void*)csr_block_order_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        GNNPN_FAIL(GNNPN_E_LAUNCH, "csr_block_row_order: cannot reserve %u B of LDS", lds);
    const int n_blocks = (n_rows + block_rows - 1) / block_rows;
    hipLaunchKernelGGL(csr_block_order_kernel, dim3((unsigned)n_blocks), dim3(1024), lds, (hipStream_t)stream, rowptr, n_rows,
                       block_rows, P, row_order);
    GNNPN_CHECK_LAUNCH("csr_block_row_order");
    return GNNPN_OK;
}

// ---------------------------------------------------------------------------------------------
// GCN normalisation.  deg: one wave per row would waste lanes on short rows; rows are summed
// sequentially (scatter_add order) by one lane each — the arrays are tiny next to the features.
__global__ void gcn_deg_kernel(const int32_t* __restrict__ rowptr, const float* __restrict__ w_raw,
                               float* __restrict__ dis, int32_t n_rows) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    float deg = 0.0f;
    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) deg = __fadd_rn(deg, w_raw[e]);
    // deg.pow(-0.5) is rsqrt = 1/sqrt (both correctly rounded on the CPU); inf -> 0
    const float r = 1.0f / sqrtf(deg);
    dis[i] = isinf(r) ? 0.0f : r;
}

__global__ void gcn_edge_norm_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                     const float* __restrict__ w_raw, const float* __restrict__ dis,
                                     float* __restrict__ norm, int32_t n_rows) {
    // one wave per destination row: dis[src] * w * dis[dst], left to right
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float d_dst = dis[row];
    for (int e = rowptr[row] + lane; e < rowptr[row + 1]; e += 64)
        norm[e] = __fmul_rn(__fmul_rn(dis[col[e]], w_raw[e]), d_dst);
}

extern "C" int gnnpn_gcn_norm_f32(const int32_t* rowptr, const int32_t* col, const float* w_raw,
                                  float* deg_inv_sqrt, float* norm, int32_t n_rows, void* stream) {
    GNNPN_REQUIRE(n_rows >= 0, "gcn_norm: bad shape");
    if (n_rows == 0) return GNNPN_OK;
    GNNPN_REQUIRE(rowptr && col && w_raw && deg_inv_sqrt && norm, "gcn_norm: null operand");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gcn_deg_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, rowptr, w_raw, deg_inv_sqrt, n_rows);
    hipLaunchKernelGGL(gcn_edge_norm_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, s, rowptr, col, w_raw,
                       deg_inv_sqrt, norm, n_rows);
    GNNPN_CHECK_LAUNCH("gcn_norm_f32");
    return GNNPN_OK;
}

// ---------------------------------------------------------------------------------------------
// One wave per (segment, chunk of 64 channels): a column's sum is a sequential chain over the segment's rows (scatter order),
// so the parallelism is segments x channels; the rows' loads are independent and go out 8 at a time.  (The first form gave a
// wave a whole segment and walked its channel chunks one after the other: 512 waves for 512 graphs of 1001 nodes — 0.83 ms
// for 262 MB at the 1000-task shape.)
__global__ __launch_bounds__(256) void segment_mean_kernel(const int32_t* __restrict__ segptr,
                                                           const float* __restrict__ x, int64_t ldx,
                                                           float* __restrict__ out, int64_t ldo, int32_t n_seg,
                                                           int32_t C, int32_t n_chunks) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int g = item / n_chunks, c = (item - g * n_chunks) * 64 + lane;
    if (g >= n_seg || c >= C) return;
    const int b = segptr[g], e = segptr[g + 1];
    const float cnt = (float)max(e - b, 1);
    const float* col = x + c;
    float acc = 0.0f;
    int n = b;
    // 32 rows in flight per lane (long segments — 1001 nodes per workflow at the 1000-task shapes — are one wave per 64 channels:
    // the loop is a latency chain, 8 in flight took 82 us for 512 x 1001 rows), summed in row order all the same
    for (; n + 32 <= e; n += 32) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = col[(int64_t)(n + u) * ldx];
#pragma unroll
        for (int u = 0; u < 32; ++u) acc = __fadd_rn(acc, v[u]);
    }
    for (; n + 8 <= e; n += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = col[(int64_t)(n + u) * ldx];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __fadd_rn(acc, v[u]);
    }
    for (; n < e; ++n) acc = __fadd_rn(acc, col[(int64_t)n * ldx]);
    out[(int64_t)g * ldo + c] = acc / cnt;
}

extern "C" int gnnpn_segment_mean_f32(const int32_t* segptr, const float* x, int64_t ldx, float* out, int64_t ldo,
                                      int32_t n_seg, int32_t C, void* stream) {
    GNNPN_REQUIRE(n_seg >= 0 && C > 0 && ldx >= C && ldo >= C, "segment_mean: bad shape");
    if (n_seg == 0) return GNNPN_OK;
    GNNPN_REQUIRE(segptr && out, "segment_mean: null operand");      // x may be NULL when every segment is empty
    const int n_chunks = (C + 63) / 64;
    const int64_t items = (int64_t)n_seg * n_chunks;
    hipLaunchKernelGGL(segment_mean_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, (hipStream_t)stream, segptr, x,
                       ldx, out, ldo, n_seg, C, n_chunks);
    GNNPN_CHECK_LAUNCH("segment_mean_f32");
    return GNNPN_OK;
}
