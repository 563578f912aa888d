// Two-dimensional LDS-staged CSR aggregate: destination tile x source tile (north star: "CSR neighbour lists read with
// coalesced HBM loads, node features staged in LDS"), for block-local graphs whose blocks are too large for the
// whole-block form of graph.hip (more than 2559 rows per block: the 1000-task / 5000-candidate and 2000-task /
// 20000-candidate service graphs), and a faster walk of the edge lists for the ones that fit.
//
// What bounds the other two forms.  The gather form (csr_aggregate_kernel) fetches every source row once per EDGE out
// of L2 — E*C*4 = 21.6 GB per layer at 5000 rows x 128 copies for 1.48 GB of algorithmic bytes — and sits on the chip's
// L2 / Infinity-Cache gather ceilings (14 % / 8 % of the HBM roofline).  The whole-block LDS form needs
// (rows + 1) * slice * 4 B <= 160 KB: 16-channel slices end at 2559 rows, and 8- / 4-channel slices lose to the gather.
//
// This form keeps the 16-channel slices and cuts the SOURCE rows into tiles of <= 2559 rows; a workgroup owns
// (block, destination tile of <= 2560 rows, channel slice), keeps ALL its destination rows' running sums in registers
// (10 rows x 4 channels per lane: the register file of a CU is 512 KB, three times its LDS) and walks the source tiles in
// order: fill the tile's slice into LDS, add every destination row's edges that point into it, next tile.  The sums are
// DEFINED by the edge order (scatter_add over the edge list), so this is only valid for rows whose neighbour lists
// visit the source tiles in non-decreasing order, with an optional trailing self loop (added after the last tile from
// global memory) — true for the reference's own co-occurrence scan (src/loadData.py:56-65 emits the pairs (i, j), i < j,
// in lexicographic order, and add_remaining_self_loops appends the loops), checked per graph by the plan builder;
// other graphs stay on the gather form.
//
// The edge lists are not read as CSR.  A PLAN, built once per graph (it is a property of the graph, like the row
// order of the whole-block form), holds them in the order the wavefronts consume them — a sliced-ELL stream:
//   * the destination rows of a tile are sorted by their per-source-tile edge counts (in groups of 4 edges:
//     lexicographic, first tile most significant) and cut into UNITS of 16 rows, one wavefront each (4 lanes per row,
//     4 channels per lane), so that rows that advance in step need the same number of steps in every source tile;
//   * per (unit, source tile) the stream holds QUADS of 512 B: [16 rows][4] LDS byte offsets of the source rows'
//     slices, then [16 rows][4] weights — quad q holds edges 4q .. 4q+3 of each of the unit's rows in that tile; lane
//     4j+p takes quad 4k+p of row j in its k-th pair of 16-byte loads (four 256-byte runs per load instruction);
//     entries past a row's last edge in that tile point at the all-zero row behind the tile with weight 0
//     (acc + 0*0 = acc exactly), so the hot loop has no masks, no row pointers and no per-row state, against two
//     16-byte loads per LANE from 16 different cache lines in the CSR walk of graph.hip (whose neighbour lists thrash
//     the 32 KB vector L1);
//   * a header word pair per (unit, source tile): first quad, number of quads (wave-uniform: scalar registers).
// Same sums in the same order with separately rounded (packed) multiply and add: bit-identical to
// csr_aggregate_kernel (tests/test_gpu_ops.py).
#include "common.h"
#include "graph_lds.h"

namespace {

constexpr int SRC_TILES_MAX = 8;         // 8 bits of the sort key per source tile
constexpr int META_WORDS = 128;
constexpr int PASSES_MAX = 10;            // destination rows per lane group: 10 x 4 accumulator registers
constexpr int META_INVALID = 0, META_QUADS = 1, META_EDGES = 2, META_SLOTS = 3, META_STREAM_ROWS = 4, META_HIST = 8, HIST_BINS = 64;

// (The variants that were measured and not kept — 8 waves with half-size tiles, a persistent walk, a shared plan, eight quads of
// stream look-ahead, deeper fills — and the timing-only ablation builds live in tools/experiments/aggregate_switches.patch, applied by
// tools/ablate_aggregate.py; profiles/LOG_r01_r04.md has the numbers.)
constexpr int WAVES = 16;                 // wavefronts per workgroup: one workgroup per CU
constexpr int TILE_ROWS_MAX = 2559;       // source rows per tile: (2559 + 1 zero row) * 16 channels * 4 B = 160 KB
constexpr int FILL_LATER = 5;             // rows in flight per lane when a later source tile is filled (the accumulators are live)
constexpr int DST_ROWS_MAX = WAVES * 16 * PASSES_MAX;   // destination rows per workgroup: 10 passes of 256 rows (1024 lanes, 4 per row)

struct Geom {
    int n_blocks, R, NT, TR, ND, DR, U, passes;
};

inline bool tile_geometry(int n_rows, int block_rows, Geom& g) {
    if (n_rows <= 0 || block_rows <= 0) return false;
    g.R = block_rows;
    g.n_blocks = (n_rows + block_rows - 1) / block_rows;
    g.NT = (block_rows + TILE_ROWS_MAX - 1) / TILE_ROWS_MAX;
    g.TR = (block_rows + g.NT - 1) / g.NT;
    g.ND = (block_rows + DST_ROWS_MAX - 1) / DST_ROWS_MAX;
    g.DR = (((block_rows + g.ND - 1) / g.ND) + 15) / 16 * 16;
    g.U = g.DR / 16;
    g.passes = (g.U + WAVES - 1) / WAVES;
    return g.NT <= SRC_TILES_MAX;
}

// ---- plan, step 1: per (block, destination tile) the rows' per-tile runs, validity, sort into units, quads per (unit, source tile)
__global__ __launch_bounds__(1024) void tile_plan_rows_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ w, int32_t n_rows, Geom g,
    int32_t P, int32_t* __restrict__ header, int32_t* __restrict__ order, int32_t* __restrict__ tstart,
    float* __restrict__ selfw, unsigned* __restrict__ meta) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);                 // [P]
    unsigned short* idx = reinterpret_cast<unsigned short*>(smem + (size_t)P * 8);          // [P]
    unsigned short* runs = reinterpret_cast<unsigned short*>(smem + (size_t)P * 10);        // [DR][NT]
    unsigned* hist = reinterpret_cast<unsigned*>(smem + (size_t)P * 10 + (size_t)g.DR * g.NT * 2);   // [HIST_BINS + 4]
    float* selfw_row = reinterpret_cast<float*>(hist + HIST_BINS + 4);                               // [DR]: by row, written out in unit order
    const int b = blockIdx.x / g.ND, d = blockIdx.x % g.ND;
    const int r0 = b * g.R, Rb = min(g.R, n_rows - r0);
    const int rows_d = max(0, min(g.DR, Rb - d * g.DR));
    for (int i = threadIdx.x; i < HIST_BINS + 4; i += 1024) hist[i] = 0u;
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += 1024) {
        unsigned long long key = 0ull;
        unsigned short id = 0xFFFFu;
        if (i < rows_d) {
            id = (unsigned short)i;
            const int r = r0 + d * g.DR + i;
            int e0 = rowptr[r], e1 = rowptr[r + 1];
            // A trailing self loop whose source lies in the LAST source tile is in tile order: an ordinary entry of the stream,
            // served from LDS.  In an earlier tile it is out of order: the epilogue adds it from global memory (selfw).
            float sw = __uint_as_float(0x7FC00000u);                      // NaN: nothing for the epilogue to add
            if (e1 > e0 && col[e1 - 1] == r && (r - r0) / g.TR != g.NT - 1) {
                sw = w ? w[e1 - 1] : 1.0f;
                --e1;
            }
            selfw_row[i] = sw;
            int run[SRC_TILES_MAX];
#pragma unroll
            for (int t = 0; t < SRC_TILES_MAX; ++t) run[t] = 0;
            bool ok = true;
            int tprev = 0;
            for (int e = e0; e < e1; ++e) {
                const int c = col[e] - r0;
                if (c < 0 || c >= Rb) {                                   // not block-local
                    ok = false;
                    break;
                }
                const int t = c / g.TR;
                ok = ok && t >= tprev;                                    // the list must visit the source tiles in order
                tprev = max(tprev, t);
#pragma unroll
                for (int u = 0; u < SRC_TILES_MAX; ++u) run[u] += (u == t) ? 1 : 0;
            }
            int ts = e0, edges = 0;
#pragma unroll
            for (int t = 0; t < SRC_TILES_MAX; ++t) {
                if (t < g.NT) {
                    tstart[(int64_t)r * (g.NT + 1) + t] = ts;
                    ok = ok && run[t] <= 0xFFFF;
                    const int rt = min(run[t], 0xFFFF);
                    runs[i * g.NT + t] = (unsigned short)rt;
                    ts += run[t];
                    edges += run[t];
                    key |= (unsigned long long)min((rt + 3) >> 2, 255) << (8 * (7 - t));
                    atomicAdd(&hist[min(rt, HIST_BINS - 1)], 1u);
                }
            }
            tstart[(int64_t)r * (g.NT + 1) + g.NT] = ts;
            atomicAdd(&hist[HIST_BINS], (unsigned)edges);
            if (!ok) atomicAdd(&hist[HIST_BINS + 1], 1u);
        }
        keys[i] = key;
        idx[i] = id;
    }
    __syncthreads();
    // bitonic sort: key descending, ties by row ascending (padding entries carry id 0xFFFF and key 0: last)
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < P; i += 1024) {
                const int p = i ^ j;
                if (p > i) {
                    const unsigned long long ka = keys[i], kb = keys[p];
                    const unsigned short ia = idx[i], ib = idx[p];
                    const bool a_first = ka > kb || (ka == kb && ia < ib);      // a belongs before b
                    if (a_first != ((i & k) == 0)) {
                        keys[i] = kb;
                        keys[p] = ka;
                        idx[i] = ib;
                        idx[p] = ia;
                    }
                }
            }
            __syncthreads();
        }
    const int64_t bd = blockIdx.x;
    for (int i = threadIdx.x; i < g.U * 16; i += 1024) {
        const unsigned short id = idx[i];
        order[bd * g.U * 16 + i] = id == 0xFFFFu ? -1 : d * g.DR + (int)id;
        selfw[bd * g.U * 16 + i] = id == 0xFFFFu ? __uint_as_float(0x7FC00000u) : selfw_row[id];
    }
    unsigned slots = 0;
    for (int it = threadIdx.x; it < g.U * g.NT; it += 1024) {
        const int t = it / g.U, u = it % g.U;
        int nq = 0;
        for (int j = 0; j < 16; ++j) {
            const unsigned short id = idx[u * 16 + j];
            if (id != 0xFFFFu) nq = max(nq, ((int)runs[(int)id * g.NT + t] + 3) >> 2);
        }
        header[((bd * g.NT + t) * g.U + u) * 2 + 0] = 0;
        header[((bd * g.NT + t) * g.U + u) * 2 + 1] = nq;
        slots += (unsigned)nq * 64u;
    }
    atomicAdd(&hist[HIST_BINS + 2], slots);
    __syncthreads();
    for (int i = threadIdx.x; i < HIST_BINS; i += 1024)
        if (hist[i]) atomicAdd(meta + META_HIST + i, hist[i]);
    if (threadIdx.x == 0) {
        atomicAdd(meta + META_EDGES, hist[HIST_BINS]);
        if (hist[HIST_BINS + 1]) atomicAdd(meta + META_INVALID, hist[HIST_BINS + 1]);
        atomicAdd(meta + META_SLOTS, hist[HIST_BINS + 2]);
        atomicAdd(meta + META_STREAM_ROWS, (unsigned)rows_d);
    }
}

// ---- plan, step 2: first quad of every (block, destination tile, source tile, unit): exclusive scan of the quad counts
__global__ __launch_bounds__(1024) void tile_plan_scan_kernel(int32_t* __restrict__ header, int64_t n, unsigned* __restrict__ meta) {
    __shared__ unsigned part[1024];
    __shared__ unsigned carry_s;
    if (threadIdx.x == 0) carry_s = 0u;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const unsigned v = i < n ? (unsigned)header[2 * i + 1] : 0u;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {                        // inclusive scan (Hillis-Steele)
            const unsigned add = threadIdx.x >= (unsigned)off ? part[threadIdx.x - off] : 0u;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        const unsigned carry = carry_s;
        if (i < n) header[2 * i] = (int32_t)(carry + part[threadIdx.x] - v);
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) meta[META_QUADS] = carry_s;
}

// ---- plan, step 3: the stream.  One wavefront per (block, destination tile, source tile, unit).
__global__ __launch_bounds__(256) void tile_plan_fill_kernel(
    const int32_t* __restrict__ col, const float* __restrict__ w, int32_t n_rows, Geom g, const int32_t* __restrict__ header,
    const int32_t* __restrict__ order, const int32_t* __restrict__ tstart, uint4* __restrict__ batches, int64_t n_items) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n_items) return;
    const int u = (int)(item % g.U);
    const int t = (int)((item / g.U) % g.NT);
    const int64_t bd = item / ((int64_t)g.U * g.NT);
    const int b = (int)(bd / g.ND);
    const int r0 = b * g.R;
    const int first = header[2 * item], nq = header[2 * item + 1];
    const int j = lane >> 2, p = lane & 3;
    const int rl = order[bd * g.U * 16 + u * 16 + j];                     // block-local destination row, -1: padding
    int e_lo = 0, e_hi = 0;
    if (rl >= 0) {
        e_lo = tstart[(int64_t)(r0 + rl) * (g.NT + 1) + t];
        e_hi = tstart[(int64_t)(r0 + rl) * (g.NT + 1) + t + 1];
    }
    const int zero_off = g.TR * 64;
    for (int q = p; q < nq; q += 4) {                                     // lane (j, p) writes row j's entries of the quads p, p+4, ...
        unsigned off[4], wv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int e = e_lo + 4 * q + m;
            const bool valid = e < e_hi;
            off[m] = valid ? (unsigned)((col[e] - r0 - t * g.TR) * 64) : (unsigned)zero_off;
            wv[m] = valid ? __float_as_uint(w ? w[e] : 1.0f) : 0u;
        }
        uint4* dst = batches + ((int64_t)first + q) * 32;                 // a quad: 16 x 16 B of offsets, 16 x 16 B of weights
        dst[j] = make_uint4(off[0], off[1], off[2], off[3]);
        dst[16 + j] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    }
}

// ---- the aggregate ------------------------------------------------------------------------------------------------
// One unit (16 destination rows, this wavefront) against the source tile in LDS: `nq` > 0 quads of 4 edges per row.
// bp: the unit's first quad (wave-uniform), lane_boff: this lane's 16 bytes of quad `sub` in it.  (co, wv) hold the unit's first four quads on entry — requested
// while the PREVIOUS unit was being consumed — and the first four quads of the next unit (nb) on exit: one pair of loads
// is always in flight under the adds, across the units, passes and source tiles of a wavefront.  (Measured before this:
// every unit began with a header load and a stream load whose L2 round trips nothing covered — 28 % of the kernel for
// the stream loads alone, tools/ablate_aggregate.py.)  Quads beyond nq are loaded — the stream ends with FOUR quads of
// slack (an empty last unit starts AT the stream's end and still requests four) — and never used.
__device__ __forceinline__ void tiled_unit(const char* __restrict__ bp, const char* __restrict__ nb, unsigned lane_boff, int nq,
                                           uint4& co, uint4& wv, const char* __restrict__ tile_b, int lane_off, f32x2& a01,
                                           f32x2& a23) {
    for (;;) {
        const int cc[4] = {(int)co.x, (int)co.y, (int)co.z, (int)co.w};
        const float ww[4] = {__uint_as_float(wv.x), __uint_as_float(wv.y), __uint_as_float(wv.z), __uint_as_float(wv.w)};
        bp += 2048;                                   // four quads
        const char* src = nq > 4 ? bp : nb;           // this unit's next four quads, or the next unit's first four (wave-uniform)
        co = *reinterpret_cast<const uint4*>(src + lane_boff);            // scalar base + 32-bit lane offset
        wv = *reinterpret_cast<const uint4*>(src + lane_boff + 256);
        if (nq >= 4) lds_agg_consume<4, true, 4>(tile_b, cc, ww, lane_off, a01, a23);
        else if (nq == 3) lds_agg_consume<4, true, 3>(tile_b, cc, ww, lane_off, a01, a23);
        else if (nq == 2) lds_agg_consume<4, true, 2>(tile_b, cc, ww, lane_off, a01, a23);
        else lds_agg_consume<4, true, 1>(tile_b, cc, ww, lane_off, a01, a23);
        nq -= 4;
        if (nq <= 0) break;
    }
}

// ---- the same walk in batches of TWO quads, two batches in flight (round 5).  Where the rows' runs per source tile are short
// (10000 / 20000 rows per block: 2 - 3 / 1 - 2 quads per unit and tile) the form above loads four quads to use one or two and has
// ONE request in flight per unit — a unit then costs an L2 round trip, not its arithmetic.  Here a lane's ONE 16-byte load per batch
// holds, by its position in the row's lane group, the offsets of quad 0 | offsets of quad 1 | weights of quad 0 | weights of
// quad 1 (same stream, other addresses), the consumer takes offsets from lane P and weights from lane P + 2 of the group, and the
// requests run two batches ahead of the adds through a cursor over (unit, quad) that crosses units, passes and source tiles.
// Same sums in the same order: bit-identical to the four-quad walk.  20000 x 8: 0.440 -> 0.378 ms, 10000 x 32: 0.527 -> 0.508;
// at 5000 x 128 and 2507 x 256 (5 - 9 quads per unit and tile) the four-quad walk stays ahead (0.739 / 0.608 against 0.749 / 0.657),
// so the launcher takes this walk for blocks of three or more source tiles.
template <int PO, int PW>      // four edges: offsets from lane PO, weights from lane PW of the row's lane group
__device__ __forceinline__ void pair_read4(const char* __restrict__ tile, const uint4& bv, int lane_off, float4 (&xv)[4]) {
    const int cc[4] = {(int)bv.x, (int)bv.y, (int)bv.z, (int)bv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) xv[k] = *reinterpret_cast<const float4*>(tile + (quad_from<4, PO>(cc[k]) + lane_off));
}
template <int PW>
__device__ __forceinline__ void pair_add4(const float4 (&xv)[4], const uint4& bv, f32x2& a01, f32x2& a23) {
    const int wb[4] = {(int)bv.x, (int)bv.y, (int)bv.z, (int)bv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float wq = __int_as_float(quad_from<4, PW>(wb[k]));
        const f32x2 w2 = {wq, wq};
        a01 = a01 + f32x2{xv[k].x, xv[k].y} * w2;     // -ffp-contract=off: the product is rounded before the add
        a23 = a23 + f32x2{xv[k].z, xv[k].w} * w2;
    }
}
template <int NP>
__device__ __forceinline__ void lds_agg_consume_pair(const char* __restrict__ tile, const uint4& bv, int lane_off, f32x2& a01, f32x2& a23) {
    float4 xa[4], xb[4];
    pair_read4<0, 2>(tile, bv, lane_off, xa);
    if (NP > 1) pair_read4<1, 3>(tile, bv, lane_off, xb);
    pair_add4<2>(xa, bv, a01, a23);
    if (NP > 1) pair_add4<3>(xb, bv, a01, a23);
}

// The unit of wavefront `wave` in pass p.  The units are sorted by descending work, so the passes deal them serpentine:
// with every pass giving wave 0 the heaviest of its units, wave 0 would carry the difference between the first and the
// last unit of the tile more than the last wave, and every source-tile switch (a workgroup barrier) waits for the slowest wave.
__device__ __forceinline__ int unit_of(int p, int wave) { return p * WAVES + ((p & 1) ? WAVES - 1 - wave : wave); }

// `srows` rows of a source tile's 16-channel slice (src: the tile's first row, uniform over the workgroup: scalar registers; c: this
// lane's channel; the row offset stays 32-bit: block_rows * ldx < 2^29, checked by the launcher) -> LDS through registers,
// DEPTH rows in flight per lane
template <int DEPTH>
__device__ __forceinline__ void fill_tile(float* __restrict__ tile, const float* __restrict__ src, int ldx, int c, int srows, int tid, int sub) {
    constexpr int RPP = WAVES * 16;
    for (int rr = tid >> 2; rr < srows; rr += DEPTH * RPP) {
        float4 v[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const int r = min(rr + k * RPP, srows - 1);                   // clamped: the loads need no branch
            v[k] = *reinterpret_cast<const float4*>(src + (unsigned)(r * ldx + c));
        }
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {                                 // (clamped duplicates rewrite row srows-1 with its own bytes)
            const int r = min(rr + k * RPP, srows - 1);
            *reinterpret_cast<float4*>(tile + r * 16 + 4 * sub) = v[k];
        }
    }
}

// One workgroup per item (block, destination tile, 16-channel slice); the tile fills a CU's LDS, so one workgroup per CU.
// (Measured and not kept, tools/ablate_aggregate.py on 2507 x 256 / 5000 x 128 / 20000 x 8 copies: PERSISTENT workgroups
// that request the next item's first tile by LDS-DMA (global_load_lds_dwordx4) before the current item's epilogue —
// 0.70 / 0.89 / 0.51 ms against 0.68 / 0.81 / 0.46 ms for this form: the DMA fill of 64-byte pieces of 16 rows per
// wave-instruction is slower than staging the same pieces through registers; two workgroups of 8 wavefronts per CU on
// half-size source tiles — 0.82 / 1.07 ms: twice the source tiles cost more slots in the stream than the overlap of one
// workgroup's fills and stores with the other's gather buys; touching a unit's stream lines ahead of time with one
// 4-byte load per 128-byte line — slower: the touches are as many L2 -> L1 line transfers again; ONE loop over a
// wavefront's batches, unrolled over three register sets (two pairs of stream loads in flight instead of one), the
// passes not unrolled and the unit's sums swapped into fixed registers by a switch — 0.71 / 0.87 / 0.56 ms: the scalar
// control per unit (switches, header look-ups, tile-switch events) costs more than the deeper look-ahead saves, and at
// 2.5 quads per unit (20000 rows) it dominates.)
template <int PASSES, int HREGS, bool PAIRS = false>      // HREGS: registers that hold the headers, 64 (source tile, pass) entries each; PAIRS: the two-quad walk
__global__ __launch_bounds__(WAVES * 64) void csr_aggregate_tiled_kernel(
    const int2* __restrict__ header, const int32_t* __restrict__ order, const float* __restrict__ selfw,
    const uint4* __restrict__ batches, const float* __restrict__ x, int64_t ldx, const float* __restrict__ self_coef,
    const float* __restrict__ bias, const float* __restrict__ scale, const float* __restrict__ shift, int act,
    float* __restrict__ y, int64_t ldy, int32_t n_rows, Geom g, int32_t n_slices) {
    extern __shared__ __attribute__((aligned(16))) float tile[];          // [TR + 1][16]: the source tile's slice and one all-zero row
    constexpr int RPP = WAVES * 16;                                       // rows per pass of the workgroup (4 lanes per row)
    // placement (speed only): the (destination tile, slice) workgroups of one block get equal blockIdx % 8 — one XCD — so
    // that the four 64-byte pieces of every 256-byte stretch of x (four slices) and the block's stream meet in that XCD's L2
    const int xcd = blockIdx.x & 7;
    const int per_block = g.ND * n_slices;
    const int tid = threadIdx.x, sub = tid & 3, lane = tid & 63, wave = tid >> 6;
    const int jj = blockIdx.x >> 3;
    const int b = (jj / per_block) * 8 + xcd;
    if (b >= g.n_blocks) return;
    const int d = (jj % per_block) / n_slices, s = (jj % per_block) % n_slices;
    const int r0 = b * g.R, Rb = min(g.R, n_rows - r0);
    if (Rb - d * g.DR <= 0) return;                             // ragged last block: no rows in this destination tile
    const int c = s * 16 + 4 * sub;
    const char* tile_b = reinterpret_cast<const char*>(tile);
    const int lane_off = 16 * sub;
    const unsigned lane_boff = (unsigned)(sub * 512 + (lane >> 2) * 16);  // this lane's 16 bytes of quad `sub` of a unit, from the unit's first byte
    const char* stream_b = reinterpret_cast<const char*>(batches);
    const int64_t bd = (int64_t)b * g.ND + d;
    if (tid < 16) tile[g.TR * 16 + tid] = 0.0f;                          // the all-zero row: no fill writes it
    // the headers {first quad, quads} of this wavefront's units in EVERY source tile, fetched once: entry e = t * PASSES + p
    // sits in lane e % 64 of register e / 64 and is read back into scalar registers where it is needed
    int hfirst[HREGS], hquads[HREGS];
#pragma unroll
    for (int k = 0; k < HREGS; ++k) {
        const int e = lane + 64 * k;
        int2 h = make_int2(0, 0);
        if (e < g.NT * PASSES) {
            const int u = unit_of(e % PASSES, wave);
            if (u < g.U) h = header[(bd * g.NT + e / PASSES) * g.U + u];
        }
        hfirst[k] = h.x;
        hquads[k] = h.y;
    }
    auto first_of = [&](int e) {
        if constexpr (HREGS == 1) return __builtin_amdgcn_readlane(hfirst[0], e);
        else return e < 64 ? __builtin_amdgcn_readlane(hfirst[0], e) : __builtin_amdgcn_readlane(hfirst[HREGS - 1], e - 64);
    };
    auto quads_of = [&](int e) {
        if constexpr (HREGS == 1) return __builtin_amdgcn_readlane(hquads[0], e);
        else return e < 64 ? __builtin_amdgcn_readlane(hquads[0], e) : __builtin_amdgcn_readlane(hquads[HREGS - 1], e - 64);
    };
    f32x2 a01[PASSES], a23[PASSES];
    // the first tile in ONE round trip (10 rows in flight per lane: the accumulators are not live yet) ...
    const float* xb = x + (int64_t)r0 * ldx;                              // the block's first row: uniform over the workgroup
    const int ldxi = (int)ldx;
    fill_tile<10>(tile, xb, ldxi, c, min(g.TR, Rb), tid, sub);
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        a01[p] = f32x2{0.f, 0.f};
        a23[p] = f32x2{0.f, 0.f};
    }
    if constexpr (PAIRS) {
        const int n_e = g.NT * PASSES;
        const unsigned pair_boff = (unsigned)((sub & 1) * 512 + (sub >> 1) * 256 + (lane >> 2) * 16);
        // the request cursor: (ce, cq) = the next batch to ask for — quads cq, cq + 1 of header entry ce; n_e: none left
        int ce = 0, cq = 0;
        while (ce < n_e && quads_of(ce) == 0) ++ce;
        auto request = [&]() -> uint4 {
            const char* src = ce < n_e ? stream_b + ((int64_t)first_of(ce) + cq) * 512 : stream_b;   // (past the end: any valid address)
            const uint4 v = *reinterpret_cast<const uint4*>(src + pair_boff);
            if (ce < n_e) {
                cq += 2;
                if (cq >= quads_of(ce)) {
                    cq = 0;
                    do ++ce; while (ce < n_e && quads_of(ce) == 0);
                }
            }
            return v;
        };
        uint4 b0 = request(), b1 = request();                             // (one and three in flight measured slower: profiles/LOG_r05.md)
        for (int t = 0; t < g.NT; ++t) {
            if (t) {
                __syncthreads();                                          // every gather from the previous tile is done
                fill_tile<FILL_LATER>(tile, xb + (int64_t)(t * g.TR) * ldx, ldxi, c, min(g.TR, Rb - t * g.TR), tid, sub);
            }
            __syncthreads();
#pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                for (int nq = quads_of(t * PASSES + p); nq > 0; nq -= 2) {
                    const uint4 cur = b0;
                    b0 = b1;
                    b1 = request();
                    if (nq >= 2) lds_agg_consume_pair<2>(tile_b, cur, lane_off, a01[p], a23[p]);
                    else lds_agg_consume_pair<1>(tile_b, cur, lane_off, a01[p], a23[p]);
                }
            }
        }
    } else {
        uint4 co, wv;                                                         // the pair of loads in flight (tiled_unit)
        {
            const char* b0 = stream_b + (int64_t)first_of(0) * 512;
            co = *reinterpret_cast<const uint4*>(b0 + lane_boff);
            wv = *reinterpret_cast<const uint4*>(b0 + lane_boff + 256);
        }
        for (int t = 0; t < g.NT; ++t) {
            if (t) {
                __syncthreads();                                              // every gather from the previous tile is done
                fill_tile<FILL_LATER>(tile, xb + (int64_t)(t * g.TR) * ldx, ldxi, c, min(g.TR, Rb - t * g.TR), tid, sub);   // ... the later ones in two (40 accumulator registers are live)
            }
            __syncthreads();
    #pragma unroll
            for (int p = 0; p < PASSES; ++p) {
                const int e = t * PASSES + p;
                const int nq = quads_of(e);
                // the unit after this one (the next pass, or the first pass of the next source tile; after the last: any valid address)
                const int e_next = e + 1 < g.NT * PASSES ? e + 1 : 0;
                const char* nb = stream_b + (int64_t)first_of(e_next) * 512;
                if (nq > 0) {
                    tiled_unit(stream_b + (int64_t)first_of(e) * 512, nb, lane_boff, nq, co, wv, tile_b, lane_off, a01[p], a23[p]);
                } else {                                  // no edges into this tile: the pair in flight was this unit's — replace it
                    co = *reinterpret_cast<const uint4*>(nb + lane_boff);
                    wv = *reinterpret_cast<const uint4*>(nb + lane_boff + 256);
                }
            }
        }
    }
    // ---- epilogue: the trailing self loop of rows outside the last source tile, then what the gather form's epilogue does.
    // Every condition but `has_loop` is uniform over the launch: the stages run pass by pass on the packed accumulators, with the
    // uniform tests outside the element arithmetic (round 5: the per-element test chain with an inlined sigmoid per element was
    // 1.8 k instructions and 13 % of the launch, issue-bound — profiles/r05_aggregate_phases.json).
    const float one_plus_eps = self_coef ? __fadd_rn(1.0f, *self_coef) : 0.0f;
    const int pos_even = wave * 16 + (lane >> 2), pos_odd = (WAVES - 1 - wave) * 16 + (lane >> 2);   // this lane's row within an even / odd pass
    const int32_t* ob = order + bd * g.U * 16;                            // wave-uniform bases + a 32-bit lane offset
    const float* swb = selfw + bd * g.U * 16;                             // (the loop weights in the same order: the two loads are independent)
    int rl[PASSES];
    float swl[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const bool in = unit_of(p, wave) < g.U;
        rl[p] = in ? ob[p * RPP + ((p & 1) ? pos_odd : pos_even)] : -1;
        swl[p] = in ? swb[p * RPP + ((p & 1) ? pos_odd : pos_even)] : 0.0f;
    }
    // The rows' own features (trailing self loop outside the last source tile, GIN self term) come from global memory: requested
    // for EPI_CHUNK passes at once, unconditionally (rows that need nothing read row 0 of the block and ignore it), then used —
    // one round trip per chunk.  A workgroup whose rows all lie in the last source tile (wave-uniform test) skips it.
    constexpr int EPI_CHUNK = PASSES < 5 ? PASSES : 5;
    const bool needs_own = self_coef != nullptr || d * g.DR < (g.NT - 1) * g.TR;
    if (needs_own) {
#pragma unroll
        for (int p0 = 0; p0 < PASSES; p0 += EPI_CHUNK) {
            float4 own[EPI_CHUNK];
#pragma unroll
            for (int k = 0; k < EPI_CHUNK; ++k)
                own[k] = p0 + k < PASSES ? *reinterpret_cast<const float4*>(xb + (unsigned)(max(rl[p0 + k < PASSES ? p0 + k : 0], 0) * ldxi + c))
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < EPI_CHUNK; ++k) {
                const int p = p0 + k;
                if (p >= PASSES) break;
                const f32x2 o01 = {own[k].x, own[k].y}, o23 = {own[k].z, own[k].w};
                const float sw = swl[p];
                const bool has_loop = sw == sw;                           // NaN: no out-of-order self loop on this row
                const f32x2 sw2 = {sw, sw};
                const f32x2 t01 = a01[p] + o01 * sw2, t23 = a23[p] + o23 * sw2;      // -ffp-contract=off: product rounded, then the add
                a01[p] = has_loop ? t01 : a01[p];
                a23[p] = has_loop ? t23 : a23[p];
                if (self_coef) {
                    const f32x2 e2 = {one_plus_eps, one_plus_eps};
                    a01[p] = a01[p] + o01 * e2;
                    a23[p] = a23[p] + o23 * e2;
                }
            }
        }
    }
    // (the epilogue vectors are fetched here, behind the self-loop stage: twelve registers the stage above has no room for)
    f32x2 bv01 = {0.f, 0.f}, bv23 = {0.f, 0.f}, sc01 = {1.f, 1.f}, sc23 = {1.f, 1.f}, sh01 = {0.f, 0.f}, sh23 = {0.f, 0.f};
    if (bias) {                                                           // (element loads: the vectors need not be 16-byte aligned)
        bv01 = f32x2{bias[c], bias[c + 1]};
        bv23 = f32x2{bias[c + 2], bias[c + 3]};
    }
    if (scale) {
        sc01 = f32x2{scale[c], scale[c + 1]};
        sc23 = f32x2{scale[c + 2], scale[c + 3]};
        sh01 = f32x2{shift[c], shift[c + 1]};
        sh23 = f32x2{shift[c + 2], shift[c + 3]};
    }
    if (bias) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            a01[p] = a01[p] + bv01;
            a23[p] = a23[p] + bv23;
        }
    }
    if (scale) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            a01[p] = a01[p] * sc01 + sh01;
            a23[p] = a23[p] * sc23 + sh23;
        }
    }
    if (act == GNNPN_ACT_RELU) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            a01[p] = f32x2{apply_act(a01[p].x, GNNPN_ACT_RELU), apply_act(a01[p].y, GNNPN_ACT_RELU)};
            a23[p] = f32x2{apply_act(a23[p].x, GNNPN_ACT_RELU), apply_act(a23[p].y, GNNPN_ACT_RELU)};
        }
    } else if (act == GNNPN_ACT_SIGMOID) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            a01[p] = f32x2{sigmoid_f32(a01[p].x), sigmoid_f32(a01[p].y)};
            a23[p] = f32x2{sigmoid_f32(a23[p].x), sigmoid_f32(a23[p].y)};
        }
    }
    // ... and leave in one burst of stores at the very end: a store reads its data registers after it has been issued, so a store
    // between two chunks made the compiler wait for its completion (vmcnt(0)) before it reused them — two more round trips
    float* yb = y + (int64_t)r0 * ldy;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        if (rl[p] < 0) continue;
        *reinterpret_cast<float4*>(yb + (unsigned)(rl[p] * (int)ldy + c)) = make_float4(a01[p].x, a01[p].y, a23[p].x, a23[p].y);
    }
}

}  // namespace

extern "C" int gnnpn_csr_tile_plan_geometry(int32_t n_rows, int32_t block_rows, gnnpn_tile_plan_geom_t* out) {
    GNNPN_REQUIRE(out, "csr_tile_plan_geometry: null output");
    Geom g;
    if (!tile_geometry(n_rows, block_rows, g))
        GNNPN_FAIL(GNNPN_E_UNSUP, "csr_tile_plan_geometry: blocks of %d rows need more than %d source tiles of %d rows "
                   "(or an empty graph): use gnnpn_csr_aggregate_f32", block_rows, SRC_TILES_MAX, TILE_ROWS_MAX);
    out->n_blocks = g.n_blocks;
    out->src_tiles = g.NT;
    out->src_tile_rows = g.TR;
    out->dst_tiles = g.ND;
    out->dst_tile_rows = g.DR;
    out->units = g.U;
    out->wavefronts = WAVES;
    out->passes = g.passes;
    const int64_t bd = (int64_t)g.n_blocks * g.ND;
    out->header_bytes = bd * g.NT * g.U * 8;
    out->order_bytes = bd * g.U * 16 * 4;
    out->tstart_bytes = (int64_t)n_rows * (g.NT + 1) * 4;
    out->selfw_bytes = bd * g.U * 16 * 4;
    out->meta_bytes = META_WORDS * 4;
    return GNNPN_OK;
}

extern "C" int gnnpn_csr_tile_plan_rows(const int32_t* rowptr, const int32_t* col, const float* w, int32_t n_rows,
                                        int32_t block_rows, int32_t* header, int32_t* order, int32_t* tstart, float* selfw,
                                        int32_t* meta, void* stream) {
    GNNPN_REQUIRE(rowptr && col && header && order && tstart && selfw && meta, "csr_tile_plan_rows: null operand");
    Geom g;
    if (!tile_geometry(n_rows, block_rows, g)) GNNPN_FAIL(GNNPN_E_UNSUP, "csr_tile_plan_rows: blocks of %d rows do not fit the tiled form", block_rows);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(meta, 0, META_WORDS * 4, st) != hipSuccess) GNNPN_FAIL(GNNPN_E_LAUNCH, "csr_tile_plan_rows: memset failed");
    int P = 16;
    while (P < g.U * 16) P <<= 1;
    const unsigned lds = (unsigned)((size_t)P * 10 + (size_t)g.DR * g.NT * 2 + (HIST_BINS + 4) * 4 + (size_t)g.DR * 4 + 16);
    if (hipFuncSetAttribute((const void*)tile_plan_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        GNNPN_FAIL(GNNPN_E_LAUNCH, "csr_tile_plan_rows: cannot reserve %u B of LDS", lds);
    hipLaunchKernelGGL(tile_plan_rows_kernel, dim3((unsigned)(g.n_blocks * g.ND)), dim3(1024), lds, st, rowptr, col, w, n_rows, g, P,
                       header, order, tstart, selfw, reinterpret_cast<unsigned*>(meta));
    const int64_t n = (int64_t)g.n_blocks * g.ND * g.NT * g.U;
    hipLaunchKernelGGL(tile_plan_scan_kernel, dim3(1), dim3(1024), 0, st, header, n, reinterpret_cast<unsigned*>(meta));
    GNNPN_CHECK_LAUNCH("csr_tile_plan_rows");
    return GNNPN_OK;
}

extern "C" int gnnpn_csr_tile_plan_fill(const int32_t* col, const float* w, int32_t n_rows, int32_t block_rows,
                                        const int32_t* header, const int32_t* order, const int32_t* tstart, void* batches,
                                        int64_t n_quads, void* stream) {
    GNNPN_REQUIRE(col && header && order && tstart && batches, "csr_tile_plan_fill: null operand");
    GNNPN_REQUIRE(gnnpn_aligned(batches, 16), "csr_tile_plan_fill: the stream must be 16-byte aligned");
    Geom g;
    if (!tile_geometry(n_rows, block_rows, g)) GNNPN_FAIL(GNNPN_E_UNSUP, "csr_tile_plan_fill: blocks of %d rows do not fit the tiled form", block_rows);
    (void)n_quads;
    const int64_t n = (int64_t)g.n_blocks * g.ND * g.NT * g.U;
    hipLaunchKernelGGL(tile_plan_fill_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, col, w, n_rows, g,
                       header, order, tstart, static_cast<uint4*>(batches), n);
    GNNPN_CHECK_LAUNCH("csr_tile_plan_fill");
    return GNNPN_OK;
}

extern "C" int gnnpn_csr_aggregate_tiled_f32(const int32_t* header, const int32_t* order, const float* selfw, const void* batches,
                                             const float* x, int64_t ldx, const float* self_coef, const float* bias,
                                             const float* scale, const float* shift, int act, float* y, int64_t ldy,
                                             int32_t n_rows, int32_t C, int32_t block_rows, void* stream) {
    GNNPN_REQUIRE(n_rows >= 0 && C > 0 && ldx >= C && ldy >= C && block_rows > 0, "csr_aggregate_tiled: bad shape");
    if (n_rows == 0) return GNNPN_OK;
    GNNPN_REQUIRE(header && order && selfw && x && y && batches, "csr_aggregate_tiled: null operand");
    GNNPN_REQUIRE((scale == nullptr) == (shift == nullptr), "csr_aggregate_tiled: scale and shift go together");
    GNNPN_REQUIRE(x != y, "csr_aggregate_tiled: in-place aggregation is not supported");
    GNNPN_REQUIRE((int64_t)block_rows * ldx < (1ll << 29) && (int64_t)block_rows * ldy < (1ll << 29),
                  "csr_aggregate_tiled: a block's rows must span less than 2 GiB (32-bit offsets inside a block)");
    Geom g;
    const bool vec = (C % 16 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && gnnpn_aligned(x, 16) && gnnpn_aligned(y, 16) &&
                     gnnpn_aligned(batches, 16);
    if (!vec || !tile_geometry(n_rows, block_rows, g))
        GNNPN_FAIL(GNNPN_E_UNSUP, "csr_aggregate_tiled: %d channels (a multiple of 16, 16-byte aligned rows) in blocks of %d rows "
                   "(at most %d source tiles of %d) do not fit the tiled form: use gnnpn_csr_aggregate_f32", C, block_rows,
                   SRC_TILES_MAX, TILE_ROWS_MAX);
    const int n_slices = C / 16;
    const unsigned lds = (unsigned)(g.TR + 1) * 64u;
    const int n_jj = ((g.n_blocks + 7) / 8) * g.ND * n_slices;
    dim3 grid((unsigned)(n_jj * 8)), block(WAVES * 64);
    hipStream_t st = (hipStream_t)stream;
    // which walk (speed only: the two give the same bits).  GNNPN_TILED_WALK = quads | pairs overrides the rule (tests run both).
    const char* walk = getenv("GNNPN_TILED_WALK");
    const bool pairs = walk && walk[0] == 'p' ? true : walk && walk[0] == 'q' ? false : g.NT >= 3;
#define GNNPN_AGG_TILED(P_, H_)                                                                                             \
    do {                                                                                                                  \
        if (hipFuncSetAttribute((const void*)csr_aggregate_tiled_kernel<P_, H_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds) != hipSuccess)                                                                  \
            GNNPN_FAIL(GNNPN_E_LAUNCH, "csr_aggregate_tiled: cannot reserve %u B of LDS", lds);                           \
        if (pairs) {                                                                                                      \
            if (hipFuncSetAttribute((const void*)csr_aggregate_tiled_kernel<P_, H_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)lds) != hipSuccess)                                                              \
                GNNPN_FAIL(GNNPN_E_LAUNCH, "csr_aggregate_tiled: cannot reserve %u B of LDS", lds);                       \
            hipLaunchKernelGGL((csr_aggregate_tiled_kernel<P_, H_, true>), grid, block, lds, st, reinterpret_cast<const int2*>(header), \
                               order, selfw, static_cast<const uint4*>(batches), x, ldx, self_coef, bias, scale, shift, act, \
                               y, ldy, n_rows, g, n_slices);                                                        \
        } else                                                                                                            \
        hipLaunchKernelGGL((csr_aggregate_tiled_kernel<P_, H_>), grid, block, lds, st, reinterpret_cast<const int2*>(header), \
                           order, selfw, static_cast<const uint4*>(batches), x, ldx, self_coef, bias, scale, shift, act, \
                           y, ldy, n_rows, g, n_slices);                                                            \
    } while (0)
    // more than 64 (source tile, pass) headers per wavefront only occur with 7 or 8 source tiles, i.e. full destination tiles
    switch (g.passes) {
        case 1: GNNPN_AGG_TILED(1, 1); break;
        case 2: GNNPN_AGG_TILED(2, 1); break;
        case 3: GNNPN_AGG_TILED(3, 1); break;
        case 4: GNNPN_AGG_TILED(4, 1); break;
        case 5: GNNPN_AGG_TILED(5, 1); break;
        case 6: GNNPN_AGG_TILED(6, 1); break;
        case 7: GNNPN_AGG_TILED(7, 1); break;
        case 8: GNNPN_AGG_TILED(8, 1); break;
        default:
            if (g.NT * g.passes <= 64) {
                if (g.passes == 9) GNNPN_AGG_TILED(9, 1);
                else GNNPN_AGG_TILED(10, 1);
            } else {
                if (g.passes == 9) GNNPN_AGG_TILED(9, 2);
                else GNNPN_AGG_TILED(10, 2);
            }
            break;
    }
#undef GNNPN_AGG_TILED
    GNNPN_CHECK_LAUNCH("csr_aggregate_tiled_f32");
    return GNNPN_OK;
}
