// The one-launch GIN layer of gin_layer.hip with its three dense products on the fp16 matrix cores through the EXACT SPLIT
// of coop_common.h (the arithmetic the recurrent kernels use; profiles/LOG_r01_r04.md section 12): every fp32 operand — activations and
// weights — is decomposed into three fp16 pieces that reproduce it bit for bit, the six cross products that can reach
// 2^-24 of a term run on v_mfma_f32_16x16x32_f16 into three fp32 accumulators (one per magnitude class), and nothing else
// of the layer changes: the neighbour aggregate, bias, BatchNorm and ReLU are the fp32 instructions of gin_layer.hip.
// Replaces, per layer, GINConv.propagate + nn (Linear, BN, ReLU, Linear, ReLU) + BN + ReLU [+ nodeLin],
// modelML.py:75-93,139-143,165.  6 x 126 GFLOP of f16 products at the 1000-task shape: 0.30 ms of matrix time against the
// 0.80 ms of the fp32 matrix core.
//
// A workgroup of 4 wavefronts owns 32 rows (two row tiles of 16); a wavefront owns column tiles of 16 outputs, two at a
// time, for BOTH row tiles (a weight fragment is loaded once per workgroup and used twice):
//   0. aggregate (fp32, in CSR order) -> per-row power-of-two scale (row maximum parked in [2^14, 2^15)) -> three fp16 piece
//      tiles in LDS, [piece][row][k] with a row stride of K + 8 halfs (16-byte fragments, bank slots spread);
//   1. pieces . W1 pieces -> + bias, BN, ReLU in fp32 -> row maximum over the 256 outputs (16-lane reduction, four waves
//      through LDS) -> split again -> T piece tiles in LDS;
//   2. T . W2 -> + bias, BN, ReLU -> global, or split again for 3;   3. (nodeLin) . W3 + bias -> global.
// Weights: split ONCE at load time by gnnpn_pack_split_weights_f16 (per output column a power-of-two scale, column maximum
// in [2^14, 2^15)) into the matrix core's B-fragment order: record (column tile, k-block of 32) = {piece 0: 64 lanes x 16 B,
// piece 1: 64 x 16 B, piece 2: 64 x 8 B}.  The third piece is zero or a single power of two — its upper byte IS the fp16 —
// so it is stored as bytes.  A lane's operands of a k-block are three coalesced loads straight into registers.
// Accuracy: tests/test_gpu_ops.py measures the layer against an fp64 evaluation next to the fp32 kernels'.
#include "common.h"
#include "coop_common.h"

// (The timing-only builds and the tile / depth / occupancy / piece-form variants that were measured and not kept live in
// tools/experiments/gin_layer_split_switches.patch, applied by tools/ablate_gin_layer.py; profiles/LOG_r01_r04.md has the numbers.)

namespace {

constexpr int H1 = 256, H2 = 128, H3 = 128;
constexpr int NCT = 2;                                      // column tiles of 16 a wave works on at a time (accumulators: 12-16 registers per (row tile, column tile))
constexpr int RT = 3, BM = 16 * RT;                           // row tiles of 16 per workgroup: a weight fragment is used RT times
constexpr int REC = 2560;                                  // bytes per (column tile, k-block) weight record
constexpr int NCH_MAX = 4;                                 // aggregate: float4 chunks per lane (128 channels / 8 lanes / 4)
constexpr int DEPTH = 2;                                     // k-blocks of weights a wave keeps in flight (registers: 20 per k-block)

struct BFrag {
    f16x8 w0, w1;
    uint2 w2;
};
// rec: the record (uniform over the wave: a scalar base), lane: 32-bit lane offsets — three global_load ... v, s[base:base+1]
__device__ __forceinline__ BFrag load_b(const unsigned char* rec, unsigned lane) {
    BFrag b;
    b.w0 = *reinterpret_cast<const f16x8*>(rec + 16u * lane);
    b.w1 = *reinterpret_cast<const f16x8*>(rec + 1024 + 16u * lane);
    b.w2 = *reinterpret_cast<const uint2*>(rec + 2048 + 8u * lane);
    return b;
}

// power-of-two scale that parks m in [2^14, 2^15), and its inverse (m = 0 and denormal rows: the largest usable scale)
__device__ __forceinline__ void row_scale(float m, float& up, float& inv) {
    int e = (int)((__float_as_uint(m) >> 23) & 255u);
    e = e < 16 ? 16 : (e > 250 ? 250 : e);
    up = __uint_as_float((unsigned)(268 - e) << 23);        // 2^(141 - e)
    inv = __uint_as_float((unsigned)(e - 14) << 23);         // 2^(e - 141)
}

struct Acc {
    f32x4 a0, a0b, a1, a2;
};
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// One k-block of the six products for 2 row tiles x 2 column tiles.  The WEIGHT fragment is the matrix core's first operand and
// the activation fragment its second: the accumulator is the transposed tile — lane (c, kq) holds features 4 kq .. 4 kq + 3 of
// batch row c, i.e. four CONSECUTIVE k's of the next stage (one 8-byte LDS store per piece, one row maximum per lane).
// FIRST: k-blocks 0..3 (the leading product goes to a0), else to a0b — 16 groups of 8 k's per accumulator at most, the bound
// of DESIGN.md section 5; profiles/LOG_r01_r04.md section 12.
template <bool FIRST>
__device__ __forceinline__ void kblock(const _Float16* a_lane, int lda, int piece, int kk, const BFrag (&b)[NCT], Acc (&acc)[RT][NCT]) {
    f16x8 h0[RT], h1[RT], h2[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const _Float16* p = a_lane + rt * 16 * lda + 32 * kk;
        h0[rt] = *reinterpret_cast<const f16x8*>(p);
        h1[rt] = *reinterpret_cast<const f16x8*>(p + piece);
        h2[rt] = *reinterpret_cast<const f16x8*>(p + 2 * piece);
    }
    f16x8 e[NCT];
#pragma unroll
    for (int n = 0; n < NCT; ++n) e[n] = split_expand(b[n].w2.x, b[n].w2.y);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int n = 0; n < NCT; ++n) {
            Acc& c = acc[rt][n];
            c.a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[n].w1, h1[rt], c.a2, 0, 0, 0);
            c.a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[n].w0, h1[rt], c.a1, 0, 0, 0);
            c.a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(e[n], h0[rt], c.a2, 0, 0, 0);
            if (FIRST) c.a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[n].w0, h0[rt], c.a0, 0, 0, 0);
            else c.a0b = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[n].w0, h0[rt], c.a0b, 0, 0, 0);
            c.a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[n].w0, h2[rt], c.a2, 0, 0, 0);
            c.a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[n].w1, h0[rt], c.a1, 0, 0, 0);
        }
}

// v[rt][n] = (A . W^T)(row 16 rt + c, features 16 (ct0 + n) + 4 kq + 0..3), un-scaled by the features' factors (NOT yet by the row's).
// a_lane: piece-0 tile + c * lda + 8 * kq.  w: packed records of this layer; ct0: first of the wave's two column tiles; q: prefetch_b's.
// The first DEPTH k-blocks of a product's weights, [slot][column tile]: issued a whole phase early (before the aggregate, before the
// previous product's epilogue and barriers), so that no product starts by waiting out an L2 round trip.
__device__ __forceinline__ void prefetch_b(const unsigned char* w, int ct0, int kb_n, int lane, BFrag (&q)[DEPTH][NCT]) {
#pragma unroll
    for (int n = 0; n < NCT; ++n) {
        const unsigned char* wl = w + (size_t)(ct0 + n) * kb_n * REC;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (d < kb_n) q[d][n] = load_b(wl + d * REC, lane);
    }
}

template <bool LONGK>   // LONGK: more than 4 k-blocks (K > 128) — the leading product then needs its second accumulator
__device__ __forceinline__ void split_gemm(const _Float16* a_lane, int lda, int piece, int kb_n, const unsigned char* w, int ct0,
                                           const float* __restrict__ col_inv, int lane, BFrag (&q)[DEPTH][NCT], f32x4 (&v)[RT][NCT]) {
    Acc acc[RT][NCT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int n = 0; n < NCT; ++n) acc[rt][n].a0 = acc[rt][n].a0b = acc[rt][n].a1 = acc[rt][n].a2 = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned char* wl[NCT];
#pragma unroll
    for (int n = 0; n < NCT; ++n) wl[n] = w + (size_t)(ct0 + n) * kb_n * REC;
    for (int kk0 = 0; kk0 < kb_n; kk0 += DEPTH) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            const int kk = kk0 + i;
            if (kk < kb_n) {
                BFrag b[NCT];
#pragma unroll
                for (int n = 0; n < NCT; ++n) b[n] = q[i][n];
                if (kk + DEPTH < kb_n) {
#pragma unroll
                    for (int n = 0; n < NCT; ++n) q[i][n] = load_b(wl[n] + (size_t)(kk + DEPTH) * REC, lane);
                }
                if (!LONGK || kk < 4) kblock<true>(a_lane, lda, piece, kk, b, acc);
                else kblock<false>(a_lane, lda, piece, kk, b, acc);
            }
        }
    }
    const int kq = lane >> 4;
#pragma unroll
    for (int n = 0; n < NCT; ++n) {
        const f32x4 ci = *reinterpret_cast<const f32x4*>(col_inv + 16 * (ct0 + n) + 4 * kq);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const Acc& a = acc[rt][n];
            f32x4 lead = a.a0;
            if (LONGK) lead = lead + a.a0b;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[rt][n][r] = __fmul_rn(fmaf(fmaf(a.a2[r], SPLIT_INV, a.a1[r]), SPLIT_INV, lead[r]), ci[r]);
        }
    }
}

__device__ __forceinline__ f32x4 load4_or(const float* p, int at, float dflt) {
    return p ? *reinterpret_cast<const f32x4*>(p + at) : f32x4{dflt, dflt, dflt, dflt};
}
// (v * row factor + bias) * scale + shift, activation — the fp32 epilogue of linear_f32_kernel
__device__ __forceinline__ void finish(f32x4 (&v)[RT][NCT], const float (&rinv)[RT], const float* __restrict__ bias,
                                       const float* __restrict__ scale, const float* __restrict__ shift, int act, int ct0, int kq) {
#pragma unroll
    for (int n = 0; n < NCT; ++n) {
        const int col = 16 * (ct0 + n) + 4 * kq;
        const f32x4 bv = load4_or(bias, col, 0.0f), sc = load4_or(scale, col, 1.0f), sh = load4_or(shift, col, 0.0f);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            f32x4 t = v[rt][n] * rinv[rt];
            if (bias) t = t + bv;
            if (scale) t = t * sc + sh;                      // (-ffp-contract=off: a multiply and an add, as linear_f32_kernel rounds)
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = apply_act(t[r], act);
            v[rt][n] = t;
        }
    }
}

__device__ __forceinline__ void store_global(const f32x4 (&v)[RT][NCT], float* __restrict__ out, int64_t ldo, int64_t m0, int64_t M,
                                             int ct0, int c, int kq) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int64_t row = m0 + 16 * rt + c;
        if (row < M) {
#pragma unroll
            for (int n = 0; n < NCT; ++n) *reinterpret_cast<f32x4*>(out + row * ldo + 16 * (ct0 + n) + 4 * kq) = v[rt][n];
        }
    }
}

// xs (already times the row's power-of-two factor) -> its three fp16 pieces, four features at a time; every residual (x - p) 2^11 is
// exact: convert the piece back, subtract, multiply.
__device__ __forceinline__ f32x4 residual_x4(const f32x4 x, const f16x4 p) {
    return (x - __builtin_convertvector(p, f32x4)) * SPLIT_SCALE;
}
__device__ __forceinline__ void split3x4(const f32x4 xs, f16x4& p0, f16x4& p1, f16x4& p2) {
    p0 = __builtin_convertvector(xs, f16x4);
    const f32x4 r1 = residual_x4(xs, p0);
    p1 = __builtin_convertvector(r1, f16x4);
    p2 = __builtin_convertvector(residual_x4(r1, p1), f16x4);
}

// per-row maximum of |v| over this wave's features -> rowmax[wave][BM]
template <int NP>
__device__ __forceinline__ void publish_row_max(const f32x4 (&v)[NP][RT][NCT], float* rowmax, int wave, int c, int kq) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        float m = 0.0f;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int n = 0; n < NCT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(v[p][rt][n][r]));
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (kq == 0) rowmax[wave * BM + 16 * rt + c] = m;
    }
}
// the row factors every wave derives alike from the four partial maxima, then the pieces of this wave's values -> LDS
template <int NP>
__device__ __forceinline__ void split_rows_to_lds(const f32x4 (&v)[NP][RT][NCT], const float* rowmax, float (&rinv)[RT], _Float16* dst,
                                                  int ld, int piece, const int (&ct0)[NP], int c, int kq) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int row = 16 * rt + c;
        const float m = fmaxf(fmaxf(rowmax[row], rowmax[BM + row]), fmaxf(rowmax[2 * BM + row], rowmax[3 * BM + row]));
        float up;
        row_scale(m, up, rinv[rt]);
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int n = 0; n < NCT; ++n) {
                f16x4 p0, p1, p2;
                split3x4(v[p][rt][n] * up, p0, p1, p2);
                _Float16* at = dst + row * ld + 16 * (ct0[p] + n) + 4 * kq;
                *reinterpret_cast<f16x4*>(at) = p0;
                *reinterpret_cast<f16x4*>(at + piece) = p1;
                *reinterpret_cast<f16x4*>(at + 2 * piece) = p2;
            }
    }
}

template <bool LIN3>
__global__ __launch_bounds__(256, 2) void gin_layer_split_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ x, int64_t ldx, int32_t c_in,
    const float* __restrict__ eps, const unsigned char* __restrict__ w1, const float* __restrict__ i1, const float* __restrict__ b1,
    const float* __restrict__ a1, const float* __restrict__ s1, const unsigned char* __restrict__ w2, const float* __restrict__ i2,
    const float* __restrict__ b2, const float* __restrict__ a2, const float* __restrict__ s2, const unsigned char* __restrict__ w3,
    const float* __restrict__ i3, const float* __restrict__ b3, float* __restrict__ out, int64_t ldo, int64_t M, int32_t k1a, int32_t vec) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // ONE operand buffer, [3 pieces][BM rows][K + 8]: the aggregate's pieces, then T's, then the third stage's operand — each is
    // written after the barrier behind the row maxima of the product that read its predecessor (every wave has finished reading)
    const int lda = k1a + 8, lda3 = H2 + 8, ldt = H1 + 8;
    _Float16* As = reinterpret_cast<_Float16*>(smem_raw);
    _Float16* Ts = As;
    float* rowmax = reinterpret_cast<float*>(As + 3 * BM * ldt);   // [4][BM]
    float* ainv = rowmax + 4 * BM;                          // [BM]: the aggregate's row factors
    const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, c = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform: the weight records' addresses stay in scalar registers
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int pieceA = BM * lda, pieceT = BM * ldt, pieceA3 = BM * lda3;
    BFrag q[DEPTH][NCT];
    // ---- 0. aggregate (the fp32 instructions of gin_layer_kernel), 8 lanes per row, then scale + split
    for (int r = tid >> 3; r < BM; r += 32) {
        const int sub = tid & 7;
        const int64_t row = m0 + r;
        const float one_plus_eps = __fadd_rn(1.0f, *eps);
        int e0 = 0, e1 = 0;
        if (row < M) {
            e0 = rowptr[row];
            e1 = rowptr[row + 1];
        }
        if (r < 32) prefetch_b(w1, 4 * wave, k1a / 32, lane, q);   // the first product's first weights travel under the aggregate's gathers
        float4 acc[NCH_MAX];
#pragma unroll
        for (int i = 0; i < NCH_MAX; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (vec) {
            const int nch = (c_in / 4 - sub + 7) / 8;
            for (int e = e0; e <= e1; ++e) {
                if (row >= M) break;
                const bool own = e == e1;
                const float* src = x + (own ? row : (int64_t)col[e]) * ldx + 4 * sub;
                float4 t[NCH_MAX];
#pragma unroll
                for (int i = 0; i < NCH_MAX; ++i)
                    if (i < nch) t[i] = *reinterpret_cast<const float4*>(src + 32 * i);
#pragma unroll
                for (int i = 0; i < NCH_MAX; ++i)
                    if (i < nch) {
                        if (own) {
                            t[i].x = __fmul_rn(one_plus_eps, t[i].x);
                            t[i].y = __fmul_rn(one_plus_eps, t[i].y);
                            t[i].z = __fmul_rn(one_plus_eps, t[i].z);
                            t[i].w = __fmul_rn(one_plus_eps, t[i].w);
                        }
                        acc[i].x = __fadd_rn(acc[i].x, t[i].x);
                        acc[i].y = __fadd_rn(acc[i].y, t[i].y);
                        acc[i].z = __fadd_rn(acc[i].z, t[i].z);
                        acc[i].w = __fadd_rn(acc[i].w, t[i].w);
                    }
            }
        } else {                                            // c_in <= 32 (the host checks): channel 4 sub + j of chunk 0
            if (row < M) {
                float s[4] = {0.f, 0.f, 0.f, 0.f};
                for (int e = e0; e < e1; ++e) {
                    const float* src = x + (int64_t)col[e] * ldx + 4 * sub;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (4 * sub + j < c_in) s[j] = __fadd_rn(s[j], src[j]);
                }
                const float* own = x + row * ldx + 4 * sub;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * sub + j < c_in) s[j] = __fadd_rn(s[j], __fmul_rn(one_plus_eps, own[j]));
                acc[0] = make_float4(s[0], s[1], s[2], s[3]);
            }
        }
        float m = 0.0f;
#pragma unroll
        for (int i = 0; i < NCH_MAX; ++i)
            m = fmaxf(m, fmaxf(fmaxf(fabsf(acc[i].x), fabsf(acc[i].y)), fmaxf(fabsf(acc[i].z), fabsf(acc[i].w))));
        m = fmaxf(m, __shfl_xor(m, 1, 64));
        m = fmaxf(m, __shfl_xor(m, 2, 64));
        m = fmaxf(m, __shfl_xor(m, 4, 64));
        float up, inv;
        row_scale(m, up, inv);
        if (sub == 0) ainv[r] = inv;
#pragma unroll
        for (int i = 0; i < NCH_MAX; ++i) {
            const int ch = 4 * (sub + 8 * i);
            if (ch < k1a) {                                 // (chunks beyond c_in: the zero padding of the last k-block)
                f16x4 p0, p1, p2;
                split3x4(f32x4{acc[i].x, acc[i].y, acc[i].z, acc[i].w} * up, p0, p1, p2);
                _Float16* at = As + r * lda + ch;
                *reinterpret_cast<f16x4*>(at) = p0;
                *reinterpret_cast<f16x4*>(at + pieceA) = p1;
                *reinterpret_cast<f16x4*>(at + 2 * pieceA) = p2;
            }
        }
    }
    __syncthreads();                                        // the aggregate's pieces and row factors are in LDS
    float rinv[RT];                                         // this lane's rows: c, 16 + c, .. in every stage
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) rinv[rt] = ainv[16 * rt + c];
    // ---- 1. Linear(c_in -> 256) + BN + ReLU: 16 column tiles, wave w owns 4 w .. 4 w + 3, NCT per pass
    {
        constexpr int NP = 4 / NCT;
        f32x4 v[NP][RT][NCT];
        int ct0[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) ct0[p] = 4 * wave + NCT * p;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            split_gemm<false>(As + c * lda + 8 * kq, lda, pieceA, k1a / 32, w1, ct0[p], i1, lane, q, v[p]);
            if (p + 1 < NP) prefetch_b(w1, ct0[p + 1], k1a / 32, lane, q);
            else prefetch_b(w2, 2 * wave, H1 / 32, lane, q);
            finish(v[p], rinv, b1, a1, s1, GNNPN_ACT_RELU, ct0[p], kq);
        }
        publish_row_max<NP>(v, rowmax, wave, c, kq);
        __syncthreads();
        split_rows_to_lds<NP>(v, rowmax, rinv, Ts, ldt, pieceT, ct0, c, kq);
    }
    __syncthreads();                                        // T is complete (nobody reads the aggregate or rowmax any more)
    // ---- 2. Linear(256 -> 128) + BN + ReLU: 8 column tiles, wave w owns 2 w, 2 w + 1
    constexpr int NP2 = 2 / NCT;
    {
        f32x4 v[NP2][RT][NCT];
        int ct0[NP2];
#pragma unroll
        for (int p = 0; p < NP2; ++p) ct0[p] = 2 * wave + NCT * p;
#pragma unroll
        for (int p = 0; p < NP2; ++p) {
            split_gemm<true>(Ts + c * ldt + 8 * kq, ldt, pieceT, H1 / 32, w2, ct0[p], i2, lane, q, v[p]);
            if (p + 1 < NP2) prefetch_b(w2, ct0[p + 1], H1 / 32, lane, q);
            else if (LIN3) prefetch_b(w3, 2 * wave, H2 / 32, lane, q);
            finish(v[p], rinv, b2, a2, s2, GNNPN_ACT_RELU, ct0[p], kq);
            if (!LIN3) store_global(v[p], out, ldo, m0, M, ct0[p], c, kq);
        }
        if (LIN3) {
            publish_row_max<NP2>(v, rowmax, wave, c, kq);
            __syncthreads();
            split_rows_to_lds<NP2>(v, rowmax, rinv, As, lda3, pieceA3, ct0, c, kq);
        }
    }
    // ---- 3. nodeLin: Linear(128 -> 128) + bias
    if (LIN3) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NP2; ++p) {
            f32x4 v[RT][NCT];
            split_gemm<false>(As + c * lda3 + 8 * kq, lda3, pieceA3, H2 / 32, w3, 2 * wave + NCT * p, i3, lane, q, v);
            if (p + 1 < NP2) prefetch_b(w3, 2 * wave + NCT * (p + 1), H2 / 32, lane, q);
            finish(v, rinv, b3, nullptr, nullptr, GNNPN_ACT_NONE, 2 * wave + NCT * p, kq);
            store_global(v, out, ldo, m0, M, 2 * wave + NCT * p, c, kq);
        }
    }
}

// One block of 64 lanes per column tile of 16 outputs: lane (c, kq) walks W[16 t + c][32 kk + 8 kq + j].
__global__ __launch_bounds__(64) void pack_split_weights_kernel(const float* __restrict__ w, int64_t ldw, int n_out, int k,
                                                                unsigned char* __restrict__ packed, float* __restrict__ col_inv) {
    const int lane = threadIdx.x, c = lane & 15, kq = lane >> 4, t = blockIdx.x;
    const int kb_n = (k + 31) / 32;
    const float* wr = w + (int64_t)(16 * t + c) * ldw;
    float m = 0.0f;
    for (int i = 0; i < k; ++i) m = fmaxf(m, fabsf(wr[i]));
    int e = 0;
    if (m > 0.0f && m < 3.0e38f) (void)frexpf(m, &e);       // m = f 2^e, f in [0.5, 1)
    int s = m > 0.0f ? 15 - e : 0;                          // m 2^s in [2^14, 2^15)
    s = s > 96 ? 96 : (s < -96 ? -96 : s);
    const float up = ldexpf(1.0f, s);
    if (kq == 0) col_inv[16 * t + c] = ldexpf(1.0f, -s);
    for (int kk = 0; kk < kb_n; ++kk) {
        unsigned short q0[8], q1[8];
        unsigned bytes[2] = {0u, 0u};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kidx = 32 * kk + 8 * kq + j;
            const float v = kidx < k ? wr[kidx] : 0.0f;
            _Float16 p0, p1, p2;
            split3(__fmul_rn(v, up), p0, p1, p2);
            q0[j] = f16_bits(p0);
            q1[j] = f16_bits(p1);
            bytes[j >> 2] |= (unsigned)(f16_bits(p2) >> 8) << (8 * (j & 3));
        }
        unsigned char* rec = packed + ((size_t)t * kb_n + kk) * REC;
        uint4* d0 = reinterpret_cast<uint4*>(rec + 16 * lane);
        uint4* d1 = reinterpret_cast<uint4*>(rec + 1024 + 16 * lane);
        *d0 = uint4{(unsigned)q0[0] | ((unsigned)q0[1] << 16), (unsigned)q0[2] | ((unsigned)q0[3] << 16),
                    (unsigned)q0[4] | ((unsigned)q0[5] << 16), (unsigned)q0[6] | ((unsigned)q0[7] << 16)};
        *d1 = uint4{(unsigned)q1[0] | ((unsigned)q1[1] << 16), (unsigned)q1[2] | ((unsigned)q1[3] << 16),
                    (unsigned)q1[4] | ((unsigned)q1[5] << 16), (unsigned)q1[6] | ((unsigned)q1[7] << 16)};
        *reinterpret_cast<uint2*>(rec + 2048 + 8 * lane) = uint2{bytes[0], bytes[1]};
    }
}

}  // namespace

extern "C" int64_t gnnpn_split_weights_bytes(int32_t n_out, int32_t k) {
    if (n_out <= 0 || k <= 0 || n_out % 16) return 0;
    return (int64_t)(n_out / 16) * ((k + 31) / 32) * REC;
}

extern "C" int gnnpn_pack_split_weights_f16(const float* w, int64_t ldw, int32_t n_out, int32_t k, void* packed, float* col_inv,
                                            void* stream) {
    GNNPN_REQUIRE(w && packed && col_inv && n_out > 0 && k > 0 && ldw >= k, "pack_split_weights: bad arguments");
    GNNPN_REQUIRE(n_out % 16 == 0, "pack_split_weights: output features must be a multiple of 16 (got %d)", n_out);
    GNNPN_REQUIRE(gnnpn_aligned(packed, 16), "pack_split_weights: the packed buffer must be 16-byte aligned");
    hipLaunchKernelGGL(pack_split_weights_kernel, dim3((unsigned)(n_out / 16)), dim3(64), 0, (hipStream_t)stream, w, ldw, n_out, k,
                       (unsigned char*)packed, col_inv);
    GNNPN_CHECK_LAUNCH("pack_split_weights_f16");
    return GNNPN_OK;
}

extern "C" int gnnpn_gin_layer_split(const int32_t* rowptr, const int32_t* col, const float* x, int64_t ldx, int32_t c_in,
                                     const float* eps, const void* w1, const float* inv1, const float* b1, const float* bn1_scale,
                                     const float* bn1_shift, int32_t h1, const void* w2, const float* inv2, const float* b2,
                                     const float* bn2_scale, const float* bn2_shift, int32_t h2, const void* w3, const float* inv3,
                                     const float* b3, int32_t h3, float* out, int64_t ldo, int64_t n_rows, void* stream) {
    GNNPN_REQUIRE(n_rows >= 0 && c_in > 0 && ldx >= c_in, "gin_layer_split: bad shape");
    if (n_rows == 0) return GNNPN_OK;
    GNNPN_REQUIRE(rowptr && x && eps && w1 && inv1 && w2 && inv2 && out, "gin_layer_split: null operand");   // col may be NULL for a graph without edges
    GNNPN_REQUIRE((w3 == nullptr) == (inv3 == nullptr), "gin_layer_split: w3 and inv3 go together");
    GNNPN_REQUIRE((bn1_scale == nullptr) == (bn1_shift == nullptr) && (bn2_scale == nullptr) == (bn2_shift == nullptr),
                  "gin_layer_split: scale and shift go together");
    GNNPN_REQUIRE(x != out, "gin_layer_split: in-place layers are not supported");
    const bool lin3 = w3 != nullptr;
    const bool vec = c_in % 4 == 0 && ldx % 4 == 0 && gnnpn_aligned(x, 16);
    if (h1 != H1 || h2 != H2 || (lin3 && h3 != H3) || c_in > 128 || ldo < (lin3 ? h3 : h2) || (!vec && c_in > 32) ||
        !gnnpn_aligned(w1, 16) || !gnnpn_aligned(w2, 16) || (lin3 && !gnnpn_aligned(w3, 16)) || ldo % 4 != 0 || !gnnpn_aligned(out, 16) ||
        !gnnpn_aligned(inv1, 16) || !gnnpn_aligned(inv2, 16) || !gnnpn_aligned(inv3, 16) || !gnnpn_aligned(b1, 16) ||
        !gnnpn_aligned(bn1_scale, 16) || !gnnpn_aligned(bn1_shift, 16) || !gnnpn_aligned(b2, 16) || !gnnpn_aligned(bn2_scale, 16) ||
        !gnnpn_aligned(bn2_shift, 16) || !gnnpn_aligned(b3, 16))
        GNNPN_FAIL(GNNPN_E_UNSUP, "gin_layer_split: built for %d -> %d -> %d (-> %d), at most 128 input channels, rows of 4 k "
                   "channels 16-byte aligned or at most 32 channels, 16-byte aligned vectors and output rows (got %d -> %d -> %d%s, ldx %lld)", c_in, H1, H2, H3, c_in, h1,
                   h2, lin3 ? " -> h3" : "", (long long)ldx);
    const int k1a = (c_in + 31) / 32 * 32;
    const unsigned lds = (unsigned)((size_t)3 * BM * (H1 + 8) * 2 + 5 * BM * sizeof(float));   // k1a <= 128 < H1: every operand fits T's tile
    dim3 grid((unsigned)((n_rows + BM - 1) / BM)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define GNNPN_GIN_SPLIT(L3_)                                                                                                    \
    do {                                                                                                                        \
        if (hipFuncSetAttribute((const void*)gin_layer_split_kernel<L3_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != \
            hipSuccess)                                                                                                         \
            GNNPN_FAIL(GNNPN_E_LAUNCH, "gin_layer_split: cannot reserve %u B of LDS", lds);                                     \
        hipLaunchKernelGGL((gin_layer_split_kernel<L3_>), grid, block, lds, st, rowptr, col, x, ldx, c_in, eps,                 \
                           (const unsigned char*)w1, inv1, b1, bn1_scale, bn1_shift, (const unsigned char*)w2, inv2, b2, bn2_scale, \
                           bn2_shift, (const unsigned char*)w3, inv3, b3, out, ldo, n_rows, k1a, (int)vec);                             \
    } while (0)
    if (lin3) GNNPN_GIN_SPLIT(true);
    else GNNPN_GIN_SPLIT(false);
#undef GNNPN_GIN_SPLIT
    GNNPN_CHECK_LAUNCH("gin_layer_split");
    return GNNPN_OK;
}
