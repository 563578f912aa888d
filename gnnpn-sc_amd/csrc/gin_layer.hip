// One GIN layer of the workflow branch for LARGE graphs, in one launch per layer: neighbour aggregate -> Linear + BN + ReLU
// -> Linear + BN + ReLU [-> nodeLin], the [rows x 256] intermediate never leaving the CU.
// Replaces, per layer, gnnpn_csr_aggregate_f32 + 2 x gnnpn_linear_f32 (and the trailing nodeLin call) of the layered path
// (modelML.py:75-93,139-143,165): at the 1000-task shape (512 problems x 1001 nodes) those five dense layers and two
// aggregates were 2.1 ms of a 23 ms step, 4.5 GB of HBM traffic for 124 GFLOP — each layer wrote a [512k x 256] intermediate
// (525 MB) and read it back.  (Graphs of <= 16 nodes take the one-launch branch of request_branch.hip instead.)
//
// A workgroup of 4 wavefronts owns 32 consecutive rows (three workgroups per CU); a wavefront owns tiles of 32 output columns:
//   0. aggregate: (1 + eps) * x_i + sum_j x_j in CSR order, rows gathered from global memory (chain graphs: the neighbours
//      are the adjacent rows, in L2), into LDS k-major;
//   1. [32 x C_in] . W1^T on v_mfma_f32_32x32x2_f32 — the A operand from LDS, the weights PRE-PACKED as the matrix core's
//      B-fragments (a one-time layout change at weight-load time, ops.pack_mfma_b32) so that a lane's operand of every MFMA is
//      one coalesced 4-byte load straight into a register, 16 k-pairs ahead: no weight staging through LDS, no barrier inside
//      a stage — + bias, BN, ReLU -> T [32 x 256] in LDS, k-major;
//   2. T . W2^T, + bias, BN, ReLU -> global (or, with the third layer, -> LDS);
//   3. (optional) . W3^T + bias -> global: nodeLin fused behind the last GIN layer.
// (First form, measured and replaced: 64-row tiles, weight tiles of 32 k staged global -> registers -> LDS with two barriers
// per tile, one workgroup per CU (133 KB of LDS): 2.33 ms for the branch at the 1000-task shape against 2.20 ms layered; the
// same on 32-row tiles with two workgroups per CU: 1.71 ms.)
// Every product is the k-ordered fp32 fma chain of the matrix core, zero-padded to whole 32-k tiles exactly as
// linear_f32_kernel pads, and the aggregate and epilogues round as csr_aggregate_kernel / linear_f32_kernel round:
// the layer's output is BIT-IDENTICAL to the layered kernels' (tests/test_gpu_ops.py).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
struct __attribute__((packed, aligned(4))) f4a4 { float v[4]; };

constexpr int BM = 32, LDT = BM + 1, BK = 32, H1 = 256, H2 = 128, H3 = 128;
constexpr int NCH_MAX = 8;                                // aggregate: float4 chunks per lane (256 channels / 8 lanes / 4)
constexpr int PF = 16;                                    // weight fragments a lane keeps in flight (k-pairs ahead of the MFMA that uses them)

// acc = A[32 x K] . W^T for this wave's tile of 32 output columns.  A: k-major in LDS (row stride LDT).  W: PACKED as the
// B-fragments of v_mfma_f32_32x32x2_f32 (ops.pack_mfma_b32: packed[column tile][k pair][lane] = W[32 t + lane % 32][2 kp + lane / 32],
// K zero-padded to whole tiles of 32 as linear_f32_kernel pads) — a lane's operand of every MFMA is ONE coalesced 4-byte load
// straight into a register, PF k-pairs ahead: no staging of the weights through LDS, no barrier inside a stage.  `wp`: this
// wave's column tile, this lane's word; kp_n = K_padded / 2 (a multiple of PF).
__device__ __forceinline__ void gemm_from_lds(const float* __restrict__ As, const float* __restrict__ wp, int kp_n, int half, int l32,
                                              f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float bq[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) bq[i] = wp[i * 64];
    const float* ap = As + half * LDT + l32;
    for (int kp0 = 0; kp0 < kp_n; kp0 += PF) {
        const bool more = kp0 + PF < kp_n;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const float a = ap[(kp0 + i) * 2 * LDT];
            const float b = bq[i];
            if (more) bq[i] = wp[(kp0 + PF + i) * 64];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
}

// epilogue of 32 columns (from col0): act((acc + bias) * scale + shift) -> the next stage's A operand (LDS, k-major) or global
template <bool TO_LDS>
__device__ __forceinline__ void epilogue(const f32x16& acc, const float* __restrict__ bias, const float* __restrict__ scale,
                                         const float* __restrict__ shift, int act, int col0, int half, int l32,
                                         float* __restrict__ Ts, float* __restrict__ out, int64_t ldo, int64_t m0, int64_t M) {
    const int col = col0 + l32;
    const float bv = bias ? bias[col] : 0.0f;
    const float sc = scale ? scale[col] : 1.0f;
    const float sh = shift ? shift[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[r];
        if (bias) v = __fadd_rn(v, bv);
        if (scale) v = __fadd_rn(__fmul_rn(v, sc), sh);
        v = apply_act(v, act);
        if (TO_LDS) Ts[col * LDT + row] = v;
        else if (m0 + row < M) out[(m0 + row) * ldo + col] = v;
    }
}

// Three workgroups per CU (51 KB of LDS, <= 168 registers): one's aggregate and epilogues run under the others' MFMAs.
template <bool LIN3>
__global__ __launch_bounds__(256, 3) void gin_layer_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ x, int64_t ldx, int32_t c_in,
    const float* __restrict__ eps, const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ a1,
    const float* __restrict__ s1, const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ a2,
    const float* __restrict__ s2, const float* __restrict__ w3, const float* __restrict__ b3, float* __restrict__ out, int64_t ldo,
    int64_t M, int32_t k1a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                     // [max(k1a, 128)][LDT]: the aggregate; later the third stage's A operand
    float* Ts = As + (size_t)(k1a > 128 ? k1a : 128) * LDT;   // [256][LDT]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    // ---- 0. aggregate: 8 lanes per row; a lane's channels: chunks of 4 dealt round-robin (or single channels when rows are not
    // 16-byte aligned); the edge loop is the OUTER one, so that a lane has all its chunks of a neighbour row in flight at once
    {
        const int r = tid >> 3, sub = tid & 7;
        const int64_t row = m0 + r;
        const float one_plus_eps = __fadd_rn(1.0f, *eps);
        const bool vec = (c_in % 4 == 0) && (ldx % 4 == 0) && c_in <= 32 * NCH_MAX;
        int e0 = 0, e1 = 0;
        if (row < M) {
            e0 = rowptr[row];
            e1 = rowptr[row + 1];
        }
        if (vec) {
            float4 acc[NCH_MAX];
#pragma unroll
            for (int i = 0; i < NCH_MAX; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int nch = (c_in / 4 - sub + 7) / 8;     // this lane's chunks: 4 * (sub + 8 i) < c_in
            for (int e = e0; e <= e1; ++e) {              // e == e1: the row itself, scaled by 1 + eps, AFTER the neighbours
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
#pragma unroll
            for (int i = 0; i < NCH_MAX; ++i) {
                const int c = 4 * (sub + 8 * i);
                if (c < k1a) {                            // (chunks beyond c_in are the zero padding of the last k-tile)
                    As[(c + 0) * LDT + r] = acc[i].x;
                    As[(c + 1) * LDT + r] = acc[i].y;
                    As[(c + 2) * LDT + r] = acc[i].z;
                    As[(c + 3) * LDT + r] = acc[i].w;
                }
            }
        } else {
            for (int c = sub; c < k1a; c += 8) {
                float acc = 0.f;
                if (row < M && c < c_in) {
                    for (int e = e0; e < e1; ++e) acc = __fadd_rn(acc, x[(int64_t)col[e] * ldx + c]);
                    acc = __fadd_rn(acc, __fmul_rn(one_plus_eps, x[row * ldx + c]));
                }
                As[c * LDT + r] = acc;
            }
        }
    }
    f32x16 acc;
    __syncthreads();                                      // the aggregate is in LDS
    // ---- 1. Linear(c_in -> 256) + BN + ReLU -> Ts: 8 column tiles of 32, two per wavefront
#pragma unroll 1
    for (int p = 0; p < H1 / 128; ++p) {
        const int ct = p * 4 + wave;
        gemm_from_lds(As, w1 + (size_t)ct * (k1a / 2) * 64 + lane, k1a / 2, half, l32, acc);
        epilogue<true>(acc, b1, a1, s1, GNNPN_ACT_RELU, ct * 32, half, l32, Ts, nullptr, 0, m0, M);
    }
    __syncthreads();                                      // T is complete (and nobody reads the aggregate any more)
    // ---- 2. Linear(256 -> 128) + BN + ReLU -> global, or -> As for the third stage
    gemm_from_lds(Ts, w2 + (size_t)wave * (H1 / 2) * 64 + lane, H1 / 2, half, l32, acc);
    epilogue<LIN3>(acc, b2, a2, s2, GNNPN_ACT_RELU, wave * 32, half, l32, As, out, ldo, m0, M);
    // ---- 3. nodeLin: Linear(128 -> 128) -> global
    if (LIN3) {
        __syncthreads();
        gemm_from_lds(As, w3 + (size_t)wave * (H2 / 2) * 64 + lane, H2 / 2, half, l32, acc);
        epilogue<false>(acc, b3, nullptr, nullptr, GNNPN_ACT_NONE, wave * 32, half, l32, nullptr, out, ldo, m0, M);
    }
}

}  // namespace

extern "C" int gnnpn_gin_layer_f32(const int32_t* rowptr, const int32_t* col, const float* x, int64_t ldx, int32_t c_in, const float* eps,
                                   const float* w1, const float* b1, const float* bn1_scale, const float* bn1_shift, int32_t h1,
                                   const float* w2, const float* b2, const float* bn2_scale, const float* bn2_shift, int32_t h2,
                                   const float* w3, const float* b3, int32_t h3, float* out, int64_t ldo, int64_t n_rows, void* stream) {
    GNNPN_REQUIRE(n_rows >= 0 && c_in > 0 && ldx >= c_in, "gin_layer: bad shape");
    if (n_rows == 0) return GNNPN_OK;
    GNNPN_REQUIRE(rowptr && x && eps && w1 && w2 && out, "gin_layer: null operand");                 // col may be NULL for a graph without edges (csr_aggregate allows it too)
    GNNPN_REQUIRE((bn1_scale == nullptr) == (bn1_shift == nullptr) && (bn2_scale == nullptr) == (bn2_shift == nullptr),
                  "gin_layer: scale and shift go together");
    GNNPN_REQUIRE(x != out, "gin_layer: in-place layers are not supported");
    const bool lin3 = w3 != nullptr;
    if (h1 != H1 || h2 != H2 || (lin3 && h3 != H3) || c_in > 256 || ldo < (lin3 ? h3 : h2) ||
        !gnnpn_aligned(x, 16) || !gnnpn_aligned(w1, 4) || !gnnpn_aligned(w2, 4) || ((c_in + BK - 1) / BK * BK / 2) % PF != 0)
        GNNPN_FAIL(GNNPN_E_UNSUP, "gin_layer: built for %d -> %d -> %d (-> %d) with at most 256 input channels (got %d -> %d -> %d%s)", c_in, H1,
                   H2, H3, c_in, h1, h2, lin3 ? " -> h3" : "");
    const int k1a = (c_in + BK - 1) / BK * BK;
    const unsigned lds = (unsigned)(((size_t)(k1a > 128 ? k1a : 128) * LDT + (size_t)H1 * LDT) * sizeof(float));
    dim3 grid((unsigned)((n_rows + BM - 1) / BM)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define GNNPN_GIN_LAYER(L3_)                                                                                                      \
    do {                                                                                                                        \
        if (hipFuncSetAttribute((const void*)gin_layer_kernel<L3_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            GNNPN_FAIL(GNNPN_E_LAUNCH, "gin_layer: cannot reserve %u B of LDS", lds);                                           \
        hipLaunchKernelGGL((gin_layer_kernel<L3_>), grid, block, lds, st, rowptr, col, x, ldx, c_in, eps, w1, b1, bn1_scale,    \
                           bn1_shift, w2, b2, bn2_scale, bn2_shift, w3, b3, out, ldo, n_rows, k1a);                             \
    } while (0)
    if (lin3) GNNPN_GIN_LAYER(true);
    else GNNPN_GIN_LAYER(false);
#undef GNNPN_GIN_LAYER
    GNNPN_CHECK_LAUNCH("gin_layer_f32");
    return GNNPN_OK;
}
