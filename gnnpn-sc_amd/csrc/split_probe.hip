// Entry points that expose the exact-split arithmetic of the recurrent kernels (coop_common.h) on its own, so that the
// tests can check it operand by operand: the three-piece decomposition (sum of the pieces == the operand, bit for bit) and
// the recurrent product W_hh.h built from it (against an exact dot product and against the fp32 fma chain).
// They run the SAME device functions the cooperative kernels inline — split3, split_weights, split_store, split_chain.
#include "common.h"
#include "coop_common.h"

namespace {
constexpr int H = 256;

__global__ __launch_bounds__(256) void split3_pieces_kernel(const float* __restrict__ x, int64_t n, int scale_log2,
                                                           uint16_t* __restrict__ p0, uint16_t* __restrict__ p1,
                                                           uint16_t* __restrict__ p2) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    _Float16 a, b, c;
    split3(__fmul_rn(x[i], ldexpf(1.0f, scale_log2)), a, b, c);
    p0[i] = f16_bits(a);
    p1[i] = f16_bits(b);
    p2[i] = f16_bits(c);
}

// gates[16][4H] = h[16][H] . W_hh^T for W_hh given in the packed recurrent layout: one workgroup per 32 hidden units
// (the ownership of a cooperative-group member: wave w, lane (c, kq) -> unit 32 m + 8 w + (c & 7), tiles [i | f], [g | o]).
// mode 0: fp32 MFMA chain (mfma_chain_pair, the parity path); mode 1: exact split (split_chain).
// Also returns the third weight pieces as the kernels hold them (bytes) and the per-column un-scaling factors.
template <bool SPLIT>
__global__ __launch_bounds__(256) void recurrent_product_kernel(const float* __restrict__ Wp, const float* __restrict__ h,
                                                                float* __restrict__ gates, float* __restrict__ col_inv) {
    __shared__ __attribute__((aligned(16))) float hs[3 * SPLIT_TILE / 2];
    __shared__ __attribute__((aligned(16))) unsigned wts[SPLIT ? SPLIT_WT_DWORDS : 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, c = lane & 15;
    const int member = blockIdx.x;
    const int unit = member * 32 + wave * 8 + (c & 7);
    int wrow[2] = {(0 + (c >> 3)) * H + unit, (2 + (c >> 3)) * H + unit};
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    if constexpr (SPLIT) {
        f16x8 w0[2][8], w1[2][8];
        float inv[2];
        unsigned* wt_lane = wts + (wave * 8 * 64 + lane) * 4;
        for (int tl = 0; tl < 2; ++tl) {
            inv[tl] = split_weights<H>(Wp, wrow[tl] / H, wrow[tl] % H, kq, w0[tl], w1[tl], wt_lane + 2 * tl);
            if (kq == 0) col_inv[wrow[tl]] = inv[tl];
        }
        for (int j = 0; j < 16; ++j) split_store(reinterpret_cast<_Float16*>(hs) + j * LDH16 + tid, h[j * H + tid]);
        __syncthreads();
        split_chain(reinterpret_cast<const _Float16*>(hs) + c * LDH16 + 8 * kq, w0, w1, wt_lane, inv, acc);
    } else {
        float wB[2][64];
        for (int tl = 0; tl < 2; ++tl)
            for (int kk = 0; kk < 64; ++kk)
                wB[tl][kk] = Wp[((size_t)(kk * 4 + wrow[tl] / H) * H + wrow[tl] % H) * 4 + kq];
        for (int j = 0; j < 16; ++j) hs[ht_index(j, tid)] = h[j * H + tid];
        __syncthreads();
        mfma_chain_pair<LDT, 16, true>(hs, c, kq, wB[0], wB[1], acc[0], acc[1]);
    }
    for (int tl = 0; tl < 2; ++tl)
        for (int r = 0; r < 4; ++r) gates[(size_t)(4 * kq + r) * (4 * H) + wrow[tl]] = acc[tl][r];
}
}  // namespace

extern "C" int gnnpn_split3_pieces_f32(const float* x, int64_t n, int32_t scale_log2, uint16_t* p0, uint16_t* p1, uint16_t* p2,
                                       void* stream) {
    GNNPN_REQUIRE(x && p0 && p1 && p2 && n >= 0 && scale_log2 >= -126 && scale_log2 <= 126, "split3_pieces: bad arguments");
    if (n == 0) return GNNPN_OK;
    hipLaunchKernelGGL(split3_pieces_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, n,
                       (int)scale_log2, p0, p1, p2);
    GNNPN_CHECK_LAUNCH("split3_pieces_f32");
    return GNNPN_OK;
}

extern "C" int gnnpn_recurrent_product_f32(const float* whh_packed, const float* h, int32_t precision, float* gates,
                                           float* col_inv, void* stream) {
    GNNPN_REQUIRE(whh_packed && h && gates, "recurrent_product: null operand");
    GNNPN_REQUIRE(precision == GNNPN_PREC_F32 || precision == GNNPN_PREC_SPLIT, "recurrent_product: precision must be GNNPN_PREC_F32 or GNNPN_PREC_SPLIT");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (precision == GNNPN_PREC_SPLIT) {
        GNNPN_REQUIRE(col_inv != nullptr, "recurrent_product: col_inv [4H] required for the split precision");
        hipLaunchKernelGGL(recurrent_product_kernel<true>, dim3(8), dim3(256), 0, s, whh_packed, h, gates, col_inv);
    } else {
        hipLaunchKernelGGL(recurrent_product_kernel<false>, dim3(8), dim3(256), 0, s, whh_packed, h, gates, col_inv);
    }
    GNNPN_CHECK_LAUNCH("recurrent_product_f32");
    return GNNPN_OK;
}

// ---- test hook: a stand-in for a collective's kernel beside the cooperative launches ------------------------------------
// An RCCL ring kernel at world size 8 holds a few dozen CUs' LDS and wave slots for tens of microseconds, every step, on a
// stream of its own — the one thing the placement of the cooperative kernels (coop_common.h: claims, seats, LDS positions)
// has never met on this one-GPU box, where RCCL at world size 1 is a copy.  `n_workgroups` workgroups of 256 threads each
// hold `lds_bytes` of LDS for `hold_us` microseconds (a sleeping spin on s_memrealtime: no memory traffic) and leave.
namespace {
__global__ __launch_bounds__(256) void lds_interferer_kernel(unsigned ticks, unsigned* __restrict__ sink) {
    extern __shared__ unsigned held[];
    held[threadIdx.x] = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (held[threadIdx.x] == 0xFFFFFFFFu && sink) *sink = 1u;      // keeps the allocation alive; never true
}
}  // namespace

extern "C" int gnnpn_debug_lds_interferer(int32_t n_workgroups, int32_t lds_bytes, int32_t hold_us, void* stream) {
    GNNPN_REQUIRE(n_workgroups > 0 && n_workgroups <= 4096 && lds_bytes >= 1024 && lds_bytes <= 160 * 1024 && hold_us >= 0 && hold_us <= 100000,
                  "debug_lds_interferer: bad arguments");
    if (hipFuncSetAttribute((const void*)lds_interferer_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
        GNNPN_FAIL(GNNPN_E_LAUNCH, "debug_lds_interferer: cannot reserve %d B of LDS", lds_bytes);
    hipLaunchKernelGGL(lds_interferer_kernel, dim3((unsigned)n_workgroups), dim3(256), (unsigned)lds_bytes, static_cast<hipStream_t>(stream),
                       (unsigned)hold_us * 100u, static_cast<unsigned*>(nullptr));
    GNNPN_CHECK_LAUNCH("debug_lds_interferer");
    return GNNPN_OK;
}

// ---- test hook: ONE v_mfma_f32_16x16x32_f16 on caller-given operands --------------------------------------------------------
// The a-priori error bound of the exact-split product (profiles/LOG_r01_r04.md section 12: <= 163 u sum|h w|) rests on HOW the matrix core
// accumulates its 32 products and C — groups of 8 consecutive k, operands aligned to the group's largest and truncated
// below 2^-24 of it, one round-to-nearest-even per group.  That model was probed (tools/probes/mfma_accum_model.hip), not
// read in a manual; its decisive cases are a -m gpu test through this entry so that another stepping or firmware that
// accumulates differently is caught by name.  A [16][32], B [32][16] (values representable in fp16), C, D [16][16], fp32.
namespace {
typedef _Float16 dbg_f16x8 __attribute__((ext_vector_type(8)));
typedef float dbg_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void debug_mfma_f16_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                            const float* __restrict__ C, float* __restrict__ D) {
    const int lane = threadIdx.x, c = lane & 15, kq = lane >> 4;
    dbg_f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[(4 * kq + r) * 16 + c];
    dbg_f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)A[c * 32 + 8 * kq + j];
        b[j] = (_Float16)B[(8 * kq + j) * 16 + c];
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * kq + r) * 16 + c] = acc[r];
}
}  // namespace

extern "C" int gnnpn_debug_mfma_f16(const float* A, const float* B, const float* C, float* D, void* stream) {
    GNNPN_REQUIRE(A && B && C && D, "debug_mfma_f16: null operand");
    hipLaunchKernelGGL(debug_mfma_f16_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), A, B, C, D);
    GNNPN_CHECK_LAUNCH("debug_mfma_f16");
    return GNNPN_OK;
}
