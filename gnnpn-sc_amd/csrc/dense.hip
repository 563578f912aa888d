// Dense fp32 layers on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, a k-ordered
// fmaf chain per output element) + the embedding/concat feeder.
//
// Tile: BM x BN x 32 per 256-thread workgroup (4 waves as 2x2, each wave (BM/2)x(BN/2) made of
// 32x32 MFMA tiles).  Both operands are "row-major with k contiguous" (A[M,K], W[N,K]); tiles go
// global -> registers -> LDS transposed to k-major so that an MFMA fragment read is 32
// consecutive floats per half-wave (conflict free), with the next tile's global loads issued
// before the current tile's MFMAs (register double buffering).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BM, int BN>
struct DenseCfg {
    static constexpr int BK = 32;
    static constexpr int LDA = BM + 1;   // odd leading dimension: transposed stores <= 2-way conflicts
    static constexpr int LDB = BN + 1;
    static constexpr int A_F4 = BM * BK / 4 / 256;   // float4 loads per thread per tile
    static constexpr int B_F4 = BN * BK / 4 / 256;
    static constexpr int TM = BM / 64;               // 32x32 tiles per wave along M
    static constexpr int TN = BN / 64;
};

struct __attribute__((packed, aligned(4))) f4_align4 { float v[4]; };

// load a [ROWS x 32] tile (rows r0.., k from k0) of a row-major matrix into registers
template <int NF4>
__device__ __forceinline__ void tile_load(const float* __restrict__ P, int64_t ld, int64_t r0,
                                          int64_t n_rows, int k0, int K, bool vec_ok, float4 (&reg)[NF4]) {
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
        int f = threadIdx.x + i * 256;   // float4 slot: row = f / 8, kc = f % 8
        int r = f >> 3, kc = f & 7;
        int64_t row = r0 + r;
        int k = k0 + kc * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < n_rows) {
            const float* src = P + row * ld + k;
            if (vec_ok && k + 3 < K) {          // one global_load_dwordx4: needs 4-byte alignment only (rows of 135 floats too)
                const f4_align4 t = *reinterpret_cast<const f4_align4*>(src);
                v = make_float4(t.v[0], t.v[1], t.v[2], t.v[3]);
            } else {
                if (k + 0 < K) v.x = src[0];
                if (k + 1 < K) v.y = src[1];
                if (k + 2 < K) v.z = src[2];
                if (k + 3 < K) v.w = src[3];
            }
        }
        reg[i] = v;
    }
}

template <int NF4, int LD>
__device__ __forceinline__ void tile_store(float* S, const float4 (&reg)[NF4]) {
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
        int f = threadIdx.x + i * 256;
        int r = f >> 3, kc = f & 7;
        S[(kc * 4 + 0) * LD + r] = reg[i].x;
        S[(kc * 4 + 1) * LD + r] = reg[i].y;
        S[(kc * 4 + 2) * LD + r] = reg[i].z;
        S[(kc * 4 + 3) * LD + r] = reg[i].w;
    }
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void linear_f32_kernel(
    const float* __restrict__ A, int64_t lda, const float* __restrict__ W, int64_t ldw,
    const float* __restrict__ bias, const float* __restrict__ scale, const float* __restrict__ shift,
    int act, float* __restrict__ C, int64_t ldc, int64_t M, int N, int K, int vec_a, int vec_w) {
    using Cfg = DenseCfg<BM, BN>;
    __shared__ float As[Cfg::BK * Cfg::LDA];
    __shared__ float Bs[Cfg::BK * Cfg::LDB];

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;

    f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 ra[Cfg::A_F4], rb[Cfg::B_F4];
    tile_load<Cfg::A_F4>(A, lda, m0, M, 0, K, vec_a, ra);
    tile_load<Cfg::B_F4>(W, ldw, n0, N, 0, K, vec_w, rb);

    const int half = lane >> 5, l32 = lane & 31;
    for (int k0 = 0; k0 < K; k0 += Cfg::BK) {
        tile_store<Cfg::A_F4, Cfg::LDA>(As, ra);
        tile_store<Cfg::B_F4, Cfg::LDB>(Bs, rb);
        __syncthreads();
        if (k0 + Cfg::BK < K) {   // prefetch the next tile while this one is multiplied
            tile_load<Cfg::A_F4>(A, lda, m0, M, k0 + Cfg::BK, K, vec_a, ra);
            tile_load<Cfg::B_F4>(W, ldw, n0, N, k0 + Cfg::BK, K, vec_w, rb);
        }
#pragma unroll
        for (int kk = 0; kk < Cfg::BK; kk += 2) {
            float a[Cfg::TM], b[Cfg::TN];
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
                a[i] = As[(kk + half) * Cfg::LDA + wm * (BM / 2) + i * 32 + l32];
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j)
                b[j] = Bs[(kk + half) * Cfg::LDB + wn * (BN / 2) + j * 32 + l32];
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue: D[row][col], col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
        const int col = n0 + wn * (BN / 2) + j * 32 + l32;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.0f;
        const float sc = scale ? scale[col] : 1.0f;
        const float sh = shift ? shift[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row >= M) continue;
                float v = acc[i][j][r];
                if (bias) v = __fadd_rn(v, bv);
                if (scale) v = __fadd_rn(__fmul_rn(v, sc), sh);
                C[row * ldc + col] = apply_act(v, act);
            }
        }
    }
}

extern "C" int gnnpn_linear_f32(const float* A, int64_t lda, const float* W, int64_t ldw,
                                const float* bias, const float* scale, const float* shift, int act,
                                float* C, int64_t ldc, int64_t M, int N, int K, void* stream) {
    GNNPN_REQUIRE(M >= 0 && N > 0 && K > 0, "linear: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    if (M == 0) return GNNPN_OK;                    // empty batch: its buffers may be NULL
    GNNPN_REQUIRE(A && W && C, "linear: null operand");
    GNNPN_REQUIRE(lda >= K && ldw >= K && ldc >= N, "linear: leading dimension too small");
    GNNPN_REQUIRE((scale == nullptr) == (shift == nullptr), "linear: scale and shift go together");
    GNNPN_REQUIRE(act >= 0 && act <= 2, "linear: unknown activation %d", act);
    if (M == 0) return GNNPN_OK;
    const int vec_a = gnnpn_aligned(A, 4), vec_w = gnnpn_aligned(W, 4);   // 16-byte loads at 4-byte alignment (tile_load)
    hipStream_t s = (hipStream_t)stream;
    // 128x128 tiles only for deep, wide products; everything this path runs (K <= 260) is faster on 64x64 tiles — 7 resident
    // workgroups per CU instead of 2 hide the four short k-tiles' latencies: [512k x 128] x [128 x 128] 301 -> 222 us,
    // [1M x 256] x [256 x 256] 1.62 -> 1.51 ms, the score product [1024 x 256] x [256 x 4056] 37 -> 33 us (measured, MI355X)
    const int64_t blocks128 = ((M + 127) / 128) * ((N + 127) / 128);
    if (blocks128 >= 256 && K >= 512 && N >= 256) {
        dim3 grid((N + 127) / 128, (unsigned)((M + 127) / 128));
        GNNPN_REQUIRE(grid.y < 65536u * 32768u, "linear: M too large");
        hipLaunchKernelGGL((linear_f32_kernel<128, 128>), grid, dim3(256), gnnpn_front_lds_pad((const void*)linear_f32_kernel<128, 128>), s, A, lda, W, ldw, bias, scale,
                           shift, act, C, ldc, M, N, K, vec_a, vec_w);
    } else if (N % 128 == 0 && M >= 64 * 512) {   // full-width 64x128 tiles: A is read once per 128 columns (5-15 % over 64x64)
        dim3 grid(N / 128, (unsigned)((M + 63) / 64));
        hipLaunchKernelGGL((linear_f32_kernel<64, 128>), grid, dim3(256), gnnpn_front_lds_pad((const void*)linear_f32_kernel<64, 128>), s, A, lda, W, ldw, bias, scale,
                           shift, act, C, ldc, M, N, K, vec_a, vec_w);
    } else {
        dim3 grid((N + 63) / 64, (unsigned)((M + 63) / 64));
        hipLaunchKernelGGL((linear_f32_kernel<64, 64>), grid, dim3(256), gnnpn_front_lds_pad((const void*)linear_f32_kernel<64, 64>), s, A, lda, W, ldw, bias, scale,
                           shift, act, C, ldc, M, N, K, vec_a, vec_w);
    }
    GNNPN_CHECK_LAUNCH("linear_f32");
    return GNNPN_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ void embed_concat_kernel(const float* __restrict__ x, const float* __restrict__ table, int vocab,
                                    int emb, int nfeat, float* __restrict__ out, int64_t n_rows) {
    const int width = emb + nfeat;
    const int64_t total = n_rows * width;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / width;
        const int c = (int)(i - n * width);
        float v;
        if (c < emb) {
            const int id = (int)x[n * (1 + nfeat)];   // .long() truncation (modelML.py:134)
            v = (id >= 0 && id < vocab) ? table[(int64_t)id * emb + c] : __int_as_float(0x7fc00000);
        } else {
            v = x[n * (1 + nfeat) + 1 + (c - emb)];
        }
        out[i] = v;
    }
}

extern "C" int gnnpn_embed_concat_f32(const float* x, const float* table, int vocab, int emb, int nfeat,
                                      float* out, int64_t n_rows, void* stream) {
    GNNPN_REQUIRE(vocab > 0 && emb > 0 && nfeat >= 0 && n_rows >= 0, "embed_concat: bad shape");
    if (n_rows == 0) return GNNPN_OK;
    GNNPN_REQUIRE(x && table && out, "embed_concat: null operand");
    const int64_t total = n_rows * (emb + nfeat);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(embed_concat_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, table, vocab,
                       emb, nfeat, out, n_rows);
    GNNPN_CHECK_LAUNCH("embed_concat_f32");
    return GNNPN_OK;
}

// ---------------------------------------------------------------------------------------------
// General fp32 GEMM for the training path (weight gradients and the gradient of the input projection):
//   C[m,n] = sum_k Aop[m,k] * Bop[n,k],   Aop[m,k] = a_kmajor ? A[k*lda + m] : A[m*lda + k]  (same for B)
// so that dW = dG^T . X (both operands k-major: the reduction runs over the B*L rows of the saved activations) and
// dX = dG . W (B given k-major) need no transposed copies.  Same 64x64x32 tile, LDS layout and k-ordered fp32 MFMA chain as
// linear_f32_kernel; a k-major operand is already in the LDS layout and is copied straight in.
template <int BM, bool KMAJOR>
__device__ __forceinline__ void gemm_tile_to_lds(const float* __restrict__ P, int64_t ld, int64_t r0, int64_t n_rows, int k0,
                                                 int K, float* S) {
    constexpr int LD = BM + 1;
    if constexpr (KMAJOR) {   // P[k*ld + r]: 32 k-rows of BM consecutive floats
        for (int f = threadIdx.x; f < 32 * BM; f += 256) {
            const int kk = f / BM, r = f - kk * BM;
            const int64_t row = r0 + r;
            const int k = k0 + kk;
            S[kk * LD + r] = (row < n_rows && k < K) ? P[(int64_t)k * ld + row] : 0.0f;
        }
    } else {                  // P[r*ld + k]
        for (int f = threadIdx.x; f < 32 * BM; f += 256) {
            const int r = f >> 5, kk = f & 31;
            const int64_t row = r0 + r;
            const int k = k0 + kk;
            S[kk * LD + r] = (row < n_rows && k < K) ? P[row * ld + k] : 0.0f;
        }
    }
}

template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm,
                                                       int64_t ldb, float* __restrict__ C, int64_t ldc, int64_t M, int N,
                                                       int K, int k_chunk) {
    constexpr int BM = 64, BN = 64;
    __shared__ float As[32 * (BM + 1)];
    __shared__ float Bs[32 * (BN + 1)];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int half = lane >> 5, l32 = lane & 31;
    // split-K: slice blockIdx.z reduces k in [z*k_chunk, (z+1)*k_chunk) into its own partial matrix C[z] (the caller sums the
    // partials in slice order: deterministic, no atomics) — the weight-gradient GEMMs have K = B*L >> M, N
    const int k_begin = blockIdx.z * k_chunk, k_end = min(K, k_begin + k_chunk);
    C += (int64_t)blockIdx.z * M * ldc;
    for (int k0 = k_begin; k0 < k_end; k0 += 32) {
        gemm_tile_to_lds<BM, A_KMAJOR>(A, lda, m0, M, k0, k_end, As);
        gemm_tile_to_lds<BN, B_KMAJOR>(Bm, ldb, n0, N, k0, k_end, Bs);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            const float a = As[(kk + half) * (BM + 1) + wm * 32 + l32];
            const float b = Bs[(kk + half) * (BN + 1) + wn * 32 + l32];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int col = n0 + wn * 32 + l32;
    if (col >= N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < M) C[row * ldc + col] = acc[r];
    }
}

extern "C" int gnnpn_gemm_f32(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb, int b_kmajor, float* C,
                              int64_t ldc, int64_t M, int N, int K, int split_k, void* stream) {
    GNNPN_REQUIRE(M >= 0 && N > 0 && K > 0 && ldc >= N, "gemm: bad shape M=%lld N=%d K=%d", (long long)M, N, K);
    if (M == 0) return GNNPN_OK;
    GNNPN_REQUIRE(A && B && C, "gemm: null operand");
    GNNPN_REQUIRE(lda >= (a_kmajor ? M : K) && ldb >= (b_kmajor ? N : K), "gemm: leading dimension too small");
    GNNPN_REQUIRE(split_k >= 1 && split_k <= 1024, "gemm: split_k must be 1..1024");
    if (M == 0) return GNNPN_OK;
    const int k_chunk = ((K + split_k - 1) / split_k + 31) / 32 * 32;
    dim3 grid((N + 63) / 64, (unsigned)((M + 63) / 64), (unsigned)split_k);
    hipStream_t s = (hipStream_t)stream;
    if (a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K, k_chunk);
    else if (a_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K, k_chunk);
    else if (b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K, k_chunk);
    else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, K, k_chunk);
    GNNPN_CHECK_LAUNCH("gemm_f32");
    return GNNPN_OK;
}
