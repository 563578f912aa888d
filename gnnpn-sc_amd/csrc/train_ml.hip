// Training step of the GNN candidate-ranking model (trainML.py:34-47 over modelML.py:131-176):
// the pieces the inference kernels do not have — BatchNorm on batch statistics and its backward, the BCE-of-sigmoid loss
// and gradient, the embedding-table gradient, a dot product for GINConv's eps.  Everything else of the step reuses the
// inference kernels (gnnpn_linear_f32, gnnpn_csr_aggregate_f32 on the graph and on its transpose, gnnpn_segment_mean_f32)
// and the training kernels of the pointer network (gnnpn_gemm_f32, gnnpn_colsum_f32, gnnpn_adam_step_f32).
// The graphs of this stage are small (trainML.py:121: batches of two workflow graphs; 2 x S service rows): the kernels
// below are written for determinism (fixed summation orders) and coalescing, not tuned.
#include "common.h"

namespace {

constexpr int BN_COLS = 64;    // channels per workgroup
constexpr int BN_RG = 4;       // row groups per workgroup (256 threads)

// column reduce inside a workgroup of [BN_RG][BN_COLS] threads: partials of row groups added in group order
__device__ __forceinline__ float rg_sum(float v, float (*sm)[BN_COLS], int rg, int cl) {
    __syncthreads();
    sm[rg][cl] = v;
    __syncthreads();
    float s = sm[0][cl];
#pragma unroll
    for (int g = 1; g < BN_RG; ++g) s = __fadd_rn(s, sm[g][cl]);
    return s;
}

// BatchNorm1d, training mode (torch.nn.BatchNorm1d defaults): y = gamma * (x - mean) / sqrt(var_biased + eps) + beta
// [then ReLU]; running_mean / running_var move by `momentum` towards mean / the UNBIASED variance.
__global__ __launch_bounds__(256) void bn_train_forward_kernel(const float* __restrict__ x, int64_t rows, int cols,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float eps, float momentum, int relu, float* __restrict__ y,
                                                               float* __restrict__ xhat, float* __restrict__ invstd_out,
                                                               float* __restrict__ run_mean, float* __restrict__ run_var) {
    __shared__ float sm[BN_RG][BN_COLS];
    const int cl = threadIdx.x & (BN_COLS - 1), rg = threadIdx.x / BN_COLS;
    const int c = blockIdx.x * BN_COLS + cl;
    const bool live = c < cols;
    float s = 0.0f;
    if (live)
        for (int64_t r = rg; r < rows; r += BN_RG) s = __fadd_rn(s, x[r * cols + c]);
    const float mean = rg_sum(s, sm, rg, cl) / (float)rows;
    float q = 0.0f;
    if (live)
        for (int64_t r = rg; r < rows; r += BN_RG) {
            const float d = __fsub_rn(x[r * cols + c], mean);
            q = fmaf(d, d, q);
        }
    const float var = rg_sum(q, sm, rg, cl) / (float)rows;
    const float invstd = 1.0f / sqrtf(__fadd_rn(var, eps));
    if (live) {
        const float g = gamma[c], b = beta[c];
        for (int64_t r = rg; r < rows; r += BN_RG) {
            const float xh = __fmul_rn(__fsub_rn(x[r * cols + c], mean), invstd);
            xhat[r * cols + c] = xh;
            float v = fmaf(xh, g, b);
            if (relu) v = v < 0.0f ? 0.0f : v;
            y[r * cols + c] = v;
        }
        if (rg == 0) {
            invstd_out[c] = invstd;
            if (run_mean) {
                const float unbiased = rows > 1 ? var * ((float)rows / (float)(rows - 1)) : var;
                run_mean[c] = fmaf(momentum, __fsub_rn(mean, run_mean[c]), run_mean[c]);
                run_var[c] = fmaf(momentum, __fsub_rn(unbiased, run_var[c]), run_var[c]);
            }
        }
    }
}

// backward of the above: dy is the gradient wrt the (post-ReLU) output y; dx, dgamma, dbeta
__global__ __launch_bounds__(256) void bn_train_backward_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                const float* __restrict__ xhat, const float* __restrict__ gamma,
                                                                const float* __restrict__ invstd, int64_t rows, int cols,
                                                                int relu, float* __restrict__ dx, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta) {
    __shared__ float sm[BN_RG][BN_COLS];
    const int cl = threadIdx.x & (BN_COLS - 1), rg = threadIdx.x / BN_COLS;
    const int c = blockIdx.x * BN_COLS + cl;
    const bool live = c < cols;
    float sb = 0.0f, sg = 0.0f;
    if (live)
        for (int64_t r = rg; r < rows; r += BN_RG) {
            const int64_t i = r * cols + c;
            const float d = (relu && !(y[i] > 0.0f)) ? 0.0f : dy[i];
            sb = __fadd_rn(sb, d);
            sg = fmaf(d, xhat[i], sg);
        }
    const float db = rg_sum(sb, sm, rg, cl);
    const float dg = rg_sum(sg, sm, rg, cl);
    if (live) {
        const float k = __fmul_rn(gamma[c], invstd[c]), inv_n = 1.0f / (float)rows;
        for (int64_t r = rg; r < rows; r += BN_RG) {
            const int64_t i = r * cols + c;
            const float d = (relu && !(y[i] > 0.0f)) ? 0.0f : dy[i];
            dx[i] = __fmul_rn(k, __fsub_rn(__fsub_rn(d, __fmul_rn(db, inv_n)), __fmul_rn(xhat[i], __fmul_rn(dg, inv_n))));
        }
        if (rg == 0) {
            dgamma[c] = dg;
            dbeta[c] = db;
        }
    }
}

// block-wide sum in a fixed order (1024 threads): strided partials, butterfly inside a wave, waves ascending
__device__ __forceinline__ double block_sum_1024(double v, double* sm) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int lo = __shfl_xor((int)(__double_as_longlong(v) & 0xffffffffll), off, 64);
        const int hi = __shfl_xor((int)(__double_as_longlong(v) >> 32), off, 64);
        v += __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = sm[0];
    for (int w = 1; w < 16; ++w) s += sm[w];
    return s;
}

// BCELoss(mean) of p = sigmoid(z) and the gradient wrt z as autograd forms it: BCELoss' backward
// (p - y) / max(p (1 - p), 1e-12) / n, then the sigmoid's p (1 - p); log terms clamped at -100 (torch.nn.BCELoss)
__global__ __launch_bounds__(1024) void bce_sigmoid_kernel(const float* __restrict__ p, const float* __restrict__ y, int64_t n,
                                                           float* __restrict__ dz, float* __restrict__ loss) {
    __shared__ double sm[16];
    double acc = 0.0;
    const float inv_n = 1.0f / (float)n;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float pi = p[i], yi = y[i];
        const float lp = fmaxf(logf(pi), -100.0f), lq = fmaxf(logf(__fsub_rn(1.0f, pi)), -100.0f);
        acc += (double)(-(yi * lp + (1.0f - yi) * lq));
        const float pq = __fmul_rn(__fsub_rn(1.0f, pi), pi);
        const float gp = __fmul_rn(__fsub_rn(pi, yi) / fmaxf(pq, 1e-12f), inv_n);
        dz[i] = __fmul_rn(gp, pq);
    }
    const double s = block_sum_1024(acc, sm);
    if (threadIdx.x == 0) loss[0] = (float)(s / (double)n);
}

__global__ __launch_bounds__(1024) void dot_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                   float* __restrict__ out) {
    __shared__ double sm[16];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) acc += (double)a[i] * (double)b[i];
    const double s = block_sum_1024(acc, sm);
    if (threadIdx.x == 0) out[0] = (float)s;
}

// d table[v][:] = sum over rows n with id(n) == v of dh[n][0:c], rows ascending (index_add order); id = x[n*ldx] as float
__global__ __launch_bounds__(64) void embed_grad_kernel(const float* __restrict__ dh, int64_t ldh, const float* __restrict__ x,
                                                        int64_t ldx, int64_t rows, int c, float* __restrict__ dtable) {
    const int v = blockIdx.x;
    for (int ch = threadIdx.x; ch < c; ch += 64) {
        float s = 0.0f;
        for (int64_t n = 0; n < rows; ++n)
            if ((int)x[n * ldx] == v) s = __fadd_rn(s, dh[n * ldh + ch]);
        dtable[(int64_t)v * c + ch] = s;
    }
}

}  // namespace

extern "C" int gnnpn_bn_train_forward_f32(const float* x, int64_t rows, int32_t cols, const float* gamma, const float* beta,
                                          float eps, float momentum, int relu, float* y, float* xhat, float* invstd,
                                          float* running_mean, float* running_var, void* stream) {
    GNNPN_REQUIRE(x && gamma && beta && y && xhat && invstd, "bn_train_forward: null argument");
    GNNPN_REQUIRE(rows >= 1 && cols >= 1, "bn_train_forward: rows >= 1, cols >= 1");
    GNNPN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_forward: both running buffers or neither");
    hipLaunchKernelGGL(bn_train_forward_kernel, dim3((cols + BN_COLS - 1) / BN_COLS), dim3(256), 0, (hipStream_t)stream, x, rows,
                       cols, gamma, beta, eps, momentum, relu, y, xhat, invstd, running_mean, running_var);
    GNNPN_CHECK_LAUNCH("bn_train_forward_f32");
    return GNNPN_OK;
}

extern "C" int gnnpn_bn_train_backward_f32(const float* dy, const float* y, const float* xhat, const float* gamma,
                                           const float* invstd, int64_t rows, int32_t cols, int relu, float* dx, float* dgamma,
                                           float* dbeta, void* stream) {
    GNNPN_REQUIRE(dy && xhat && gamma && invstd && dx && dgamma && dbeta, "bn_train_backward: null argument");
    GNNPN_REQUIRE(!relu || y, "bn_train_backward: the ReLU mask needs y");
    GNNPN_REQUIRE(rows >= 1 && cols >= 1, "bn_train_backward: rows >= 1, cols >= 1");
    hipLaunchKernelGGL(bn_train_backward_kernel, dim3((cols + BN_COLS - 1) / BN_COLS), dim3(256), 0, (hipStream_t)stream, dy, y,
                       xhat, gamma, invstd, rows, cols, relu, dx, dgamma, dbeta);
    GNNPN_CHECK_LAUNCH("bn_train_backward_f32");
    return GNNPN_OK;
}

extern "C" int gnnpn_bce_sigmoid_f32(const float* p, const float* y, int64_t n, float* dz, float* loss, void* stream) {
    GNNPN_REQUIRE(p && y && dz && loss && n >= 1, "bce_sigmoid: null argument or n < 1");
    hipLaunchKernelGGL(bce_sigmoid_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, p, y, n, dz, loss);
    GNNPN_CHECK_LAUNCH("bce_sigmoid_f32");
    return GNNPN_OK;
}

extern "C" int gnnpn_dot_f32(const float* a, const float* b, int64_t n, float* out, void* stream) {
    GNNPN_REQUIRE(a && b && out && n >= 0, "dot: null argument");
    hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, b, n, out);
    GNNPN_CHECK_LAUNCH("dot_f32");
    return GNNPN_OK;
}

extern "C" int gnnpn_embed_grad_f32(const float* dh, int64_t ldh, const float* x, int64_t ldx, int64_t rows, int32_t c,
                                    int32_t vocab, float* dtable, void* stream) {
    GNNPN_REQUIRE(dh && x && dtable, "embed_grad: null argument");
    GNNPN_REQUIRE(rows >= 0 && c >= 1 && vocab >= 1 && ldh >= c && ldx >= 1, "embed_grad: bad shape");
    hipLaunchKernelGGL(embed_grad_kernel, dim3(vocab), dim3(64), 0, (hipStream_t)stream, dh, ldh, x, ldx, rows, c, dtable);
    GNNPN_CHECK_LAUNCH("embed_grad_f32");
    return GNNPN_OK;
}
