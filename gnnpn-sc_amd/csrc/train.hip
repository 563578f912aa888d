// REINFORCE training step of the High-level pointer network (reference
// src/models/trainPNHigh.py:76-112 driving src/models/modelPN.py:175-306): the kernels of the backward pass.
//
// The picks of a training step are drawn by the sampled inference forward (decode kernels, gnnpn_decode_net_t.sample).
// The step is then differentiated with the picks as constants:
//   1. teacher-forced forward that SAVES what the backward needs (pre-activation gates and cell states of both LSTMs, the
//      decoder's inputs / queries, window logits and probabilities)        lstm_train_forward / decode_train_forward
//   2. decoder backward, T steps in reverse: softmax / C*tanh / dot-attention gradient into d_enc_out (every (problem, row)
//      belongs to exactly one step's window: plain stores, no atomics) and into dh_k, LSTM-cell backward, dh_{k-1} and dx_k
//      by transposed matrix-vector products                                  decode_train_backward
//   3. encoder backward, L steps in reverse                                    lstm_train_backward
//   4. weight gradients as GEMMs over the saved gate gradients (gnnpn_gemm_f32, k-major operands), bias gradients as column
//      sums, gradient-norm clipping and Adam                                   colsum / sumsq / adam_step
// One workgroup owns whole problems for all steps (the streaming form of the inference recurrences): thread j = hidden unit j,
// the [4H,H] weight matrices are re-streamed from L2 every step with coalesced reads along j (row gu, column j), which is
// what the TRANSPOSED products dh_{t-1} = W_hh^T . dgates need.  fp32 throughout; accurate expf/tanhf.
// Parity: gradients within 2e-4 (relative, per parameter) of the reference's own autograd (tests/golden/pn_train_*.npz).
#include "common.h"
#include "recurrent.h"

#include "train_common.h"

// ---- 1a. encoder forward with saves: pregates [B,L,4H] -> enc_out [B,L,H], gates_pre [B,L,4H] (full pre-activations), c_all [B,L,H]
template <int H>
__global__ __launch_bounds__((H < 64 ? 64 : H)) void lstm_train_forward_kernel(
    const float* __restrict__ pregates, const float* __restrict__ whh, const float* __restrict__ bhh, float* __restrict__ enc_out,
    float* __restrict__ gates_pre, float* __restrict__ c_all, int32_t B, int32_t L) {
    __shared__ __attribute__((aligned(16))) float hs[H];
    const int b = blockIdx.x, j = threadIdx.x;
    const bool owner = j < H;
    float c = 0.0f;
    if (owner) hs[j] = 0.0f;
    __syncthreads();
    for (int t = 0; t < L; ++t) {
        float gate[4] = {0.f, 0.f, 0.f, 0.f};
        float h = 0.0f;
        if (owner) {
            matvec_rows<H>(whh, hs, j, gate);
            const int64_t base = ((int64_t)b * L + t) * (4 * H);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                gate[g] = (gate[g] + bhh[g * H + j]) + pregates[base + g * H + j];
                gates_pre[base + g * H + j] = gate[g];
            }
            c = sigm(gate[1]) * c + sigm(gate[0]) * tanhf(gate[2]);
            h = sigm(gate[3]) * tanhf(c);
            c_all[((int64_t)b * L + t) * H + j] = c;
            enc_out[((int64_t)b * L + t) * H + j] = h;
        }
        __syncthreads();
        if (owner) hs[j] = h;
        __syncthreads();
    }
}

// ---- 1b. teacher-forced decoder forward with saves.  picks idx [B,T] (global positions) are GIVEN.
struct DecTrain {
    const float* embedded;   // [B,L,H]
    const float* enc_out;    // [B,L,H]
    const float* h0;         // [B,H]  encoder final h (= enc_out[:, L-1])
    const float* c0;         // [B,H]
    const float* start;      // [H]
    const float* wih;        // TRANSPOSED [H,4H] (forward products); the backward takes the plain [4H,H]
    const float* whh;
    const float* bih;
    const float* bhh;
    const float* latent_win; // [B,T,K] or null
    const int32_t* idx;      // [B,T]
    float* x_all;            // [B,T,H] decoder inputs
    float* gates_pre;        // [B,T,4H]
    float* c_all;            // [B,T,H]
    float* h_all;            // [B,T,H]
    float* z0;               // [B,T,K]  C*tanh(dot) (or dot)
    float* probs;            // [B,T,K]
    float* logp;             // [B,T]    log-probability of the pick
    float tanh_c;
    int use_tanh;
    int32_t B, T, K;
};

template <int H>
__global__ __launch_bounds__((H < 64 ? 64 : H)) void decode_train_forward_kernel(DecTrain a) {
    __shared__ __attribute__((aligned(16))) float xs[H];
    __shared__ __attribute__((aligned(16))) float hs[H];
    __shared__ float lg[64];
    const int b = blockIdx.x, j = threadIdx.x, lane = j & 63, wave = j >> 6;
    constexpr int NW = (H < 64 ? 64 : H) / 64;
    const bool owner = j < H;
    const int T = a.T, K = a.K, L = T * K;
    float c = owner ? a.c0[(int64_t)b * H + j] : 0.0f;
    if (owner) {
        hs[j] = a.h0[(int64_t)b * H + j];
        xs[j] = a.start[j];
    }
    __syncthreads();
    for (int k = 0; k < T; ++k) {
        float h = 0.0f;
        if (owner) {
            float gi[4] = {0.f, 0.f, 0.f, 0.f}, gh[4] = {0.f, 0.f, 0.f, 0.f};
            matvec_rows<H>(a.wih, xs, j, gi);
            matvec_rows<H>(a.whh, hs, j, gh);
            const int64_t base = ((int64_t)b * T + k) * (4 * H);
            float gate[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                gate[g] = (gh[g] + a.bhh[g * H + j]) + (gi[g] + a.bih[g * H + j]);
                a.gates_pre[base + g * H + j] = gate[g];
            }
            a.x_all[((int64_t)b * T + k) * H + j] = xs[j];
            c = sigm(gate[1]) * c + sigm(gate[0]) * tanhf(gate[2]);
            h = sigm(gate[3]) * tanhf(c);
            a.c_all[((int64_t)b * T + k) * H + j] = c;
            a.h_all[((int64_t)b * T + k) * H + j] = h;
        }
        __syncthreads();
        if (owner) hs[j] = h;
        __syncthreads();
        for (int r = wave; r < K; r += NW) {          // window dots: one wave per candidate row
            const float* row = a.enc_out + ((int64_t)b * L + (int64_t)k * K + r) * H;
            float part = 0.0f;
            for (int e = lane; e < H; e += 64) part = fmaf(row[e], hs[e], part);
            const float dot = wave_sum(part);
            if (lane == 0) lg[r] = dot;
        }
        __syncthreads();
        if (j == 0) {
            const int64_t wb = ((int64_t)b * T + k) * K;
            float best = -INFINITY;
            for (int r = 0; r < K; ++r) {
                float v = a.use_tanh ? a.tanh_c * tanhf(lg[r]) : lg[r];
                a.z0[wb + r] = v;
                if (a.latent_win) v += a.latent_win[wb + r];
                lg[r] = v;
                best = fmaxf(best, v);
            }
            float denom = 0.0f;
            for (int r = 0; r < K; ++r) denom += expf(lg[r] - best);
            const int pick = a.idx[(int64_t)b * T + k] - k * K;
            for (int r = 0; r < K; ++r) a.probs[wb + r] = expf(lg[r] - best) / denom;
            a.logp[(int64_t)b * T + k] = (lg[pick] - best) - logf(denom);
        }
        if (owner) xs[j] = a.embedded[((int64_t)b * L + a.idx[(int64_t)b * T + k]) * H + j];   // modelPN.py:235
        __syncthreads();
    }
}

// ---- 2. decoder backward
struct DecBack {
    const float* enc_out;
    const float* whh;        // [4H,H]
    const float* wih;
    const float* c0;         // [B,H]
    const float* gates_pre;  // [B,T,4H]
    const float* c_all;
    const float* h_all;
    const float* z0;
    const float* probs;
    const int32_t* idx;
    const float* gscale;     // [B]  dLoss/dlogp of every step of problem b (= advantage_b / B, 0 where the reference zeroes)
    float* d_enc_out;        // [B,L,H]   (every element written exactly once)
    float* dgates;           // [B,T,4H]
    float* dx;               // [B,T,H]
    float* dh0;              // [B,H]
    float* dc0;
    float tanh_c;
    int use_tanh;
    int32_t B, T, K;
};

template <int H>
__global__ __launch_bounds__((H < 64 ? 64 : H)) void decode_train_backward_kernel(DecBack a) {
    __shared__ float dgs[4 * H];
    __shared__ float du[64];
    const int b = blockIdx.x, j = threadIdx.x;
    const bool owner = j < H;
    const int T = a.T, K = a.K, L = T * K;
    const float gs = a.gscale[b];
    float dh = 0.0f, dc = 0.0f;
    for (int k = T - 1; k >= 0; --k) {
        const int64_t wb = ((int64_t)b * T + k) * K;
        if (j < K) {   // softmax -> (+latent: constant) -> C*tanh backward
            const int pick = a.idx[(int64_t)b * T + k] - k * K;
            const float dz = gs * ((j == pick ? 1.0f : 0.0f) - a.probs[wb + j]);
            const float z = a.z0[wb + j];
            du[j] = a.use_tanh ? dz * (a.tanh_c - z * z / a.tanh_c) : dz;
        }
        __syncthreads();
        if (owner) {
            const float hk = a.h_all[((int64_t)b * T + k) * H + j];
            for (int r = 0; r < K; ++r) {
                const int64_t row = ((int64_t)b * L + (int64_t)k * K + r) * H + j;
                dh = fmaf(du[r], a.enc_out[row], dh);
                a.d_enc_out[row] = du[r] * hk;
            }
            const int64_t base = ((int64_t)b * T + k) * (4 * H);
            const float c_prev = k > 0 ? a.c_all[((int64_t)b * T + k - 1) * H + j] : a.c0[(int64_t)b * H + j];
            float dg[4];
            cell_backward(a.gates_pre[base + j], a.gates_pre[base + H + j], a.gates_pre[base + 2 * H + j],
                          a.gates_pre[base + 3 * H + j], c_prev, a.c_all[((int64_t)b * T + k) * H + j], dh, dc, dg);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                dgs[g * H + j] = dg[g];
                a.dgates[base + g * H + j] = dg[g];
            }
        }
        __syncthreads();
        if (owner) {
            a.dx[((int64_t)b * T + k) * H + j] = matvec_cols<H>(a.wih, dgs, j);
            dh = matvec_cols<H>(a.whh, dgs, j);
        }
        __syncthreads();
    }
    if (owner) {
        a.dh0[(int64_t)b * H + j] = dh;
        a.dc0[(int64_t)b * H + j] = dc;
    }
}

// ---- 3. encoder backward: dh_t = d_enc_out[b,t] (+ dh0 at t = L-1) + recurrent part
template <int H>
__global__ __launch_bounds__((H < 64 ? 64 : H)) void lstm_train_backward_kernel(
    const float* __restrict__ whh, const float* __restrict__ gates_pre, const float* __restrict__ c_all,
    const float* __restrict__ d_enc_out, const float* __restrict__ dh0, const float* __restrict__ dc0, float* __restrict__ dgates,
    int32_t B, int32_t L) {
    __shared__ float dgs[4 * H];
    const int b = blockIdx.x, j = threadIdx.x;
    const bool owner = j < H;
    float dh = owner ? dh0[(int64_t)b * H + j] : 0.0f, dc = owner ? dc0[(int64_t)b * H + j] : 0.0f;
    for (int t = L - 1; t >= 0; --t) {
        if (owner) {
            const int64_t base = ((int64_t)b * L + t) * (4 * H);
            dh += d_enc_out[((int64_t)b * L + t) * H + j];
            const float c_prev = t > 0 ? c_all[((int64_t)b * L + t - 1) * H + j] : 0.0f;
            float dg[4];
            cell_backward(gates_pre[base + j], gates_pre[base + H + j], gates_pre[base + 2 * H + j], gates_pre[base + 3 * H + j],
                          c_prev, c_all[((int64_t)b * L + t) * H + j], dh, dc, dg);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                dgs[g * H + j] = dg[g];
                dgates[base + g * H + j] = dg[g];
            }
        }
        __syncthreads();
        if (owner) dh = matvec_cols<H>(whh, dgs, j);
        __syncthreads();
    }
}

// ---- 4. small reductions / optimiser
// out[c] (+)= sum_r X[r, c] : one workgroup per 64 columns, 4 row-interleaved partial sums
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int64_t ld, int64_t rows, int cols,
                                                     float* __restrict__ out) {
    __shared__ float part[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float acc = 0.0f;
    if (c < cols)
        for (int64_t r = ry; r < rows; r += 4) acc += X[r * ld + c];
    part[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && c < cols) out[c] = (part[0][cx] + part[1][cx]) + (part[2][cx] + part[3][cx]);
}
__global__ __launch_bounds__(256) void colsum_chunks_kernel(const float* __restrict__ X, int64_t ld, int64_t rows, int cols,
                                                            int64_t per, float* __restrict__ partial) {
    __shared__ float part[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int64_t r0 = (int64_t)blockIdx.y * per, r1 = min(rows, r0 + per);
    float acc = 0.0f;
    if (c < cols)
        for (int64_t r = r0 + ry; r < r1; r += 4) acc += X[r * ld + c];
    part[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && c < cols) partial[(int64_t)blockIdx.y * cols + c] = (part[0][cx] + part[1][cx]) + (part[2][cx] + part[3][cx]);
}
// dst[idx[b*T + k], :] += src[b*T + k, :] for k >= 1 ... see the launcher: scatter of the decoder-input gradients into
// d_embedded (the rows a problem picks are distinct: one window per step) and of step 0 into the start-input sum buffer
__global__ void scatter_dx_kernel(const float* __restrict__ dx, const int32_t* __restrict__ idx, float* __restrict__ d_embedded,
                                  int32_t B, int32_t T, int32_t L, int32_t H) {
    const int64_t n = (int64_t)B * (T - 1) * H;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % H);
        const int64_t bk = i / H;
        const int b = (int)(bk / (T - 1)), k = (int)(bk % (T - 1)) + 1;          // x_k = embedded[b, idx[b,k-1]]
        d_embedded[((int64_t)b * L + idx[(int64_t)b * T + k - 1]) * H + j] += dx[((int64_t)b * T + k) * H + j];
    }
}
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, int64_t n, double* __restrict__ out) {
    __shared__ double part[256];
    double acc = 0.0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        acc += (double)x[i] * (double)x[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out, part[0]);
}
// Adam (torch.optim.Adam defaults, trainPNHigh.py:62) on g * clip, clip = min(1, max_norm / (sqrt(*sumsq) + 1e-6))
__global__ void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 int64_t n, const double* __restrict__ sumsq, float max_norm, float lr, float b1, float b2, float eps,
                                 float bc1, float bc2) {
    const float total = (float)sqrt(*sumsq);
    const float clip = fminf(max_norm / (total + 1e-6f), 1.0f);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * clip;
        const float mi = m[i] * b1 + (1.0f - b1) * gi;
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    }
}

// ---- C ABI -----------------------------------------------------------------------------------------------------------------
#define GNNPN_H_DISPATCH(H_, KERNEL, GRID, ...)                                                              \
    do {                                                                                                     \
        if ((H_) == 256) hipLaunchKernelGGL((KERNEL<256>), GRID, dim3(256), 0, s, __VA_ARGS__);              \
        else hipLaunchKernelGGL((KERNEL<32>), GRID, dim3(64), 0, s, __VA_ARGS__);                            \
    } while (0)

extern "C" int gnnpn_lstm_train_forward_f32(const float* pregates, const float* whh, const float* bhh, float* enc_out,
                                            float* gates_pre, float* c_all, int32_t B, int32_t L, int32_t H, void* stream) {
    GNNPN_REQUIRE(pregates && whh && bhh && enc_out && gates_pre && c_all, "lstm_train_forward: null operand");
    GNNPN_REQUIRE(B >= 0 && L > 0, "lstm_train_forward: bad shape");
    if (H != 256 && H != 32) GNNPN_FAIL(GNNPN_E_UNSUP, "lstm_train_forward: hidden size %d not built (256, 32)", H);
    GNNPN_REQUIRE(gnnpn_aligned(whh, 16), "lstm_train_forward: weights must be 16-byte aligned");
    if (B == 0) return GNNPN_OK;
    hipStream_t s = (hipStream_t)stream;
    GNNPN_H_DISPATCH(H, lstm_train_forward_kernel, dim3(B), pregates, whh, bhh, enc_out, gates_pre, c_all, B, L);
    GNNPN_CHECK_LAUNCH("lstm_train_forward_f32");
    return GNNPN_OK;
}

extern "C" int gnnpn_decode_train_forward_f32(const gnnpn_decode_train_t* t, int32_t B, int32_t T, int32_t n_per, int32_t H,
                                              float tanh_c, int use_tanh, void* stream) {
    GNNPN_REQUIRE(t && t->embedded && t->enc_out && t->h0 && t->c0 && t->start && t->wih && t->whh && t->bih && t->bhh && t->idx &&
                      t->x_all && t->gates_pre && t->c_all && t->h_all && t->z0 && t->probs && t->logp,
                  "decode_train_forward: null operand");
    GNNPN_REQUIRE(B >= 0 && T > 0 && n_per >= 1 && n_per <= 64, "decode_train_forward: bad shape");
    if (H != 256 && H != 32) GNNPN_FAIL(GNNPN_E_UNSUP, "decode_train_forward: hidden size %d not built (256, 32)", H);
    GNNPN_REQUIRE(gnnpn_aligned(t->wih, 16) && gnnpn_aligned(t->whh, 16), "decode_train_forward: weights must be 16-byte aligned");
    if (B == 0) return GNNPN_OK;
    DecTrain a{t->embedded, t->enc_out, t->h0, t->c0, t->start, t->wih, t->whh, t->bih, t->bhh, t->latent_win, t->idx, t->x_all,
               t->gates_pre, t->c_all, t->h_all, t->z0, t->probs, t->logp, tanh_c, use_tanh, B, T, n_per};
    hipStream_t s = (hipStream_t)stream;
    GNNPN_H_DISPATCH(H, decode_train_forward_kernel, dim3(B), a);
    GNNPN_CHECK_LAUNCH("decode_train_forward_f32");
    return GNNPN_OK;
}

extern "C" int gnnpn_decode_train_backward_f32(const gnnpn_decode_train_t* t, const float* gscale, float* d_enc_out, float* dgates,
                                               float* dx, float* dh0, float* dc0, int32_t B, int32_t T, int32_t n_per, int32_t H,
                                               float tanh_c, int use_tanh, void* stream) {
    GNNPN_REQUIRE(t && t->enc_out && t->c0 && t->wih && t->whh && t->idx && t->gates_pre && t->c_all && t->h_all && t->z0 &&
                      t->probs && gscale && d_enc_out && dgates && dx && dh0 && dc0, "decode_train_backward: null operand");
    GNNPN_REQUIRE(B >= 0 && T > 0 && n_per >= 1 && n_per <= 64, "decode_train_backward: bad shape");
    if (H != 256 && H != 32) GNNPN_FAIL(GNNPN_E_UNSUP, "decode_train_backward: hidden size %d not built (256, 32)", H);
    if (B == 0) return GNNPN_OK;
    DecBack a{t->enc_out, t->whh, t->wih, t->c0, t->gates_pre, t->c_all, t->h_all, t->z0, t->probs, t->idx, gscale, d_enc_out,
              dgates, dx, dh0, dc0, tanh_c, use_tanh, B, T, n_per};
    hipStream_t s = (hipStream_t)stream;
    GNNPN_H_DISPATCH(H, decode_train_backward_kernel, dim3(B), a);
    GNNPN_CHECK_LAUNCH("decode_train_backward_f32");
    return GNNPN_OK;
}

extern "C" int gnnpn_lstm_train_backward_f32(const float* whh, const float* gates_pre, const float* c_all, const float* d_enc_out,
                                             const float* dh0, const float* dc0, float* dgates, int32_t B, int32_t L, int32_t H,
                                             void* stream) {
    GNNPN_REQUIRE(whh && gates_pre && c_all && d_enc_out && dh0 && dc0 && dgates, "lstm_train_backward: null operand");
    GNNPN_REQUIRE(B >= 0 && L > 0, "lstm_train_backward: bad shape");
    if (H != 256 && H != 32) GNNPN_FAIL(GNNPN_E_UNSUP, "lstm_train_backward: hidden size %d not built (256, 32)", H);
    if (B == 0) return GNNPN_OK;
    hipStream_t s = (hipStream_t)stream;
    GNNPN_H_DISPATCH(H, lstm_train_backward_kernel, dim3(B), whh, gates_pre, c_all, d_enc_out, dh0, dc0, dgates, B, L);
    GNNPN_CHECK_LAUNCH("lstm_train_backward_f32");
    return GNNPN_OK;
}

extern "C" int gnnpn_colsum_f32(const float* X, int64_t ld, int64_t rows, int32_t cols, float* out, void* stream) {
    GNNPN_REQUIRE(X && out && rows >= 0 && cols > 0 && ld >= cols, "colsum: bad argument");
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64), dim3(256), 0, (hipStream_t)stream, X, ld, rows, cols, out);
    GNNPN_CHECK_LAUNCH("colsum_f32");
    return GNNPN_OK;
}

extern "C" int gnnpn_colsum_chunks_f32(const float* X, int64_t ld, int64_t rows, int32_t cols, int64_t rows_per_chunk,
                                       float* partial, void* stream) {
    GNNPN_REQUIRE(X && partial && rows >= 0 && cols > 0 && ld >= cols && rows_per_chunk > 0, "colsum_chunks: bad argument");
    const int64_t chunks = (rows + rows_per_chunk - 1) / rows_per_chunk;
    if (chunks == 0) return GNNPN_OK;
    hipLaunchKernelGGL(colsum_chunks_kernel, dim3((cols + 63) / 64, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, X, ld,
                       rows, cols, rows_per_chunk, partial);
    GNNPN_CHECK_LAUNCH("colsum_chunks_f32");
    return GNNPN_OK;
}

extern "C" int gnnpn_scatter_dx_f32(const float* dx, const int32_t* idx, float* d_embedded, int32_t B, int32_t T, int32_t L,
                                    int32_t H, void* stream) {
    GNNPN_REQUIRE(dx && idx && d_embedded && B >= 0 && T >= 1 && L >= T && H > 0, "scatter_dx: bad argument");
    if (B == 0 || T == 1) return GNNPN_OK;
    const int64_t n = (int64_t)B * (T - 1) * H;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scatter_dx_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dx, idx, d_embedded, B, T, L, H);
    GNNPN_CHECK_LAUNCH("scatter_dx_f32");
    return GNNPN_OK;
}

extern "C" int gnnpn_sumsq_f32(const float* x, int64_t n, double* accum, void* stream) {
    GNNPN_REQUIRE(x && accum && n >= 0, "sumsq: bad argument");
    if (n == 0) return GNNPN_OK;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n, accum);
    GNNPN_CHECK_LAUNCH("sumsq_f32");
    return GNNPN_OK;
}

extern "C" int gnnpn_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, const double* grad_sumsq,
                                   float max_grad_norm, float lr, float beta1, float beta2, float eps, int32_t step, void* stream) {
    GNNPN_REQUIRE(p && g && m && v && grad_sumsq && n >= 0 && step >= 1, "adam_step: bad argument");
    if (n == 0) return GNNPN_OK;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(adam_step_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, grad_sumsq, max_grad_norm,
                       lr, beta1, beta2, eps, bc1, bc2);
    GNNPN_CHECK_LAUNCH("adam_step_f32");
    return GNNPN_OK;
}
