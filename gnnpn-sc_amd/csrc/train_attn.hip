// REINFORCE training step THROUGH the attention forms the reference's configurations switch off: 'Bahdanau' attention
// (src/models/modelPN.py:80-90,103-109) and glimpse rounds (:208-211) — the decoder half of the backward pass of train.hip for
// them.  The picks are constants of the differentiated graph, as there.
//
// Per decode step k (modelPN.py:204-239), with q_0 = h_k and `chosen` = the picks of the steps before k:
//   glimpse round g < G (the ONE glimpse module, every round):   u_i = att_g(q_g, i) for all L positions, -inf at the chosen ones,
//        a = softmax(u),   q_{g+1} = sum_i a_i r_i          r_i = W_ref enc_i + b_ref ('Bahdanau': `ref`, given) | enc_i ('Dot')
//   pointer:  z_i = C tanh(att_p(q_G, i)) on the step's window (+ the Low net's logits), softmax, log p(pick)
//   att(q, i) = V . tanh(W_q q + b_q + r_i) ('Bahdanau')  |  enc_i . q ('Dot')
// Forward saves q_0..q_G and the glimpse softmaxes; the backward recomputes the tanh terms (B*T*L*H of them: 1.4 GB at the QWS
// shape if saved) and leaves, besides what train.hip's backward leaves, d ref (pointer and glimpse: [B,L,H], += over the steps),
// the gradients wrt the projected queries per step ([B,T,H], [B,T,G,H]: the weight gradients of W_query are GEMMs over them) and
// the per-problem sums for V.  W_ref's gradient and d enc_out's share through `ref` are GEMMs over d ref (trainPNHigh.py).
// One workgroup owns a problem for all steps, thread j = hidden unit j, as train.hip; fp32, accurate expf / tanhf.
// Parity: tests/golden/pn_train_{bahdanau,dot}_*.npz (the reference's own autograd).
#include "common.h"
#include "recurrent.h"
#include "train_common.h"

namespace {

struct AttnTrain {
    gnnpn_decode_attn_train_t t;
    const float* gscale;     // backward: [B]
    float* d_enc_out;        // backward: [B,L,H], += (zeroed by the caller)
    float* dgates;           // [B,T,4H]
    float* dx;               // [B,T,H]
    float* dh0;              // [B,H]
    float* dc0;
    float tanh_c;
    int use_tanh;
    int32_t B, T, K;
};

// deterministic workgroup reductions through LDS (wave sums in lane order, then the waves in order)
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    constexpr int NW = NT / 64;
    v = wave_sum(v);
    if constexpr (NW == 1) return v;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += red[w];
    return s;
}
__device__ __forceinline__ float wave_maxf(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
template <int NT>
__device__ __forceinline__ float block_max(float v, float* red) {
    constexpr int NW = NT / 64;
    v = wave_maxf(v);
    if constexpr (NW == 1) return v;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) s = fmaxf(s, red[w]);
    return s;
}

// qp[j] = b[j] + sum_k W[j][k] q[k] from the TRANSPOSED matrix Wt[k][j] (coalesced along j), k ascending
template <int H>
__device__ __forceinline__ float project(const float* __restrict__ Wt, const float* __restrict__ bias, const float* q, int j) {
    float acc = 0.0f;
#pragma unroll 4
    for (int k = 0; k < H; ++k) acc = fmaf(Wt[(size_t)k * H + j], q[k], acc);
    return acc + bias[j];
}
// sum_m W[m][j] d[m]: the transposed product from the row-major matrix (coalesced along j)
template <int H>
__device__ __forceinline__ float project_back(const float* __restrict__ W, const float* d, int j) {
    float a0 = 0.0f, a1 = 0.0f;
    for (int m = 0; m < H; m += 2) {
        a0 = fmaf(W[(size_t)m * H + j], d[m], a0);
        a1 = fmaf(W[(size_t)(m + 1) * H + j], d[m + 1], a1);
    }
    return a0 + a1;
}

// logits of the positions [i0, i0 + n) against the query in LDS: one wavefront per position.  bah: V . tanh(qp + ref_i); else enc_i . q
template <int H, int NT>
__device__ __forceinline__ void attention_logits(bool bah, const float* __restrict__ rows, const float* qv, const float* __restrict__ v,
                                                 int i0, int n, float* out) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < n; i += NW) {
        const float* row = rows + (size_t)(i0 + i) * H;
        float part = 0.0f;
        for (int e = lane; e < H; e += 64) part = bah ? fmaf(v[e], tanhf(qv[e] + row[e]), part) : fmaf(row[e], qv[e], part);
        const float dot = wave_sum(part);
        if (lane == 0) out[i] = dot;
    }
}

// ---- forward with saves --------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__((H < 64 ? 64 : H)) void decode_attn_train_forward_kernel(AttnTrain a) {
    constexpr int NT = H < 64 ? 64 : H;
    extern __shared__ __attribute__((aligned(16))) float dyn[];           // [L] logits / softmax, then [L] bytes: chosen
    __shared__ __attribute__((aligned(16))) float xs[H], hs[H], qs[H], qp[H];
    __shared__ float lg[64], red[NT / 64 + 1];
    const gnnpn_decode_attn_train_t& t = a.t;
    const int b = blockIdx.x, j = threadIdx.x;
    const bool owner = j < H, bah = t.bahdanau != 0;
    const int T = a.T, K = a.K, L = T * K, G = t.n_glimpses;
    float* ul = dyn;
    unsigned char* chosen = reinterpret_cast<unsigned char*>(dyn + L);
    for (int i = j; i < L; i += NT) chosen[i] = 0;
    float c = owner ? t.base.c0[(int64_t)b * H + j] : 0.0f;
    if (owner) {
        hs[j] = t.base.h0[(int64_t)b * H + j];
        xs[j] = t.base.start[j];
    }
    __syncthreads();
    const float* enc_b = t.base.enc_out + (int64_t)b * L * H;
    const float* gref_b = bah && G ? t.g_ref + (int64_t)b * L * H : enc_b;
    const float* pref_b = bah ? t.p_ref + (int64_t)b * L * H : enc_b;
    for (int k = 0; k < T; ++k) {
        float h = 0.0f;
        if (owner) {                                                      // the decoder's LSTM cell (train.hip)
            float gi[4] = {0.f, 0.f, 0.f, 0.f}, gh[4] = {0.f, 0.f, 0.f, 0.f};
            matvec_rows<H>(t.base.wih, xs, j, gi);
            matvec_rows<H>(t.base.whh, hs, j, gh);
            const int64_t base = ((int64_t)b * T + k) * (4 * H);
            float gate[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                gate[g] = (gh[g] + t.base.bhh[g * H + j]) + (gi[g] + t.base.bih[g * H + j]);
                t.base.gates_pre[base + g * H + j] = gate[g];
            }
            t.base.x_all[((int64_t)b * T + k) * H + j] = xs[j];
            c = sigm(gate[1]) * c + sigm(gate[0]) * tanhf(gate[2]);
            h = sigm(gate[3]) * tanhf(c);
            t.base.c_all[((int64_t)b * T + k) * H + j] = c;
            t.base.h_all[((int64_t)b * T + k) * H + j] = h;
        }
        __syncthreads();
        if (owner) {
            hs[j] = h;
            qs[j] = h;
            t.q_all[(((int64_t)b * T + k) * (G + 1)) * H + j] = h;       // q_0
        }
        if (j == 0 && k > 0) chosen[t.base.idx[(int64_t)b * T + k - 1]] = 1;   // modelPN.py:169-172: the previous step's pick
        __syncthreads();
        for (int g = 0; g < G; ++g) {                                     // glimpse rounds (:208-211)
            if (bah) {
                const float v = owner ? project<H>(t.g_wq_t, t.g_bq, qs, j) : 0.0f;
                if (owner) qp[j] = v;
                __syncthreads();
            }
            attention_logits<H, NT>(bah, gref_b, bah ? qp : qs, t.g_v, 0, L, ul);
            __syncthreads();
            float m = -INFINITY;
            for (int i = j; i < L; i += NT) {
                if (chosen[i]) ul[i] = -INFINITY;
                m = fmaxf(m, ul[i]);
            }
            m = block_max<NT>(m, red);
            float part = 0.0f;
            for (int i = j; i < L; i += NT) part += expf(ul[i] - m);
            const float denom = block_sum<NT>(part, red);
            __syncthreads();
            float* a_out = t.a_all + (((int64_t)b * T + k) * G + g) * L;
            for (int i = j; i < L; i += NT) {
                const float p = expf(ul[i] - m) / denom;
                ul[i] = p;
                a_out[i] = p;
            }
            __syncthreads();
            float qn = 0.0f;
            if (owner)
                for (int i = 0; i < L; ++i) qn = fmaf(ul[i], gref_b[(size_t)i * H + j], qn);       // :209
            __syncthreads();
            if (owner) {
                qs[j] = qn;
                t.q_all[(((int64_t)b * T + k) * (G + 1) + g + 1) * H + j] = qn;
            }
            __syncthreads();
        }
        if (bah) {                                                        // the pointer (:213)
            const float v = owner ? project<H>(t.p_wq_t, t.p_bq, qs, j) : 0.0f;
            if (owner) qp[j] = v;
            __syncthreads();
        }
        attention_logits<H, NT>(bah, pref_b, bah ? qp : qs, t.p_v, k * K, K, lg);
        __syncthreads();
        if (j == 0) {
            const int64_t wb = ((int64_t)b * T + k) * K;
            float best = -INFINITY;
            for (int r = 0; r < K; ++r) {
                float v = a.use_tanh ? a.tanh_c * tanhf(lg[r]) : lg[r];
                t.base.z0[wb + r] = v;
                if (t.base.latent_win) v += t.base.latent_win[wb + r];
                lg[r] = v;
                best = fmaxf(best, v);
            }
            float denom = 0.0f;
            for (int r = 0; r < K; ++r) denom += expf(lg[r] - best);
            const int pick = t.base.idx[(int64_t)b * T + k] - k * K;
            for (int r = 0; r < K; ++r) t.base.probs[wb + r] = expf(lg[r] - best) / denom;
            t.base.logp[(int64_t)b * T + k] = (lg[pick] - best) - logf(denom);
        }
        if (owner) xs[j] = t.base.embedded[((int64_t)b * L + t.base.idx[(int64_t)b * T + k]) * H + j];   // :235
        __syncthreads();
    }
}

// ---- backward: T steps in reverse -------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__((H < 64 ? 64 : H)) void decode_attn_train_backward_kernel(AttnTrain a) {
    constexpr int NT = H < 64 ? 64 : H;
    extern __shared__ __attribute__((aligned(16))) float dyn[];           // [L] glimpse softmax a, [L] da / du
    __shared__ float dgs[4 * H];
    __shared__ __attribute__((aligned(16))) float qs[H], qp[H], dq[H], dqp[H];
    __shared__ float du[64], red[NT / 64 + 1];
    const gnnpn_decode_attn_train_t& t = a.t;
    const int b = blockIdx.x, j = threadIdx.x;
    const bool owner = j < H, bah = t.bahdanau != 0;
    const int T = a.T, K = a.K, L = T * K, G = t.n_glimpses;
    float* al = dyn;
    float* dal = dyn + L;
    const float gs = a.gscale[b];
    const float* enc_b = t.base.enc_out + (int64_t)b * L * H;
    const float* gref_b = bah && G ? t.g_ref + (int64_t)b * L * H : enc_b;
    const float* pref_b = bah ? t.p_ref + (int64_t)b * L * H : enc_b;
    float* denc_b = a.d_enc_out + (int64_t)b * L * H;
    float* dgref_b = bah && G ? t.d_g_ref + (int64_t)b * L * H : denc_b;  // 'Dot': r_i = enc_i
    float* dpref_b = bah ? t.d_p_ref + (int64_t)b * L * H : denc_b;
    float dh = 0.0f, dc = 0.0f, dv_p = 0.0f, dv_g = 0.0f;
    const float vp = bah && owner ? t.p_v[j] : 0.0f, vg = bah && G && owner ? t.g_v[j] : 0.0f;
    for (int k = T - 1; k >= 0; --k) {
        const int64_t wb = ((int64_t)b * T + k) * K;
        if (j < K) {                                                      // softmax -> (+ latent: constant) -> C*tanh backward
            const int pick = t.base.idx[(int64_t)b * T + k] - k * K;
            const float dz = gs * ((j == pick ? 1.0f : 0.0f) - t.base.probs[wb + j]);
            const float z = t.base.z0[wb + j];
            du[j] = a.use_tanh ? dz * (a.tanh_c - z * z / a.tanh_c) : dz;
        }
        if (owner) qs[j] = t.q_all[(((int64_t)b * T + k) * (G + 1) + G) * H + j];   // q_G: the pointer's query
        __syncthreads();
        // ---- the pointer: d q_G, d ref (window rows) / d enc_out, d V, d qp
        float dqj = 0.0f;
        if (bah) {
            const float v = owner ? project<H>(t.p_wq_t, t.p_bq, qs, j) : 0.0f;
            float dp = 0.0f;
            if (owner) {
                for (int r = 0; r < K; ++r) {
                    const size_t at = (size_t)(k * K + r) * H + j;
                    const float th = tanhf(v + pref_b[at]);
                    dv_p = fmaf(du[r], th, dv_p);
                    const float ds = du[r] * vp * (1.0f - th * th);
                    dp += ds;
                    dpref_b[at] += ds;
                }
                dqp[j] = dp;
                t.d_p_qp[((int64_t)b * T + k) * H + j] = dp;
            }
            __syncthreads();
            if (owner) dqj = project_back<H>(t.p_wq, dqp, j);
        } else if (owner) {
            const float qj = qs[j];
            for (int r = 0; r < K; ++r) {
                const size_t at = (size_t)(k * K + r) * H + j;
                dqj = fmaf(du[r], enc_b[at], dqj);
                denc_b[at] += du[r] * qj;
            }
        }
        __syncthreads();
        // ---- the glimpse rounds in reverse
        for (int g = G - 1; g >= 0; --g) {
            if (owner) {
                dq[j] = dqj;                                              // gradient wrt q_{g+1}
                qs[j] = t.q_all[(((int64_t)b * T + k) * (G + 1) + g) * H + j];           // q_g: the round's query
            }
            const float* a_in = t.a_all + (((int64_t)b * T + k) * G + g) * L;
            for (int i = j; i < L; i += NT) al[i] = a_in[i];
            __syncthreads();
            attention_logits<H, NT>(false, gref_b, dq, nullptr, 0, L, dal);              // da_i = dq . r_i
            if (owner)
                for (int i = 0; i < L; ++i)
                    if (al[i] != 0.0f) dgref_b[(size_t)i * H + j] += al[i] * dqj;        // d r_i += a_i dq   (q_{g+1} = sum a_i r_i)
            __syncthreads();
            float part = 0.0f;
            for (int i = j; i < L; i += NT) part = fmaf(al[i], dal[i], part);
            const float dot = block_sum<NT>(part, red);
            __syncthreads();
            for (int i = j; i < L; i += NT) dal[i] = al[i] * (dal[i] - dot);             // du_i (0 at the chosen positions: a_i = 0)
            __syncthreads();
            float dnew = 0.0f;
            if (bah) {
                const float v = owner ? project<H>(t.g_wq_t, t.g_bq, qs, j) : 0.0f;
                float dp = 0.0f;
                if (owner) {
                    for (int i = 0; i < L; ++i) {
                        const float dui = dal[i];
                        if (dui == 0.0f) continue;
                        const size_t at = (size_t)i * H + j;
                        const float th = tanhf(v + gref_b[at]);
                        dv_g = fmaf(dui, th, dv_g);
                        const float ds = dui * vg * (1.0f - th * th);
                        dp += ds;
                        dgref_b[at] += ds;
                    }
                    dqp[j] = dp;
                    t.d_g_qp[(((int64_t)b * T + k) * G + g) * H + j] = dp;
                }
                __syncthreads();
                if (owner) dnew = project_back<H>(t.g_wq, dqp, j);
            } else if (owner) {
                const float qj = qs[j];
                for (int i = 0; i < L; ++i) {
                    const float dui = dal[i];
                    if (dui == 0.0f) continue;
                    const size_t at = (size_t)i * H + j;
                    dnew = fmaf(dui, enc_b[at], dnew);
                    denc_b[at] += dui * qj;
                }
            }
            __syncthreads();
            dqj = dnew;
        }
        // ---- the LSTM cell (train.hip): dh_k = recurrent part + the attention's d q_0
        if (owner) {
            dh += dqj;
            const int64_t base = ((int64_t)b * T + k) * (4 * H);
            const float c_prev = k > 0 ? t.base.c_all[((int64_t)b * T + k - 1) * H + j] : t.base.c0[(int64_t)b * H + j];
            float dg[4];
            cell_backward(t.base.gates_pre[base + j], t.base.gates_pre[base + H + j], t.base.gates_pre[base + 2 * H + j],
                          t.base.gates_pre[base + 3 * H + j], c_prev, t.base.c_all[((int64_t)b * T + k) * H + j], dh, dc, dg);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                dgs[g * H + j] = dg[g];
                a.dgates[base + g * H + j] = dg[g];
            }
        }
        __syncthreads();
        if (owner) {
            a.dx[((int64_t)b * T + k) * H + j] = matvec_cols<H>(t.base.wih, dgs, j);
            dh = matvec_cols<H>(t.base.whh, dgs, j);
        }
        __syncthreads();
    }
    if (owner) {
        a.dh0[(int64_t)b * H + j] = dh;
        a.dc0[(int64_t)b * H + j] = dc;
        if (bah) {
            t.d_p_v[(int64_t)b * H + j] = dv_p;
            if (G) t.d_g_v[(int64_t)b * H + j] = dv_g;
        }
    }
}

int check(const gnnpn_decode_attn_train_t* t, int32_t B, int32_t T, int32_t n_per, int32_t H, const char* who) {
    GNNPN_REQUIRE(t && t->base.embedded && t->base.enc_out && t->base.h0 && t->base.c0 && t->base.start && t->base.wih && t->base.whh &&
                      t->base.bih && t->base.bhh && t->base.idx && t->base.x_all && t->base.gates_pre && t->base.c_all && t->base.h_all &&
                      t->base.z0 && t->base.probs && t->base.logp && t->q_all,
                  "decode_attn_train: null operand");
    GNNPN_REQUIRE(B >= 0 && T > 0 && n_per >= 1 && n_per <= 64 && t->n_glimpses >= 0 && t->n_glimpses <= 8, "decode_attn_train: bad shape");
    GNNPN_REQUIRE(t->n_glimpses == 0 || t->a_all, "decode_attn_train: glimpse rounds need a_all");
    if (t->bahdanau) {
        GNNPN_REQUIRE(t->p_wq_t && t->p_wq && t->p_bq && t->p_v && t->p_ref, "decode_attn_train: 'Bahdanau' pointer parameters missing");
        GNNPN_REQUIRE(t->n_glimpses == 0 || (t->g_wq_t && t->g_wq && t->g_bq && t->g_v && t->g_ref),
                      "decode_attn_train: 'Bahdanau' glimpse parameters missing");
    }
    if (H != 256 && H != 32) GNNPN_FAIL(GNNPN_E_UNSUP, "%s: hidden size %d not built (256, 32)", who, H);
    if ((int64_t)T * n_per * 12 > 150 * 1024) GNNPN_FAIL(GNNPN_E_UNSUP, "%s: %d positions do not fit the LDS", who, T * n_per);
    return GNNPN_OK;
}
}  // namespace

#define GNNPN_ATTN_DISPATCH(H_, KERNEL, GRID, LDS_, ARG)                                                                      \
    do {                                                                                                                     \
        if ((H_) == 256) {                                                                                                   \
            (void)hipFuncSetAttribute((const void*)KERNEL<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_));    \
            hipLaunchKernelGGL((KERNEL<256>), GRID, dim3(256), LDS_, s, ARG);                                                \
        } else {                                                                                                             \
            (void)hipFuncSetAttribute((const void*)KERNEL<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS_));     \
            hipLaunchKernelGGL((KERNEL<32>), GRID, dim3(64), LDS_, s, ARG);                                                  \
        }                                                                                                                    \
    } while (0)

extern "C" int gnnpn_decode_attn_train_forward_f32(const gnnpn_decode_attn_train_t* t, int32_t B, int32_t T, int32_t n_per, int32_t H,
                                                   float tanh_c, int use_tanh, void* stream) {
    if (int rc = check(t, B, T, n_per, H, "decode_attn_train_forward")) return rc;
    if (B == 0) return GNNPN_OK;
    AttnTrain a{*t, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, tanh_c, use_tanh, B, T, n_per};
    hipStream_t s = (hipStream_t)stream;
    const unsigned lds = (unsigned)((size_t)T * n_per * 4 + ((size_t)T * n_per + 15) / 16 * 16);
    GNNPN_ATTN_DISPATCH(H, decode_attn_train_forward_kernel, dim3(B), lds, a);
    GNNPN_CHECK_LAUNCH("decode_attn_train_forward_f32");
    return GNNPN_OK;
}

extern "C" int gnnpn_decode_attn_train_backward_f32(const gnnpn_decode_attn_train_t* t, const float* gscale, float* d_enc_out,
                                                    float* dgates, float* dx, float* dh0, float* dc0, int32_t B, int32_t T,
                                                    int32_t n_per, int32_t H, float tanh_c, int use_tanh, void* stream) {
    if (int rc = check(t, B, T, n_per, H, "decode_attn_train_backward")) return rc;
    GNNPN_REQUIRE(gscale && d_enc_out && dgates && dx && dh0 && dc0, "decode_attn_train_backward: null operand");
    if (t->bahdanau) {
        GNNPN_REQUIRE(t->d_p_ref && t->d_p_qp && t->d_p_v, "decode_attn_train_backward: 'Bahdanau' pointer gradient buffers missing");
        GNNPN_REQUIRE(t->n_glimpses == 0 || (t->d_g_ref && t->d_g_qp && t->d_g_v),
                      "decode_attn_train_backward: 'Bahdanau' glimpse gradient buffers missing");
    }
    if (B == 0) return GNNPN_OK;
    AttnTrain a{*t, gscale, d_enc_out, dgates, dx, dh0, dc0, tanh_c, use_tanh, B, T, n_per};
    hipStream_t s = (hipStream_t)stream;
    const unsigned lds = (unsigned)((size_t)T * n_per * 8);
    GNNPN_ATTN_DISPATCH(H, decode_attn_train_backward_kernel, dim3(B), lds, a);
    GNNPN_CHECK_LAUNCH("decode_attn_train_backward_f32");
    return GNNPN_OK;
}
