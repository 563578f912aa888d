// Pointer decode with the attention forms the reference's configurations leave switched off:
// 'Bahdanau' attention (modelPN.py:80-90,103-109) and glimpse rounds (:208-211), for either attention name.
//
// General form, one workgroup per problem for all T steps (as decode.hip, no inter-workgroup communication); the
// shipped configuration ('Dot', no glimpses) never comes here.  Per step k (modelPN.py:204-239):
//   decoder LSTM cell                       two k-ordered fmaf-chain GEMVs streamed from L2 (recurrent.h)
//   query = h; n_glimpses times:            logits over ALL L positions (a glimpse is not windowed), -inf at the positions
//                                           chosen so far (:165-173), softmax over L, query = ref' . softmax  (:209-211)
//   pointer logits of the step's window     the only positions the window mask keeps (:220-222); C*tanh (:119-120)
//   + latent (High net), softmax over the window, first-max argmax, gather of the next input and of the action row.
// Attention (modelPN.py:92-122):  Dot:      logits_l = ref_l . q,                      ref' = ref
//                                 Bahdanau: logits_l = V . tanh(W_query q + b + R_l),  ref' = R = W_ref(ref) + b_ref
// R (the 1x1 Conv1d of enc_out) does not depend on the step and is precomputed by the caller with gnnpn_linear_f32.
#include "common.h"
#include <string.h>
#include "recurrent.h"
#include "decode_shared.h"

namespace {

struct AttnSide {          // one Attention module
    const float* wq;       // [H,H] W_query.weight (row-major, as stored)      Bahdanau only
    const float* bq;       // [H]
    const float* ref;      // [B,L,H] W_ref(enc_out) + bias                     Bahdanau only
    const float* v;        // [H]
};

template <int NT>
__device__ __forceinline__ float block_max(float v, float* red) {
    constexpr int NW = NT / 64;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    if (NW == 1) return v;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float m = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
    return m;
}
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {   // fixed order: butterfly inside a wave, waves ascending
    constexpr int NW = NT / 64;
    v = wave_sum(v);
    if (NW == 1) return v;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) s = __fadd_rn(s, red[w]);
    return s;
}

// logits of positions [l0, l0+n) for query q (LDS), one wavefront per position; out[i] for position l0+i
template <int H, int NT, bool BAHDANAU>
__device__ __forceinline__ void attention_logits(const float* __restrict__ enc_b, const AttnSide& a, const float* ref_b,
                                                 const float* q, const float* qw, int l0, int n, float* out) {
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = wave; i < n; i += NW) {
        const float* row = (BAHDANAU ? ref_b : enc_b) + (int64_t)(l0 + i) * H;
        float part = 0.0f;
        for (int e = lane * 4; e < H; e += 256) {
            const float4 rv = *reinterpret_cast<const float4*>(row + e);
            if constexpr (BAHDANAU) {
                const float4 qv = *reinterpret_cast<const float4*>(qw + e);
                const float4 vv = *reinterpret_cast<const float4*>(a.v + e);
                part = fmaf(vv.x, tanhf(__fadd_rn(qv.x, rv.x)), part);
                part = fmaf(vv.y, tanhf(__fadd_rn(qv.y, rv.y)), part);
                part = fmaf(vv.z, tanhf(__fadd_rn(qv.z, rv.z)), part);
                part = fmaf(vv.w, tanhf(__fadd_rn(qv.w, rv.w)), part);
            } else {
                const float4 qv = *reinterpret_cast<const float4*>(q + e);
                part = fmaf(rv.x, qv.x, part);
                part = fmaf(rv.y, qv.y, part);
                part = fmaf(rv.z, qv.z, part);
                part = fmaf(rv.w, qv.w, part);
            }
        }
        const float dot = wave_sum(part);
        if (lane == 0) out[i] = dot;
    }
}

template <int H, bool BAHDANAU>
__global__ __launch_bounds__((H < 64 ? 64 : H)) void pointer_decode_glimpse_kernel(DecodeNet net, AttnSide ptr, AttnSide gl,
                                                                                  const float* __restrict__ inputs,
                                                                                  int n_glimpses, float tanh_c, int use_tanh,
                                                                                  int32_t B, int32_t T, int32_t n_per) {
    constexpr int NT = H < 64 ? 64 : H;
    extern __shared__ __attribute__((aligned(16))) float dyn[];   // lg[L] (glimpse logits / probabilities), chosen[T]
    __shared__ __attribute__((aligned(16))) float xs[1][H];
    __shared__ __attribute__((aligned(16))) float hs[2][1][H];
    __shared__ __attribute__((aligned(16))) float qv[H];
    __shared__ __attribute__((aligned(16))) float qw[H];
    __shared__ float win[64];
    __shared__ float red[NT / 64];
    __shared__ int sel;

    const int j = threadIdx.x, b = blockIdx.x;
    const bool owner = j < H;
    const int L = T * n_per;
    float* lg = dyn;
    int* chosen = reinterpret_cast<int*>(dyn + L);
    const float4* __restrict__ Wi = reinterpret_cast<const float4*>(net.wih);
    const float4* __restrict__ Wh = reinterpret_cast<const float4*>(net.whh);
    const float* enc_b = net.enc_out + (int64_t)b * L * H;
    const float* pref_b = BAHDANAU ? ptr.ref + (int64_t)b * L * H : nullptr;
    const float* gref_b = BAHDANAU ? gl.ref + (int64_t)b * L * H : nullptr;

    float bi[4], bh[4], c[1], h[1];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bi[g] = owner ? net.bih[g * H + j] : 0.0f;
        bh[g] = owner ? net.bhh[g * H + j] : 0.0f;
    }
    c[0] = owner ? net.c0[(int64_t)b * H + j] : 0.0f;
    h[0] = owner ? net.h0[(int64_t)b * H + j] : 0.0f;
    if (owner) {
        hs[0][0][j] = h[0];
        xs[0][j] = net.start[j];
    }
    __syncthreads();

    int cur = 0;
    for (int k = 0; k < T; ++k) {
        // ---- decoder LSTM cell (modelPN.py:205)
        if (owner) {
            float ai[1][4] = {{0.f, 0.f, 0.f, 0.f}}, ah[1][4] = {{0.f, 0.f, 0.f, 0.f}};
            gemv_chain<H, 1>(Wi, xs, j, ai);
            gemv_chain<H, 1>(Wh, hs[cur], j, ah);
            float gate[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) gate[g] = __fadd_rn(__fadd_rn(ah[0][g], bh[g]), __fadd_rn(ai[0][g], bi[g]));
            lstm_cell_update(gate[0], gate[1], gate[2], gate[3], c[0], h[0]);
            hs[cur ^ 1][0][j] = h[0];
            qv[j] = h[0];                                              // query = hidden (:207)
        }
        __syncthreads();
        cur ^= 1;

        // ---- glimpse rounds (:208-211)
        for (int g = 0; g < n_glimpses; ++g) {
            if constexpr (BAHDANAU) {   // W_query q + b: one fmaf chain per output, inputs ascending
                if (owner) {
                    float acc = 0.0f;
                    const float* wr = gl.wq + (int64_t)j * H;
                    for (int i = 0; i < H; ++i) acc = fmaf(wr[i], qv[i], acc);
                    qw[j] = __fadd_rn(acc, gl.bq[j]);
                }
                __syncthreads();
            }
            attention_logits<H, NT, BAHDANAU>(enc_b, gl, gref_b, qv, qw, 0, L, lg);
            __syncthreads();
            for (int i = j; i < k; i += NT) lg[chosen[i]] = -INFINITY;   // the positions chosen so far (:169-172)
            __syncthreads();
            float m = -INFINITY;
            for (int l = j; l < L; l += NT) m = fmaxf(m, lg[l]);
            m = block_max<NT>(m, red);
            float s = 0.0f;
            for (int l = j; l < L; l += NT) {
                const float e = expf(__fsub_rn(lg[l], m));
                lg[l] = e;
                s = __fadd_rn(s, e);
            }
            s = block_sum<NT>(s, red);
            __syncthreads();
            // query = ref' . softmax (:211): thread j owns output j, positions ascending
            if (owner) {
                const float* rp = (BAHDANAU ? gref_b : enc_b) + j;
                float acc = 0.0f;
                for (int l = 0; l < L; ++l) acc = fmaf(rp[(int64_t)l * H], lg[l] / s, acc);
                qw[j] = acc;                                           // qv is still being read by the others
            }
            __syncthreads();
            if (owner) qv[j] = qw[j];
            __syncthreads();
        }

        // the query the pointer sees (after the glimpse rounds): what the full-length logits of the step are formed from on demand
        if (owner && net.queries) net.queries[((int64_t)b * T + k) * H + j] = qv[j];
        // ---- pointer logits of the step's window (:213; every other position is -inf after :220-222)
        if constexpr (BAHDANAU) {
            if (owner) {
                float acc = 0.0f;
                const float* wr = ptr.wq + (int64_t)j * H;
                for (int i = 0; i < H; ++i) acc = fmaf(wr[i], qv[i], acc);
                qw[j] = __fadd_rn(acc, ptr.bq[j]);
            }
            __syncthreads();
        }
        attention_logits<H, NT, BAHDANAU>(enc_b, ptr, pref_b, qv, qw, k * n_per, n_per, win);
        __syncthreads();

        // ---- C*tanh, latent bias, softmax, first-max argmax
        if (j == 0) {
            const int64_t wbase = ((int64_t)b * T + k) * n_per;
            float best = 0.0f;
            int best_r = -1;
            for (int r = 0; r < n_per; ++r) {
                float v = win[r];
                if (use_tanh) v = __fmul_rn(tanh_c, tanhf(v));
                net.win_logits[wbase + r] = v;
                if (net.latent_win) v = __fadd_rn(v, net.latent_win[wbase + r]);
                win[r] = v;
                if (best_r < 0 || v > best) {   // strict '>' keeps the first maximum
                    best = v;
                    best_r = r;
                }
            }
            float denom = 0.0f;
            for (int r = 0; r < n_per; ++r) denom = __fadd_rn(denom, expf(__fsub_rn(win[r], best)));
            float prob = 1.0f / denom;
            if (net.sample) {   // multinomial(1) from the window softmax (modelPN.py:227-228): first r with u < cdf_r — the draw of
                                // decode.hip, from the same counter-based stream (problem b, step k -> counter b * T + k)
                const float u = stream_uniform24(net.sample_seed, (unsigned long long)b * T + k);
                float cdf = 0.0f;
                int pick = -1, last_pos = 0;
                for (int r = 0; r < n_per; ++r) {
                    const float pr = expf(__fsub_rn(win[r], best)) / denom;
                    cdf = __fadd_rn(cdf, pr);
                    if (pr > 0.0f) last_pos = r;
                    if (pick < 0 && u < cdf) pick = r;
                }
                if (pick < 0) pick = last_pos;
                best_r = pick;
                prob = expf(__fsub_rn(win[pick], best)) / denom;
            }
            net.pick_prob[(int64_t)b * T + k] = prob;
            net.idx[(int64_t)b * T + k] = k * n_per + best_r;
            sel = k * n_per + best_r;
            chosen[k] = k * n_per + best_r;
        }
        __syncthreads();

        // ---- gathers: next decoder input (:235) and the action row (:293-295)
        const int64_t row = (int64_t)b * L + sel;
        if (owner) xs[0][j] = net.embedded[row * H + j];
        if (j < 8) net.actions[((int64_t)b * T + k) * 8 + j] = inputs[row * 8 + j];
        __syncthreads();
    }
}

template <int H>
int launch_glimpse(const DecodeNet& net, const AttnSide& ptr, const AttnSide& gl, const float* inputs, int attention,
                   int n_glimpses, float tanh_c, int use_tanh, int32_t B, int32_t T, int32_t n_per, hipStream_t s) {
    constexpr int NT = H < 64 ? 64 : H;
    const size_t dyn = ((size_t)T * n_per + (size_t)T) * 4;
    if (dyn > 120 * 1024) GNNPN_FAIL(GNNPN_E_UNSUP, "pointer_decode_attn: seq_len %d does not fit the LDS buffer", T * n_per);
    if (attention == 1) {
        auto k = pointer_decode_glimpse_kernel<H, true>;
        if (dyn > 48 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess)
            GNNPN_FAIL(GNNPN_E_LAUNCH, "pointer_decode_attn: cannot reserve %zu B of LDS", dyn);
        hipLaunchKernelGGL(k, dim3(B), dim3(NT), dyn, s, net, ptr, gl, inputs, n_glimpses, tanh_c, use_tanh, B, T, n_per);
    } else {
        auto k = pointer_decode_glimpse_kernel<H, false>;
        if (dyn > 48 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess)
            GNNPN_FAIL(GNNPN_E_LAUNCH, "pointer_decode_attn: cannot reserve %zu B of LDS", dyn);
        hipLaunchKernelGGL(k, dim3(B), dim3(NT), dyn, s, net, ptr, gl, inputs, n_glimpses, tanh_c, use_tanh, B, T, n_per);
    }
    return GNNPN_OK;
}

}  // namespace

extern "C" int gnnpn_pointer_decode_attn_f32(const gnnpn_decode_net_t* net_in, const gnnpn_attention_t* attn,
                                             const float* inputs, float tanh_c, int use_tanh, int32_t B, int32_t T,
                                             int32_t n_per, int32_t H, void* stream) {
    GNNPN_REQUIRE(B >= 0 && T >= 1 && n_per >= 1 && n_per <= 64, "pointer_decode_attn: B >= 0, T >= 1, 1 <= n_per <= 64");
    if (B == 0) return GNNPN_OK;
    GNNPN_REQUIRE(net_in && attn && inputs, "pointer_decode_attn: null argument");
    GNNPN_REQUIRE(H == 256 || H == 32, "pointer_decode_attn: built for H = 256 and H = 32, got %d", H);
    GNNPN_REQUIRE(attn->attention == 0 || attn->attention == 1, "pointer_decode_attn: attention 0 ('Dot') or 1 ('Bahdanau')");
    GNNPN_REQUIRE(attn->n_glimpses >= 0, "pointer_decode_attn: n_glimpses >= 0");
    DecodeNet net;
    static_assert(sizeof(net) == sizeof(*net_in), "layout");
    memcpy(&net, net_in, sizeof(net));
    GNNPN_REQUIRE(net.embedded && net.enc_out && net.h0 && net.c0 && net.start && net.wih && net.whh && net.bih && net.bhh,
                  "pointer_decode_attn: embedded, enc_out, h0, c0, start and the decoder weights are required");
    GNNPN_REQUIRE(net.idx && net.win_logits && net.pick_prob && net.actions, "pointer_decode_attn: output pointers required");
    GNNPN_REQUIRE(net.sample == 0 || net.sample == 1, "pointer_decode_attn: sample must be 0 (greedy) or 1 (multinomial)");
    AttnSide ptr{attn->pointer_wq, attn->pointer_bq, attn->pointer_ref, attn->pointer_v};
    AttnSide gl{attn->glimpse_wq, attn->glimpse_bq, attn->glimpse_ref, attn->glimpse_v};
    if (attn->attention == 1) {
        GNNPN_REQUIRE(ptr.wq && ptr.bq && ptr.ref && ptr.v, "pointer_decode_attn: 'Bahdanau' needs the pointer's W_query, bias, W_ref(enc_out) and V");
        if (attn->n_glimpses > 0)
            GNNPN_REQUIRE(gl.wq && gl.bq && gl.ref && gl.v, "pointer_decode_attn: 'Bahdanau' glimpses need the glimpse's W_query, bias, W_ref(enc_out) and V");
    }
    if (B == 0) return GNNPN_OK;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (H == 256) rc = launch_glimpse<256>(net, ptr, gl, inputs, attn->attention, attn->n_glimpses, tanh_c, use_tanh, B, T, n_per, s);
    else rc = launch_glimpse<32>(net, ptr, gl, inputs, attn->attention, attn->n_glimpses, tanh_c, use_tanh, B, T, n_per, s);
    if (rc != GNNPN_OK) return rc;
    GNNPN_CHECK_LAUNCH("pointer_decode_attn_f32");
    return GNNPN_OK;
}
