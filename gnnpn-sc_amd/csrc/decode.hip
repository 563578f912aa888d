// Greedy pointer decode of one pointer network (v1) and the QoS reward of the decoded
// compositions.
//
// One workgroup owns BT problems for all T steps (no inter-workgroup communication).  Per step:
//   decoder LSTM cell  : two k-ordered fmaf-chain GEMVs (W_ih.x, W_hh.h) streamed from L2
//   attention          : only the step's n_per candidate rows of enc_out matter (everything
//                        outside the window is -inf in the reference, modelPN.py:220-222), so the
//                        kernel reads exactly those rows: one wavefront per (problem,row) dot
//                        product, 16 B per lane, fixed butterfly reduction
//   C*tanh, + latent (High net), softmax over the window, first-max argmax (torch.max on CPU
//   returns the first maximal index), gather of the next decoder input and of the action row.
#include <atomic>

#include "common.h"
#include <string.h>
#include "recurrent.h"
#include "decode_shared.h"
#include "lstm_shared.h"

template <int H, int BT>
__global__ __launch_bounds__((H < 64 ? 64 : H)) void pointer_decode_kernel(
    const float* __restrict__ embedded, const float* __restrict__ enc_out, const float* __restrict__ h0,
    const float* __restrict__ c0, const float* __restrict__ start, const float* __restrict__ wih,
    const float* __restrict__ whh, const float* __restrict__ bih, const float* __restrict__ bhh,
    const float* __restrict__ latent_win, const float* __restrict__ inputs, float tanh_c, int use_tanh,
    int32_t* __restrict__ idx_out, float* __restrict__ win_logits, float* __restrict__ pick_prob,
    float* __restrict__ actions, float* __restrict__ queries, int32_t B, int32_t T, int32_t n_per, int sample,
    unsigned long long sample_seed) {
    constexpr int NT = H < 64 ? 64 : H;
    constexpr int NW = NT / 64;
    __shared__ __attribute__((aligned(16))) float xs[BT][H];
    __shared__ __attribute__((aligned(16))) float hs[2][BT][H];
    __shared__ float lg[BT][64];
    __shared__ int sel[BT];

    const int j = threadIdx.x;
    const bool owner = j < H;
    const int lane = j & 63, wave = j >> 6;
    const int b0 = blockIdx.x * BT;
    const int L = T * n_per;
    const float4* __restrict__ Wi = reinterpret_cast<const float4*>(wih);
    const float4* __restrict__ Wh = reinterpret_cast<const float4*>(whh);

    float bi[4], bh[4], c[BT], h[BT];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        bi[g] = owner ? bih[g * H + j] : 0.0f;
        bh[g] = owner ? bhh[g * H + j] : 0.0f;
    }
#pragma unroll
    for (int p = 0; p < BT; ++p) {
        const bool ok = owner && (b0 + p < B);
        c[p] = ok ? c0[(int64_t)(b0 + p) * H + j] : 0.0f;
        h[p] = ok ? h0[(int64_t)(b0 + p) * H + j] : 0.0f;
        if (owner) {
            hs[0][p][j] = h[p];
            xs[p][j] = start[j];
        }
    }
    __syncthreads();

    int cur = 0;
    for (int k = 0; k < T; ++k) {
        // ---- decoder LSTM cell (modelPN.py:205)
        if (owner) {
            float ai[BT][4], ah[BT][4];
#pragma unroll
            for (int p = 0; p < BT; ++p)
#pragma unroll
                for (int g = 0; g < 4; ++g) ai[p][g] = ah[p][g] = 0.0f;
            gemv_chain<H, BT>(Wi, xs, j, ai);
            gemv_chain<H, BT>(Wh, hs[cur], j, ah);
#pragma unroll
            for (int p = 0; p < BT; ++p) {
                float gate[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    gate[g] = __fadd_rn(__fadd_rn(ah[p][g], bh[g]), __fadd_rn(ai[p][g], bi[g]));
                lstm_cell_update(gate[0], gate[1], gate[2], gate[3], c[p], h[p]);
                hs[cur ^ 1][p][j] = h[p];
                if (queries && b0 + p < B) queries[((int64_t)(b0 + p) * T + k) * H + j] = h[p];
            }
        }
        __syncthreads();
        cur ^= 1;

        // ---- dot-attention logits over the window rows (modelPN.py:111-114): wave per (p,row)
        for (int q = wave; q < BT * n_per; q += NW) {
            const int p = q / n_per, r = q - p * n_per;
            float part = 0.0f;
            if (b0 + p < B) {
                const float* row = enc_out + ((int64_t)(b0 + p) * L + (int64_t)k * n_per + r) * H;
                for (int e = lane * 4; e < H; e += 256) {
                    const float4 ev = *reinterpret_cast<const float4*>(row + e);
                    const float4 hv = *reinterpret_cast<const float4*>(&hs[cur][p][e]);
                    part = fmaf(ev.x, hv.x, part);
                    part = fmaf(ev.y, hv.y, part);
                    part = fmaf(ev.z, hv.z, part);
                    part = fmaf(ev.w, hv.w, part);
                }
            }
            const float dot = wave_sum(part);
            if (lane == 0) lg[p][r] = dot;
        }
        __syncthreads();

        // ---- C*tanh, latent bias, softmax, first-max argmax: one thread per problem (n_per <= 64)
        if (j < BT && b0 + j < B) {
            const int p = j;
            const int64_t wbase = ((int64_t)(b0 + p) * T + k) * n_per;
            float best = 0.0f;
            int best_r = -1;
            for (int r = 0; r < n_per; ++r) {
                float v = lg[p][r];
                if (use_tanh) v = __fmul_rn(tanh_c, tanhf(v));
                win_logits[wbase + r] = v;
                if (latent_win) v = __fadd_rn(v, latent_win[wbase + r]);
                lg[p][r] = v;
                if (best_r < 0 || v > best) {   // strict '>' keeps the first maximum
                    best = v;
                    best_r = r;
                }
            }
            float denom = 0.0f;
            for (int r = 0; r < n_per; ++r) denom = __fadd_rn(denom, expf(__fsub_rn(lg[p][r], best)));
            float prob = 1.0f / denom;                              // exp(best-best)/sum
            if (sample) {   // multinomial(1) from the window softmax (modelPN.py:227-228): first r with u < cdf_r
                const float u = stream_uniform24(sample_seed, (unsigned long long)(b0 + p) * T + k);
                float cdf = 0.0f;
                int pick = -1, last_pos = 0;
                for (int r = 0; r < n_per; ++r) {
                    const float pr = expf(__fsub_rn(lg[p][r], best)) / denom;
                    cdf = __fadd_rn(cdf, pr);
                    if (pr > 0.0f) last_pos = r;
                    if (pick < 0 && u < cdf) pick = r;
                }
                if (pick < 0) pick = last_pos;
                best_r = pick;
                prob = expf(__fsub_rn(lg[p][pick], best)) / denom;
            }
            pick_prob[(int64_t)(b0 + p) * T + k] = prob;
            idx_out[(int64_t)(b0 + p) * T + k] = k * n_per + best_r;
            sel[p] = k * n_per + best_r;
        }
        __syncthreads();

        // ---- gathers: next decoder input (modelPN.py:235) and the action row (:293-295)
#pragma unroll
        for (int p = 0; p < BT; ++p) {
            if (b0 + p < B) {
                const int64_t row = (int64_t)(b0 + p) * L + sel[p];
                if (owner) xs[p][j] = embedded[row * H + j];
                if (j < 8) actions[((int64_t)(b0 + p) * T + k) * 8 + j] = inputs[row * 8 + j];
            }
        }
        __syncthreads();
    }
}

template <int H>
static void launch_decode(const float* embedded, const float* enc_out, const float* h0, const float* c0,
                          const float* start, const float* wih, const float* whh, const float* bih,
                          const float* bhh, const float* latent_win, const float* inputs, float tanh_c,
                          int use_tanh, int32_t* idx, float* win_logits, float* pick_prob, float* actions,
                          float* queries, int32_t B, int32_t T, int32_t n_per, int sample, unsigned long long sample_seed,
                          hipStream_t s) {
    constexpr int NT = H < 64 ? 64 : H;
    int bt = 1;
    while (bt < 4 && B / (bt * 2) >= 256) bt *= 2;
    dim3 grid((B + bt - 1) / bt), block(NT);
#define GNNPN_LAUNCH_DEC(BT_)                                                                                  \
    hipLaunchKernelGGL((pointer_decode_kernel<H, BT_>), grid, block, 0, s, embedded, enc_out, h0, c0, start, wih, \
                       whh, bih, bhh, latent_win, inputs, tanh_c, use_tanh, idx, win_logits, pick_prob, actions,  \
                       queries, B, T, n_per, sample, sample_seed)
    switch (bt) {
        case 1: GNNPN_LAUNCH_DEC(1); break;
        case 2: GNNPN_LAUNCH_DEC(2); break;
        default: GNNPN_LAUNCH_DEC(4); break;
    }
#undef GNNPN_LAUNCH_DEC
}

extern "C" int gnnpn_pointer_decode_f32(int n_nets, const gnnpn_decode_net_t* nets, const float* inputs,
                                        float tanh_c, int use_tanh, int32_t B, int32_t T, int32_t n_per, int32_t H,
                                        int32_t precision, const gnnpn_launch_opts_t* opts_in, void* workspace,
                                        int64_t workspace_bytes, void* stream) {
    g_gnnpn_last_units = 0;
    GNNPN_REQUIRE(B >= 0 && T > 0, "pointer_decode: bad shape");
    if (B == 0) return GNNPN_OK;                    // empty batch: its buffers may be NULL
    GNNPN_REQUIRE(nets && inputs, "pointer_decode: null input");
    const CoopOpts opts = coop_opts(opts_in);
    GNNPN_REQUIRE(opts.impl >= 0 && opts.impl <= 4 && opts.impl != 3, "pointer_decode: opts.impl must be 0 (auto), 1 (streaming), 2 (cooperative, 8-CU groups) or 4 (8-CU groups, 256-register build) — the 16-CU-group form (3) was removed in ABI version 7");
    GNNPN_REQUIRE(opts.lds_kb >= 0 && opts.lds_kb <= 160, "pointer_decode: opts.lds_kb must be 0..160");
    GNNPN_REQUIRE(precision == GNNPN_PREC_F32 || precision == GNNPN_PREC_SPLIT, "pointer_decode: precision must be GNNPN_PREC_F32 or GNNPN_PREC_SPLIT, got %d", precision);
    GNNPN_REQUIRE(n_nets >= 1 && n_nets <= GNNPN_MAX_DECODE_NETS, "pointer_decode: n_nets must be 1..%d",
                  GNNPN_MAX_DECODE_NETS);
    GNNPN_REQUIRE(B >= 0 && T > 0, "pointer_decode: bad shape");
    GNNPN_REQUIRE(n_per >= 1 && n_per <= 64, "pointer_decode: n_per must be in [1,64], got %d", n_per);
    if (H != 256 && H != 32) GNNPN_FAIL(GNNPN_E_UNSUP, "pointer_decode: hidden size %d not built (256, 32)", H);
    DecodeArgs args{};
    for (int n = 0; n < n_nets; ++n) {
        const gnnpn_decode_net_t& d = nets[n];
        GNNPN_REQUIRE(d.enc_out && d.h0 && d.c0 && d.start && d.wih_packed && d.whh_packed && d.bih && d.bhh,
                      "pointer_decode: null input of net %d", n);
        GNNPN_REQUIRE(d.embedded || (d.emb_w && d.emb_b) || d.xw_fold,
                      "pointer_decode: net %d needs embedded, (emb_w, emb_b) or the folded input side", n);
        GNNPN_REQUIRE((d.xw_fold != nullptr) == (d.xb_fold != nullptr) && (d.xw_fold != nullptr) == (d.start_fold != nullptr),
                      "pointer_decode: net %d: xw_fold, xb_fold and start_fold go together", n);
        GNNPN_REQUIRE(d.idx && d.win_logits && d.pick_prob && d.actions, "pointer_decode: null output of net %d", n);
        GNNPN_REQUIRE(d.latent_from < n && d.latent_from >= -1, "pointer_decode: latent_from of net %d must name an "
                      "earlier net of the call", n);
        GNNPN_REQUIRE(!(d.latent_win && d.latent_from >= 0), "pointer_decode: net %d has two latent sources", n);
        GNNPN_REQUIRE(d.sample == 0 || d.sample == 1, "pointer_decode: net %d: sample must be 0 (greedy) or 1 (multinomial)", n);
        GNNPN_REQUIRE(gnnpn_aligned(d.wih_packed, 16) && gnnpn_aligned(d.whh_packed, 16) &&
                          gnnpn_aligned(d.enc_out, 16) && (!d.embedded || gnnpn_aligned(d.embedded, 16)),
                      "pointer_decode: weights / enc_out / embedded must be 16-byte aligned");
        GNNPN_REQUIRE(!d.whh_split || gnnpn_aligned(d.whh_split, 16), "pointer_decode: whh_split must be 16-byte aligned");
        static_assert(sizeof(DecodeNet) == sizeof(gnnpn_decode_net_t), "layout");
        memcpy(&args.net[n], &d, sizeof(DecodeNet));
    }
    args.inputs = inputs;
    args.tanh_c = tanh_c;
    args.use_tanh = use_tanh;
    args.B = B;
    args.T = T;
    args.K = n_per;
    if (B == 0) return GNNPN_OK;
    hipStream_t s = (hipStream_t)stream;
    bool any_sample = false;
    for (int n = 0; n < n_nets; ++n) any_sample |= nets[n].sample != 0;
    if (any_sample && (precision != GNNPN_PREC_F32 || opts.impl > 2))
        GNNPN_FAIL(GNNPN_E_UNSUP, "pointer_decode: sampling is built in the streaming and the 8-CU-group fp32 forms (impl 0-2, "
                   "GNNPN_PREC_F32)");
    const int impl = opts.impl;   // 0 auto, 1 streaming, 2 8-CU groups, 4 8-CU groups sized for 2 per CU
    if (impl != 1 && gnnpn_decode_coop_supported(H, n_per) && (workspace != nullptr || impl >= 2)) {
        // auto / 2: fastest when the launch has the GPU to itself; 4: the 256-register build that shares every SIMD with a wave of
        // ANOTHER cooperative launch in flight on a second stream (what PipelinedRunner selects)
        const int rc = gnnpn_launch_decode_coop(args, n_nets, precision, impl == 4, opts, workspace, workspace_bytes, s);
        // auto mode: a device the cooperative forms are not built for (fewer than 8 XCDs x 32 CUs, a net the
        // cooperative sampling build does not cover) is served by the streaming form below instead of an error
        const bool fall_through = rc == GNNPN_E_UNSUP && impl == 0 && precision == GNNPN_PREC_F32;
        if (rc != GNNPN_OK && !fall_through) return rc;
        if (fall_through) {                                  // said once per process: the form below is ~10 x slower per step
            static std::atomic<bool> said{false};
            if (!said.exchange(true))
                fprintf(stderr, "[gnnpn] pointer_decode: %s — taking the per-workgroup streaming form (about 10 x slower per step)\n", gnnpn_last_error());
        }
        if (rc == GNNPN_OK) {
            GNNPN_CHECK_LAUNCH("pointer_decode_f32(coop)");
            return GNNPN_OK;
        }
    }
    if (precision == GNNPN_PREC_SPLIT)
        GNNPN_FAIL(GNNPN_E_UNSUP, "pointer_decode: the split-operand precision needs the cooperative form (H = 256)");
    for (int n = 0; n < n_nets; ++n) {   // streaming form: one net after the other (Low before High)
        const DecodeNet& d = args.net[n];
        if (!d.embedded) GNNPN_FAIL(GNNPN_E_UNSUP, "pointer_decode: the streaming form needs the embedded tensor");
        const float* latent = d.latent_from >= 0 ? args.net[d.latent_from].win_logits : d.latent_win;
        if (H == 256)
            launch_decode<256>(d.embedded, d.enc_out, d.h0, d.c0, d.start, d.wih, d.whh, d.bih, d.bhh, latent, inputs,
                               tanh_c, use_tanh, d.idx, d.win_logits, d.pick_prob, d.actions, d.queries, B, T, n_per,
                               d.sample, d.sample_seed, s);
        else
            launch_decode<32>(d.embedded, d.enc_out, d.h0, d.c0, d.start, d.wih, d.whh, d.bih, d.bhh, latent, inputs,
                              tanh_c, use_tanh, d.idx, d.win_logits, d.pick_prob, d.actions, d.queries, B, T, n_per,
                              d.sample, d.sample_seed, s);
    }
    GNNPN_CHECK_LAUNCH("pointer_decode_f32");
    return GNNPN_OK;
}

// ---------------------------------------------------------------------------------------------
// Full-length logits of one step (API-compat, not on the fast path): wave per (b,l) row.
__global__ __launch_bounds__(256) void attention_logits_kernel(const float* __restrict__ enc_out,
                                                               const float* __restrict__ queries, int64_t ld_q,
                                                               float tanh_c, int use_tanh, float* __restrict__ logits,
                                                               int64_t n_rows, int32_t L, int32_t H) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int64_t b = row / L;
    const float* e = enc_out + row * H;
    const float* q = queries + b * ld_q;
    float part = 0.0f;
    for (int i = lane * 4; i < H; i += 256) {
        const float4 ev = *reinterpret_cast<const float4*>(e + i);
        const float4 qv = *reinterpret_cast<const float4*>(q + i);
        part = fmaf(ev.x, qv.x, part);
        part = fmaf(ev.y, qv.y, part);
        part = fmaf(ev.z, qv.z, part);
        part = fmaf(ev.w, qv.w, part);
    }
    float v = wave_sum(part);
    if (use_tanh) v = __fmul_rn(tanh_c, tanhf(v));
    if (lane == 0) logits[row] = v;
}

// 'Bahdanau' form of the same (modelPN.py:103-109): V . tanh(qp + ref_l), qp = W_query q + b_query and ref = W_ref(enc_out) + b_ref given
__global__ __launch_bounds__(256) void attention_logits_bahdanau_kernel(const float* __restrict__ ref, const float* __restrict__ qp,
                                                                        int64_t ld_q, const float* __restrict__ v, float tanh_c,
                                                                        int use_tanh, float* __restrict__ logits, int64_t n_rows,
                                                                        int32_t L, int32_t H) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float* r = ref + row * H;
    const float* q = qp + (row / L) * ld_q;
    float part = 0.0f;
    for (int i = lane; i < H; i += 64) part = fmaf(v[i], tanhf(q[i] + r[i]), part);
    float u = wave_sum(part);
    if (use_tanh) u = __fmul_rn(tanh_c, tanhf(u));
    if (lane == 0) logits[row] = u;
}

__global__ void mask_logits_kernel(const int32_t* __restrict__ masked_idx, float* __restrict__ logits, int32_t B,
                                   int32_t L, int32_t n_masked, int32_t ld_idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * n_masked) return;
    const int b = i / n_masked, m = i - b * n_masked;
    const int pos = masked_idx[(int64_t)b * ld_idx + m];
    if (pos >= 0 && pos < L) logits[(int64_t)b * L + pos] = -INFINITY;
}

extern "C" int gnnpn_attention_logits_f32(const float* enc_out, const float* queries, int64_t ld_q,
                                          const int32_t* masked_idx, float tanh_c, int use_tanh, float* logits,
                                          int32_t B, int32_t L, int32_t H, int32_t n_masked, int32_t ld_idx,
                                          void* stream) {
    GNNPN_REQUIRE(B >= 0 && L > 0 && H > 0 && H % 4 == 0 && ld_q >= H && ld_q % 4 == 0, "attention_logits: bad shape");
    if (B == 0) return GNNPN_OK;
    GNNPN_REQUIRE(enc_out && queries && logits, "attention_logits: null operand");
    GNNPN_REQUIRE(n_masked == 0 || (masked_idx && ld_idx >= n_masked), "attention_logits: bad mask list");
    GNNPN_REQUIRE(gnnpn_aligned(enc_out, 16) && gnnpn_aligned(queries, 16), "attention_logits: 16-byte alignment");
    if (B == 0) return GNNPN_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_rows = (int64_t)B * L;
    hipLaunchKernelGGL(attention_logits_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, enc_out, queries,
                       ld_q, tanh_c, use_tanh, logits, n_rows, L, H);
    if (n_masked > 0)
        hipLaunchKernelGGL(mask_logits_kernel, dim3((B * n_masked + 255) / 256), dim3(256), 0, s, masked_idx, logits,
                           B, L, n_masked, ld_idx);
    GNNPN_CHECK_LAUNCH("attention_logits_f32");
    return GNNPN_OK;
}

extern "C" int gnnpn_attention_logits_bahdanau_f32(const float* ref, const float* qp, int64_t ld_q, const float* v,
                                                   const int32_t* masked_idx, float tanh_c, int use_tanh, float* logits, int32_t B,
                                                   int32_t L, int32_t H, int32_t n_masked, int32_t ld_idx, void* stream) {
    GNNPN_REQUIRE(ref && qp && v && logits && B >= 0 && L > 0 && H > 0 && ld_q >= H, "attention_logits_bahdanau: bad argument");
    GNNPN_REQUIRE(n_masked == 0 || (masked_idx && ld_idx >= n_masked), "attention_logits_bahdanau: bad mask");
    if (B == 0) return GNNPN_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_rows = (int64_t)B * L;
    hipLaunchKernelGGL(attention_logits_bahdanau_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, ref, qp, ld_q, v,
                       tanh_c, use_tanh, logits, n_rows, L, H);
    if (n_masked > 0)
        hipLaunchKernelGGL(mask_logits_kernel, dim3((B * n_masked + 255) / 256), dim3(256), 0, s, masked_idx, logits, B, L,
                           n_masked, ld_idx);
    GNNPN_CHECK_LAUNCH("attention_logits_bahdanau_f32");
    return GNNPN_OK;
}

// ---------------------------------------------------------------------------------------------
// QoS reward (modelPN.py:15-72): one WAVE per problem (round 3; one thread per problem took 123 us at T = 1000: its 32-byte
// row loads were a serial chain of round trips).  The 64 lanes fetch 64 consecutive action rows with one coalesced 16-byte
// load each; the three order-dependent accumulations — the two running fp32 products of np.cumprod (:20) and the fp64 sum —
// then run over the lanes' values in step order on the scalar-broadcast path (v_readlane), exactly the sequence of
// operations of the one-thread form; the count of real rows and the minimum are order-free and reduce across the wave.
__global__ __launch_bounds__(256) void qos_reward_kernel(const float* __restrict__ actions, float* __restrict__ R, int32_t B,
                                                        int32_t T, int level) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;                                            // the whole wave
    const float* a = actions + (int64_t)b * T * 8;
    const float lo0 = a[4], hi0 = a[5], lo1 = a[6], hi1 = a[7];    // step-0 row (:51-54)
    float prod2 = 1.0f, prod3 = 1.0f, mn = INFINITY;               // 1.0f * z == z: the first row needs no special case
    double sum0 = 0.0;
    int n_real = 0;
    // 64 rows per batch: one 16-byte load per lane (from a clamped row; lanes beyond T are neutralised when the batch is
    // consumed), the NEXT batch requested before this batch's serial chain starts — two named buffers, so that no register copy
    // makes the chain wait for the load it is supposed to hide
    auto fetch = [&](int t) { return *reinterpret_cast<const float4*>(a + (int64_t)min(t, T - 1) * 8); };
    auto consume = [&](float4 q, int t0) {
        if (t0 + lane >= T) q = make_float4(0.0f, INFINITY, 1.0f, 1.0f);
        n_real += q.x > 0.0f;                                      // :26-28 (a padded lane holds 0)
        mn = fminf(mn, q.y);
        const double xd = (double)q.x;                             // np.sum(float32) is pairwise; fp64 then one rounding
        const int lo = __double2loint(xd), hi = __double2hiint(xd);
        const int zi = __float_as_int(q.z), wi = __float_as_int(q.w);
        const int cnt = min(64, T - t0);
#define GNNPN_REWARD_STEP(i)                                                                                      \
    prod2 = __fmul_rn(prod2, __int_as_float(__builtin_amdgcn_readlane(zi, i)));                                   \
    prod3 = __fmul_rn(prod3, __int_as_float(__builtin_amdgcn_readlane(wi, i)));                                   \
    sum0 += __hiloint2double(__builtin_amdgcn_readlane(hi, i), __builtin_amdgcn_readlane(lo, i));
        if (cnt == 64) {                                           // strictly in step order; lane numbers as immediates
#pragma unroll
            for (int i = 0; i < 64; ++i) { GNNPN_REWARD_STEP(i) }
        } else {
            for (int i = 0; i < cnt; ++i) { GNNPN_REWARD_STEP(i) }
        }
#undef GNNPN_REWARD_STEP
    };
    float4 qa = fetch(lane), qb;
    for (int t0 = 0; t0 < T; t0 += 128) {
        qb = fetch(t0 + 64 + lane);
        consume(qa, t0);
        if (t0 + 64 >= T) break;
        qa = fetch(t0 + 128 + lane);
        consume(qb, t0 + 64);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n_real += __shfl_xor(n_real, off, 64);
        mn = fminf(mn, __shfl_xor(mn, off, 64));
    }
    if (lane != 0) return;
    int violate = 0;
    if (prod2 < lo0 || prod2 > hi0) ++violate;    // :23
    if (prod3 < lo1 || prod3 > hi1) ++violate;
    if (level == 0) {
        R[b] = (float)violate;
        return;
    }
    // objFunc = (sum/n + 1 - min)/2 in fp32 (:29), then round(violate + obj, 5) (:61)
    float obj = (float)sum0 / (float)n_real;
    obj = __fadd_rn(obj, 1.0f);
    obj = __fsub_rn(obj, mn);
    obj = obj / 2.0f;
    const double v = (double)violate + (double)obj;
    R[b] = (float)(rint(v * 1e5) / 1e5);
}

extern "C" int gnnpn_qos_reward_f32(const float* actions, float* R, int32_t B, int32_t T, int level,
                                    void* stream) {
    GNNPN_REQUIRE(B >= 0 && T > 0 && (level == 0 || level == 1), "qos_reward: bad argument");
    if (B == 0) return GNNPN_OK;
    GNNPN_REQUIRE(actions && R, "qos_reward: null operand");
    hipLaunchKernelGGL(qos_reward_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, actions, R, B, T,
                       level);
    GNNPN_CHECK_LAUNCH("qos_reward_f32");
    return GNNPN_OK;
}
