// Greedy pointer decode, cooperative form (v2): both pointer networks (Low, High) in ONE launch.
//
// Same ownership as the cooperative encoder (lstm_coop.hip): a group of G = 8 workgroups owns a
// tile of 16 problems; member m keeps the decoder's W_ih and W_hh rows of hidden units
// [32m, 32m+32) in registers (2 x 128 VGPRs/lane of fp32 MFMA B-fragments).  One hand-off per step:
//
//   iteration k:  sweep {h_{k-1} (all 256 units), the 8 members' partial attention dots, the Low
//                 net's window logits (High net only)}  ->  logits, first-max argmax = idx_{k-1}
//                 -> gather x_k = embedded[idx_{k-1}] (in flight under the W_hh.h MFMAs)
//                 -> W_hh.h MFMAs, W_ih.x MFMAs (two independent k-ordered fma chains, added as the
//                    reference adds its two linears)  ->  cell update (c in registers)
//                 -> publish the h_k slice; partial dots of the step-k window rows against the OWN
//                    h_k slice (dot = sum of 8 partials of 32, in member order) -> publish them.
//
// With the folded input side (net.xw_fold != NULL) the decoder input x_k = embedding2(inputs[pick])
// never exists: W_ih.(W_e r + b_e) + b_ih = (W_ih W_e) r + (W_ih b_e + b_ih) is evaluated from the raw
// 8-feature row r with 4 MFMAs per step instead of 128 (same identity as the encoder's folded input
// projection; step 0 uses the precomputed W_ih.start + b_ih).
//
// The attention therefore needs no second hand-off: the partial dots ride the same exchange as h.
// Every member computes every row's argmax (identical arithmetic on identical data), so all agree
// without exchanging indices; member 0 writes the outputs.  The Low net never waits for the High
// net; the High net reads Low's window logits through write-once tagged granules, so the two run
// one step apart in the same launch.  Granule hand-off, double buffering by step parity, bounded
// spins and error word: as in lstm_coop.hip.
#include "common.h"
#include "recurrent.h"
#include "decode_shared.h"
#include "coop_common.h"
#include "lstm_shared.h"

namespace {
constexpr int H = 256;
constexpr int G = 8;
constexpr int ROWS = 16;
constexpr int UNITS = H / G;      // 32
constexpr int LDH = 258;
constexpr int KMAX = 16;          // candidates per category the cooperative form is built for
constexpr unsigned SPIN_LIMIT = 400000;
}  // namespace

// ---- failure record (diagnosis of a timed-out sweep; written only on the failure path, read by gnnpn_decode_diag) ----
// [0] failures so far (all launches) | per failing wave, first 31 of them, 16 words: group, member, tile, k, wave, tag,
// mask of members whose h granules are missing, mask of members whose partial dots are missing, missing latent lanes,
// the 8 per-XCD claim counters of the launch packed as bytes (2 words), realtime lo/hi, gpx, blockIdx.x
__device__ unsigned g_dec_diag[512];
__device__ __forceinline__ void decode_diag_record(int group, int member, int tile, const unsigned* err, int gpx) {
    const unsigned n = atomicAdd(g_dec_diag, 1u);
    if (n < 31) {
        unsigned* r = g_dec_diag + 16 * (n + 1);
        const unsigned* cnt = err + COOP_XCDCNT_OFFSET / 4;
        unsigned c0 = 0, c1 = 0;
        for (int x = 0; x < 4; ++x) {
            c0 |= (cnt[COOP_XCD_STRIDE * x] & 0xffu) << (8 * x);
            c1 |= (cnt[COOP_XCD_STRIDE * (4 + x)] & 0xffu) << (8 * x);
        }
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        r[0] = group; r[1] = member; r[2] = tile; r[9] = c0; r[10] = c1; r[11] = (unsigned)t; r[12] = (unsigned)(t >> 32);
        r[13] = gpx; r[14] = blockIdx.x; r[15] = err[0];
    }
}

// max / sum over each row of 16 lanes by DPP rotations (every lane ends with the row's result)
__device__ __forceinline__ int ror16(int v, int n) {
    switch (n) {
        // (a rotation has a source for every lane: bound_ctrl only spares the compiler an `old = 0` move per use)
        case 1: return __builtin_amdgcn_update_dpp(0, v, 0x121, 0xF, 0xF, true);
        case 2: return __builtin_amdgcn_update_dpp(0, v, 0x122, 0xF, 0xF, true);
        case 4: return __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, true);
        default: return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);
    }
}

// DIAG: diagnostic build with phase stamps (tools/stamp_decode.py); production carries none of it.
// SPLIT: W_hh.h from exact three-piece fp16 operands (coop_common.h); everything else as in the fp32 form
// OCC: workgroups per CU the build is sized for — 1 (512 registers: fastest alone) or 2 (256 registers: shares the
// CU with a workgroup of another launch, pipeline.PipelinedRunner)
// SAMPLE: the build that can draw the pick from the window softmax (gnnpn_decode_net_t.sample); the greedy builds carry
// none of that code
template <bool FOLDX, bool DIAG, bool SPLIT, int OCC, int EVH_ = (OCC == 2 ? 2 : 1), bool SAMPLE = false>
__global__ __launch_bounds__(256, OCC) void pointer_decode_coop_kernel(DecodeArgs a, u64* __restrict__ xh,
                                                                     u64* __restrict__ xp, u64* __restrict__ xl,
                                                                     unsigned* __restrict__ err, unsigned* __restrict__ sticky,
                                                                     int n_nets, int groups_per_net, int gpx, int ablate_arg, unsigned* __restrict__ seats) {
    const int ablate = DIAG ? ablate_arg : (ablate_arg & 128);
    __shared__ __attribute__((aligned(16))) float hs[SPLIT ? 3 * SPLIT_TILE / 2 : ROWS * LDH16];   // fp32 tile (k-quarter-major, stride LDT) | three fp16 piece tiles (stride LDH16 halfs)
    __shared__ float xs[FOLDX ? 1 : ROWS * LDH];
    __shared__ __attribute__((aligned(16))) float hsl[ROWS][UNITS + 4];
    __shared__ __attribute__((aligned(16))) float part_lin[4][G * 4 * KMAX];   // per wave: the partial dots of its 4 rows, [row%4][cand][member]
    __shared__ float lat[ROWS][KMAX];
    __shared__ int sel[ROWS];
    __shared__ float xin[FOLDX ? 1 : ROWS][8];
    __shared__ int abort_flag;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane >> 4, c = lane & 15;
    __shared__ int place[2];
    int group, member;
    if (!coop_place<G>(err, gpx, place, group, member, seats, false, sticky)) return;   // surplus workgroup of the over-subscribed launch (coop_common.h)
    const int net_id = group / groups_per_net, gi = group % groups_per_net;
    if (net_id >= n_nets) return;
    const DecodeNet& net = a.net[net_id];
    if (tid == 0) abort_flag = 0;
    __syncthreads();
    const bool same_xcd = !(ablate & 128);   // h and partial-dot granules stay inside the group's XCD
    if (tid == 0 && same_xcd) atomicAdd(err + COOP_PLACED_OFFSET / 4 + COOP_XCD_STRIDE * xcc_id(), 1u);   // statistics: workgroups on the same-XCD fast path (per XCD: its own line)

    const int B = a.B, T = a.T, K = a.K, L = T * K;
    const bool latent_in_launch = net.latent_from >= 0;
    const bool has_lat = latent_in_launch || net.latent_win;
    bool publishes_latent = false;
    for (int n = 0; n < n_nets; ++n) publishes_latent |= (a.net[n].latent_from == net_id);

    u64* xh_g = xh + (size_t)group * (2 * ROWS * H);
    u64* xp_g = xp + (size_t)group * (2 * G * ROWS * K);

    const int unit = member * UNITS + wave * 8 + (c & 7);
    int wrow[2];
    wrow[0] = (0 + (c >> 3)) * H + unit;
    wrow[1] = (2 + (c >> 3)) * H + unit;
    float wBh[SPLIT ? 1 : 2][SPLIT ? 1 : 64], wBx[FOLDX ? 1 : 2][FOLDX ? 1 : 64], bh[2], bi[2], wXf[2][2], sg[2];
    f16x8 wH16[SPLIT ? 2 : 1][8], wL16[SPLIT ? 2 : 1][8];   // exact split: pieces 0 and 1 (piece 2: LDS bytes, coop_common.h)
    float winv[2] = {1.0f, 1.0f};
    __shared__ __attribute__((aligned(16))) unsigned wts[SPLIT ? SPLIT_WT_DWORDS : 4];
    unsigned* wt_lane = wts + (SPLIT ? (wave * 8 * 64 + lane) * 4 : 0);
    const uint4* __restrict__ Wsplit = SPLIT ? static_cast<const uint4*>(net.whh_split) : nullptr;   // uniform: packed once per model, or split here
    if constexpr (SPLIT) {
        if (Wsplit) load_split_weights(Wsplit + (size_t)member * SPLIT_PACK_U4_PER_MEMBER, tid, wH16, wL16, wt_lane, winv);
    }
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        const int gate = wrow[tl] / H, u = wrow[tl] % H;
        bh[tl] = net.bhh[wrow[tl]];
        if constexpr (SPLIT) {
            if (!Wsplit) winv[tl] = split_weights<H>(net.whh, gate, u, kq, wH16[tl], wL16[tl], wt_lane + 2 * tl);
        } else {
#pragma unroll
            for (int kk = 0; kk < 64; ++kk) wBh[tl][kk] = net.whh[((size_t)(kk * 4 + gate) * H + u) * 4 + kq];
        }
        if constexpr (FOLDX) {   // B-fragments of (W_ih W_e) [4H,8], its bias, and the step-0 gates W_ih.start + b_ih
            wXf[tl][0] = net.xw_fold[wrow[tl] * 8 + kq];
            wXf[tl][1] = net.xw_fold[wrow[tl] * 8 + 4 + kq];
            bi[tl] = net.xb_fold[wrow[tl]];
            sg[tl] = net.start_fold[wrow[tl]];
        } else {
            bi[tl] = net.bih[wrow[tl]];
            wXf[tl][0] = wXf[tl][1] = sg[tl] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < 64; ++kk)
                wBx[tl][kk] = net.wih[((size_t)(kk * 4 + gate) * H + u) * 4 + kq];
        }
    }
    // unfolded path, in-kernel embedding2 of the picked row (net.embedded == nullptr): column tid of emb_w
    float ew[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, eb = 0.f;
    if (!FOLDX && !net.embedded) {
#pragma unroll
        for (int i = 0; i < 8; ++i) ew[i] = net.emb_w[tid * 8 + i];
        eb = net.emb_b[tid];
    }

    const int n_tiles = (B + ROWS - 1) / ROWS;
    unsigned step = 0;   // publish counter: tag = step+1, parity = step&1
    unsigned tiles_done = 0;   // proof of work (coop_note_finished)
    for (int tile = gi; tile < n_tiles; tile += groups_per_net) {
        const int b0 = tile * ROWS;
        // cell state / output of the two rows this lane finishes: rows kq*4 + {0,1} (c < 8) or kq*4 + {2,3} (c >= 8)
        const int own0 = kq * 4 + (c < 8 ? 0 : 2);
        f32x2 cst, hl = {0.f, 0.f};
        cst.x = b0 + own0 < B ? net.c0[(int64_t)(b0 + own0) * H + unit] : 0.0f;
        cst.y = b0 + own0 + 1 < B ? net.c0[(int64_t)(b0 + own0 + 1) * H + unit] : 0.0f;
        __syncthreads();   // previous tile is completely done with the LDS arrays
        for (int j = 0; j < ROWS; ++j) {
            const float h0v = (b0 + j < B) ? net.h0[(int64_t)(b0 + j) * H + tid] : 0.0f;
            if constexpr (SPLIT) split_store(reinterpret_cast<_Float16*>(hs) + j * LDH16 + tid, h0v);
            else hs[ht_index(j, tid)] = h0v;
            if (!FOLDX) xs[j * LDH + tid] = net.start[tid];
        }
        __syncthreads();

        for (int k = 0; k <= T; ++k) {
            float4 xg[4];
            float xraw = 0.0f, axf[2] = {0.f, 0.f};
            const bool stamps = DIAG && (ablate & 32);
            u64 st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (stamps) st[0] = phase_stamp();
            if (k > 0) {
                // ---- hand-off of publish #(step-1): h_{k-1}, partial dots, Low's window logits — ONE
                // combined sweep (all loads issued, then all tags checked): one round trip, not three
                const unsigned tag = step;
                const int par = (step - 1) & 1;
                {
                    const u64* src_h = xh_g + par * (ROWS * H) + wave * 4 * H;
                    const u64* src_p = xp_g + par * (G * ROWS * K) + wave * (G * 4 * K);   // [quarter][row%4][cand][member]
                    const u64* src_l = xl + ((size_t)tile * T + (k - 1)) * ROWS * K + wave * 4 * K;
                    const int n_p = G * 4 * K;            // this wave's rows 4w..4w+3 from all members
                    const int n_l = 4 * K;
                    constexpr int NPJ = EVH_ == 2 ? 4 : 8;   // partial-dot granules per lane: G*4*K / 64, K <= 8 in the EVH_ = 2 builds
                    constexpr int NPJ2 = NPJ / 2;            // 16-byte loads: two adjacent granules each (coop_common.h)
                    u32x4 vh[8], vp[NPJ2];
                    unsigned vl = 0;
                    bool ok = false;
                    for (unsigned spins = 0; spins <= SPIN_LIMIT; ++spins) {
                        bool good = true;
                        // pairs of this lane beyond n_p (K < 8 / K < 16) are loaded and never looked at: they lie at most
                        // 4 KB past this wave's segment, inside the workspace (COOP_OVERREAD_BYTES of slack at its end)
                        if (latent_in_launch) {   // Low's window logits ride the same statement (one round trip, not two)
                            u64 x;
                            if constexpr (NPJ2 == 2) granule_load2_x8_x2_lat(vh, vp, x, uniform_ptr(src_h), uniform_ptr(src_p), uniform_ptr(src_l), 16u * lane);
                            else granule_load2_x8_x4_lat(vh, vp, x, uniform_ptr(src_h), uniform_ptr(src_p), uniform_ptr(src_l), 16u * lane);
                            vl = (unsigned)x;
                            good &= (lane >= n_l) | ((unsigned)(x >> 32) == 1u);
                        } else {
                            if constexpr (NPJ2 == 2) granule_load2_x8_x2(vh, vp, uniform_ptr(src_h), uniform_ptr(src_p), 16u * lane);
                            else granule_load2_x8_x4(vh, vp, uniform_ptr(src_h), uniform_ptr(src_p), 16u * lane);
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            if constexpr (SPLIT) good &= split_pair_tagged(vh[j], tag);
                            else good &= (vh[j].y == tag) & (vh[j].w == tag);
                        }
#pragma unroll
                        for (int j = 0; j < NPJ2; ++j)
                            if (2 * (lane + 64 * j) < n_p) good &= (vp[j].y == tag) & (vp[j].w == tag);
                        if (__all(good)) {
                            ok = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (!ok) abort_flag = 1;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int i = 2 * (j * 64 + lane);          // even: i and i + 1 share a row
                        if constexpr (SPLIT) {
                            split_pair_to_lds(reinterpret_cast<_Float16*>(hs) + (wave * 4 + (i >> 8)) * LDH16 + (i & 255), vh[j]);
                        } else {
                            float* d = &hs[ht_index(wave * 4 + (i >> 8), i & 255)];   // units i, i+1: 64 floats apart
                            d[0] = __uint_as_float(vh[j].x);
                            d[64] = __uint_as_float(vh[j].z);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < NPJ2; ++j) {
                        if (2 * (lane + 64 * j) < n_p)
                            *reinterpret_cast<float2*>(&part_lin[wave][2 * (lane + 64 * j)]) =
                                make_float2(__uint_as_float(vp[j].x), __uint_as_float(vp[j].z));
                    }
                    if (lane < n_l) {
                        float lv = 0.0f;
                        if (latent_in_launch) {
                            lv = __uint_as_float(vl);
                        } else if (net.latent_win) {
                            const int b = b0 + wave * 4 + lane / K;
                            lv = b < B ? net.latent_win[((int64_t)b * T + (k - 1)) * K + lane % K] : 0.0f;
                        }
                        lat[wave * 4 + lane / K][lane % K] = lv;
                    }
                }
                if (stamps) st[1] = phase_stamp();

                // ---- logits, softmax and first-max argmax inside the wave: wave w owns rows 4w..4w+3,
                // 16 lanes per row, lane = candidate (K <= 16).  Row-wide max / sum by DPP rotations.
                // No workgroup barrier in front of it: the partial dots and latent logits of these rows were staged by
                // THIS wave; the h tile of the other waves is first needed by the MFMAs behind the barrier below.
                {
                    const int row = wave * 4 + kq, b = b0 + row, r = c;
                    const bool live = r < K;
                    const float lv = lat[row][r];   // read beside the partial dots (r < KMAX: in range; used only where live)
                    float dot = 0.0f;
                    if (live) {
                        // the G members' partial dots of (row, candidate) are contiguous: two 16-byte LDS reads
                        const float4 p0 = *reinterpret_cast<const float4*>(&part_lin[wave][(kq * K + r) * G]);
                        const float4 p1 = *reinterpret_cast<const float4*>(&part_lin[wave][(kq * K + r) * G + 4]);
                        dot = __fadd_rn(__fadd_rn(__fadd_rn(p0.x, p0.y), p0.z), p0.w);                      // member order
                        dot = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(dot, p1.x), p1.y), p1.z), p1.w);
                    }
                    // C*tanh with the device-library tanhf: these values decide the pick
                    float v = a.use_tanh ? __fmul_rn(a.tanh_c, tanhf(dot)) : dot;
                    if (live && member == 0) {
                        if (b < B) net.win_logits[((int64_t)b * T + (k - 1)) * K + r] = v;
                        if (publishes_latent)
                            granule_store(xl + (((size_t)tile * T + (k - 1)) * ROWS + row) * K + r, 1u, v);
                    }
                    v = (has_lat && live) ? __fadd_rn(v, lv) : v;
                    // key: larger logit wins, ties -> lower candidate index (torch.max returns the first)
                    const unsigned hi = live ? float_order_key(v) : 0u;
                    unsigned long long key = ((unsigned long long)hi << 32) | (unsigned)(15 - r);
#pragma unroll
                    for (int n = 1; n <= 8; n <<= 1) {
                        const unsigned olo = (unsigned)ror16((int)(unsigned)key, n);
                        const unsigned ohi = (unsigned)ror16((int)(unsigned)(key >> 32), n);
                        const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
                        key = o > key ? o : key;
                    }
                    const int best_r = 15 - (int)(key & 0xffffffffu);
                    // best logit value = the winner's v: the inverse of its order key (the map is a bijection; a second
                    // rotation-reduce for the maximum cost four more dependent DPP stages per step)
                    const unsigned bkey = (unsigned)(key >> 32);
                    const float best = __uint_as_float((bkey & 0x80000000u) ? (bkey ^ 0x80000000u) : ~bkey);
                    float e = live ? expf(__fsub_rn(v, best)) : 0.0f;
                    const float e_own = e;
#pragma unroll
                    for (int n = 8; n >= 1; n >>= 1) e = __fadd_rn(e, __int_as_float(ror16(__float_as_int(e), n)));
                    int pick_r = best_r;
                    float pick_p = 1.0f / e;
                    if constexpr (SAMPLE) {
                        if (net.sample) {   // multinomial(1) from the window softmax (modelPN.py:227-228)
                            // every lane r of the row accumulates cdf_r = p_0 + ... + p_r in candidate order (the order the
                            // oracle's running sum uses); the pick is the first r with u < cdf_r
                            const float pr = e_own / e;
                            float cdf = 0.0f;
                            for (int j = 0; j < K; ++j) {
                                const float pj = __shfl(pr, (lane & ~15) + j, 64);
                                if (j <= r) cdf = __fadd_rn(cdf, pj);
                            }
                            const float u = stream_uniform24(net.sample_seed, (unsigned long long)min(b, B - 1) * T + (k - 1));
                            const unsigned below = (unsigned)(__ballot(live && !(u < cdf)) >> (16 * kq)) & 0xffffu;
                            const unsigned pos = (unsigned)(__ballot(live && pr > 0.0f) >> (16 * kq)) & 0xffffu;
                            int cand = min((int)__popc(below), K - 1);           // cdf is non-decreasing: count = first r with u < cdf_r
                            const unsigned upto = pos & ((2u << cand) - 1u);     // rounding left none: the last r with p_r > 0
                            if (!((pos >> cand) & 1u) && upto) cand = 31 - __clz(upto);
                            pick_r = cand;
                            pick_p = __shfl(pr, (lane & ~15) + cand, 64);
                        }
                    }
                    if (r == 0) {
                        sel[row] = (k - 1) * K + pick_r;
                        if (member == 0 && b < B) {
                            net.pick_prob[(int64_t)b * T + (k - 1)] = pick_p;
                            net.idx[(int64_t)b * T + (k - 1)] = (k - 1) * K + pick_r;
                        }
                    }
                }
                __syncthreads();   // the h tile of every wave and sel[] are complete
                if (abort_flag) break;
                if constexpr (FOLDX) {
                    // raw 8-feature row of the pick as MFMA A-fragments (row c), in flight under the W_hh.h MFMAs.
                    // Unconditional loads (row clamped, zeroed at the use): a branch here makes the compiler wait
                    // for them at the join, in front of the MFMA chain.  The same registers ARE the action row
                    // (lane (c, kq) holds features kq and 4+kq of row c): wave 0 of member 0 stores them below,
                    // once they have arrived, instead of a load -> wait -> store on the group's critical path.
                    const int bc = min(b0 + c, B - 1);
                    const float* rowp = a.inputs + ((int64_t)bc * L + sel[c]) * 8;
                    axf[0] = rowp[kq];
                    axf[1] = rowp[4 + kq];
                } else if (member == 0 && tid < ROWS * 8) {
                    const int row = tid >> 3, b = b0 + row;
                    if (b < B)
                        net.actions[((int64_t)b * T + (k - 1)) * 8 + (tid & 7)] =
                            a.inputs[((int64_t)b * L + sel[row]) * 8 + (tid & 7)];
                }
                if (k == T) {
                    if (FOLDX && member == 0 && wave == 0 && b0 + c < B) {   // last pick: nothing left to hide the load behind
                        float* act = net.actions + ((int64_t)(b0 + c) * T + (k - 1)) * 8;
                        act[kq] = axf[0];
                        act[4 + kq] = axf[1];
                    }
                    break;
                }
                if (stamps) st[2] = phase_stamp();
                // decoder input of step k, in flight under the W_hh.h MFMAs
                if constexpr (FOLDX) {
                } else if (net.embedded) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = tid + 256 * j, row = f >> 6, q4 = f & 63, b = b0 + row;
                        xg[j] = b < B ? *reinterpret_cast<const float4*>(net.embedded + ((int64_t)b * L + sel[row]) * H + q4 * 4)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                } else if (tid < ROWS * 8) {   // only the raw 8-feature row of the pick is fetched
                    const int row = tid >> 3, b = b0 + row;
                    xraw = b < B ? a.inputs[((int64_t)b * L + sel[row]) * 8 + (tid & 7)] : 0.0f;
                }
            }

            // this step's window rows, own 32-unit slice: thread (row, cand) holds 32 floats
            // (OCC == 2: two threads per (row, cand), 16 floats each — half the registers; K <= 8 there)
            constexpr int EVH = EVH_, EVN = 8 / EVH;   // EVH = 2 needs 2*16*K <= 256 threads, i.e. K <= 8
            float4 ev[EVN];
            const int ppair = tid / EVH, phalf = tid - ppair * EVH;
            const int prow = ppair / K, pcand = ppair - prow * K;
            const bool pdot = ppair < ROWS * K;
            if (pdot) {
                const int b = b0 + prow;
                const float* src = net.enc_out + ((int64_t)b * L + (int64_t)k * K + pcand) * H + member * UNITS + phalf * (4 * EVN);
#pragma unroll
                for (int j = 0; j < EVN; ++j)
                    ev[j] = b < B ? *reinterpret_cast<const float4*>(src + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
            }

            // ---- decoder LSTM cell: W_hh.h and the input side as independent fma chains per gate column
            f32x4 ah0 = {0.f, 0.f, 0.f, 0.f}, ah1 = ah0, ax0 = ah0, ax1 = ah0;
            if constexpr (SPLIT) {
                f32x4 acc[2] = {ah0, ah1};
                split_chain(reinterpret_cast<const _Float16*>(hs) + c * LDH16 + 8 * kq, wH16, wL16, wt_lane, winv, acc);
                ah0 = acc[0];
                ah1 = acc[1];
            } else {
                mfma_chain_pair<LDT, OCC == 2 ? 8 : 16, true>(hs, c, kq, wBh[0], wBh[SPLIT ? 0 : 1], ah0, ah1);
            }
            if (stamps) {
                asm volatile("" ::"v"(ah0[0]), "v"(ah1[0]));
                st[3] = phase_stamp();
            }
            float gx[2][4];
            if constexpr (FOLDX) {
                if (k > 0) {
                    ax0 = __builtin_amdgcn_mfma_f32_16x16x4f32(axf[0], wXf[0][0], ax0, 0, 0, 0);
                    ax1 = __builtin_amdgcn_mfma_f32_16x16x4f32(axf[0], wXf[1][0], ax1, 0, 0, 0);
                    ax0 = __builtin_amdgcn_mfma_f32_16x16x4f32(axf[1], wXf[0][1], ax0, 0, 0, 0);
                    ax1 = __builtin_amdgcn_mfma_f32_16x16x4f32(axf[1], wXf[1][1], ax1, 0, 0, 0);
                    if (member == 0 && wave == 0 && b0 + c < B) {   // the action row of pick k-1 (see the load above)
                        float* act = net.actions + ((int64_t)(b0 + c) * T + (k - 1)) * 8;
                        act[kq] = axf[0];
                        act[4 + kq] = axf[1];
                    }
                }
                if (stamps) st[4] = st[5] = phase_stamp();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gx[0][r] = k > 0 ? __fadd_rn(ax0[r], bi[0]) : sg[0];
                    gx[1][r] = k > 0 ? __fadd_rn(ax1[r], bi[1]) : sg[1];
                }
            } else {
                if (k > 0) {
                    if (net.embedded) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int f = tid + 256 * j, row = f >> 6, q4 = f & 63;
                            float* d = &xs[row * LDH + q4 * 4];
                            d[0] = xg[j].x; d[1] = xg[j].y; d[2] = xg[j].z; d[3] = xg[j].w;
                        }
                    } else {
                        if (tid < ROWS * 8) xin[tid >> 3][tid & 7] = xraw;
                        __syncthreads();
                        // x[row][tid] = (sum_i inputs[row][i] * emb_w[tid][i], i ascending, from 0) + emb_b[tid]:
                        // the k-ordered chain + bias of gnnpn_linear_f32 -> the same bits as the stored row
#pragma unroll
                        for (int row = 0; row < ROWS; ++row) {
                            float acc = 0.0f;
#pragma unroll
                            for (int i = 0; i < 8; ++i) acc = fmaf(xin[row][i], ew[i], acc);
                            xs[row * LDH + tid] = __fadd_rn(acc, eb);
                        }
                    }
                    __syncthreads();
                }
                if (stamps) st[4] = phase_stamp();
                mfma_chain_pair<LDH>(xs, c, kq, wBx[0], wBx[1], ax0, ax1);
                if (stamps) {
                    asm volatile("" ::"v"(ax0[0]), "v"(ax1[0]));
                    st[5] = phase_stamp();
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gx[0][r] = __fadd_rn(ax0[r], bi[0]);
                    gx[1][r] = __fadd_rn(ax1[r], bi[1]);
                }
            }
            u64* out_h = xh_g + (step & 1) * (ROWS * H);
            {   // one cell update per lane for the two rows it finishes (coop_common.h: cell_update_split)
                f32x2 g0[2], g1[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    g0[q] = (f32x2{ah0[2 * q], ah0[2 * q + 1]} + pk_set(bh[0])) + f32x2{gx[0][2 * q], gx[0][2 * q + 1]};
                    g1[q] = (f32x2{ah1[2 * q], ah1[2 * q + 1]} + pk_set(bh[1])) + f32x2{gx[1][2 * q], gx[1][2 * q + 1]};
                }
                cell_update_split(g0, g1, c < 8, cst, hl);
            }
            {
                u64* dst = out_h + own0 * H + unit;
                if constexpr (SPLIT) {
                    split_granule_store(dst, step + 1, hl.x, same_xcd);
                    split_granule_store(dst + H, step + 1, hl.y, same_xcd);
                } else if (same_xcd) {
                    granule_store_l2(dst, step + 1, hl.x);
                    granule_store_l2(dst + H, step + 1, hl.y);
                } else {
                    granule_store(dst, step + 1, hl.x);
                    granule_store(dst + H, step + 1, hl.y);
                }
                hsl[own0][wave * 8 + (c & 7)] = hl.x;
                hsl[own0 + 1][wave * 8 + (c & 7)] = hl.y;
                if (net.queries) {
                    if (b0 + own0 < B) net.queries[((int64_t)(b0 + own0) * T + k) * H + unit] = hl.x;
                    if (b0 + own0 + 1 < B) net.queries[((int64_t)(b0 + own0 + 1) * T + k) * H + unit] = hl.y;
                }
            }
            if (stamps) {
                asm volatile("" ::"v"(hl.y));
                st[6] = phase_stamp();
            }
            __syncthreads();
            // ---- partial attention dots of the step-k window against the own h_k slice
            {
                float p = 0.0f;
                if (pdot) {
#pragma unroll
                    for (int j = 0; j < EVN; ++j) {
                        const float4 hv = *reinterpret_cast<const float4*>(&hsl[prow][phalf * (4 * EVN) + 4 * j]);
                        p = fmaf(ev[j].x, hv.x, p);
                        p = fmaf(ev[j].y, hv.y, p);
                        p = fmaf(ev[j].z, hv.z, p);
                        p = fmaf(ev[j].w, hv.w, p);
                    }
                }
                if constexpr (EVH == 2)   // lanes 2p, 2p+1 hold the two halves of the slice: low half + high half
                    p = __fadd_rn(p, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false)));
                if (pdot && phalf == 0)
                    granule_publish(xp_g + (step & 1) * (G * ROWS * K) + ((size_t)prow * K + pcand) * G + member, step + 1, p, same_xcd);
            }
            if (stamps) {
                st[7] = phase_stamp();
                if (blockIdx.x == 0 && tid == 0 && k > 0) {
                    u64* prof = reinterpret_cast<u64*>(err) + 4;
                    prof[0] += st[1] - st[0];   // combined sweep + LDS fill
                    prof[1] += st[2] - st[1];   // barrier, logits+argmax, barrier, actions
                    prof[2] += st[3] - st[2];   // gathers issue + W_hh.h MFMAs
                    prof[3] += st[4] - st[3];   // input side: folded MFMAs | x embedding into LDS
                    prof[4] += st[5] - st[4];   // W_ih.x MFMAs (unfolded only)
                    prof[5] += st[6] - st[5];   // cell + publish h
                    prof[6] += st[7] - st[6];   // barrier + partial dots + publish
                    prof[7] += 1;
                }
            }
            ++step;
        }
        if (abort_flag) {
            if (tid == 0) decode_diag_record(group, member, tile, err, gpx);
            break;
        }
        ++tiles_done;
    }
    if (tid == 0) {
        if (abort_flag) coop_raise(err, sticky, 2u, seats);
        coop_note_finished(sticky, GNNPN_STATUS_DEC_FINISHED, tiles_done);
    }
}

extern "C" int64_t gnnpn_pointer_decode_workspace_bytes(int32_t B, int32_t T, int32_t n_per) {
    (void)gnnpn_cu_seat_table();   // callers size their workspace before the first launch and outside any capture: create the seat table here
    const int64_t groups = 64, tiles = (B + ROWS - 1) / ROWS;
    const int64_t a8 = COOP_STATUS_BYTES + groups * 2 * ROWS * H * 8 + groups * 2 * G * ROWS * (int64_t)n_per * 8 +
                       tiles * T * ROWS * (int64_t)n_per * 8 + COOP_OVERREAD_BYTES;
    const int64_t lean = gnnpn_decode_lean_workspace_bytes(B, T, n_per);
    return a8 > lean ? a8 : lean;
}

// device address of the failure record, for the kernels of other translation units (decode_lean.hip)
unsigned* gnnpn_decode_diag_buffer() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_dec_diag)) != hipSuccess) return nullptr;
    return static_cast<unsigned*>(p);
}

extern "C" int gnnpn_decode_diag(uint32_t* out, int32_t n_words, int32_t clear) {
    if (!out || n_words < 0 || n_words > 512) GNNPN_FAIL(GNNPN_E_ARG, "decode_diag: up to 512 words");
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dec_diag), (size_t)n_words * 4) != hipSuccess)
        GNNPN_FAIL(GNNPN_E_LAUNCH, "decode_diag: copy failed");
    if (clear) {
        static const unsigned zeros[512] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_dec_diag), zeros, sizeof(zeros)) != hipSuccess)
            GNNPN_FAIL(GNNPN_E_LAUNCH, "decode_diag: clear failed");
    }
    return GNNPN_OK;
}

bool gnnpn_decode_coop_supported(int32_t H_, int32_t n_per) { return H_ == H && n_per <= KMAX; }

int gnnpn_launch_decode_coop(const DecodeArgs& args, int n_nets, int precision, bool shared_cu, const CoopOpts& opts,
                             void* workspace, int64_t workspace_bytes, hipStream_t s) {
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        GNNPN_FAIL(GNNPN_E_LAUNCH, "pointer_decode: cannot query the device");
    bool fold = args.net[0].xw_fold != nullptr;
    for (int n = 0; n < n_nets; ++n)
        if ((args.net[n].xw_fold != nullptr) != fold)
            GNNPN_FAIL(GNNPN_E_ARG, "pointer_decode: all nets of a call must use the same input-side form");
    {   // the shipped configuration (folded input side, greedy picks) runs on the production build, decode_lean.hip; this file
        // keeps the two builds that one does not cover: the literal two-stage input side and the sampling decoder
        bool any_sample0 = false;
        for (int n = 0; n < n_nets; ++n) any_sample0 |= args.net[n].sample != 0;
        if (fold && !any_sample0 && args.K <= 16)
            return gnnpn_launch_decode_lean(args, n_nets, precision, shared_cu, opts, workspace, workspace_bytes, s);
    }
    const int n_tiles = (args.B + ROWS - 1) / ROWS;
    int gpx = n_cu / (8 * G);
    if (gpx > 8) gpx = 8;
    while (gpx > 1 && (gpx - 1) * 8 >= n_nets * n_tiles) --gpx;
    // coop_place assumes 8 XCDs with workgroup ids dealt round-robin over them (MI355X in SPX mode: 256 CUs); a partitioned
    // device (CPX / DPX / QPX: fewer XCDs) would never finish staffing
    if (gpx < 1 || n_cu < 256) GNNPN_FAIL(GNNPN_E_UNSUP, "pointer_decode: device has %d CUs, the cooperative form is built for 8 XCDs x 32 CUs", n_cu);
    const int groups = gpx * 8;
    if (groups < n_nets) GNNPN_FAIL(GNNPN_E_UNSUP, "pointer_decode: %d groups for %d nets", groups, n_nets);
    const int groups_per_net = groups / n_nets;
    const int64_t h_bytes = (int64_t)groups * 2 * ROWS * H * 8;
    const int64_t p_bytes = (int64_t)groups * 2 * G * ROWS * args.K * 8;
    const int64_t l_bytes = (int64_t)n_tiles * args.T * ROWS * args.K * 8;
    const int64_t need = COOP_STATUS_BYTES + h_bytes + p_bytes + l_bytes + COOP_OVERREAD_BYTES;
    if (!workspace || workspace_bytes < need || !gnnpn_aligned(workspace, 256))
        GNNPN_FAIL(GNNPN_E_ARG, "pointer_decode: workspace of %lld B (256-B aligned) required", (long long)need);
    if (coop_zero_workspace(workspace, (size_t)need, s, opts.sticky, GNNPN_STATUS_DEC_EXPECTED, (unsigned)(G * n_nets * n_tiles),
                            (gnnpn_option_lstm_ablate() & 0x2000) != 0) != hipSuccess)
        GNNPN_FAIL(GNNPN_E_LAUNCH, "pointer_decode: workspace memset failed");
    g_gnnpn_last_units = opts.sticky ? (int64_t)G * n_nets * n_tiles : 0;
    unsigned* p_seats = gnnpn_cu_seat_table();
    if (!p_seats) GNNPN_FAIL(GNNPN_E_LAUNCH, "%s: cannot allocate the seat table", "pointer_decode");
    char* base = static_cast<char*>(workspace);
    u64* p_h = reinterpret_cast<u64*>(base + COOP_STATUS_BYTES);
    u64* p_p = reinterpret_cast<u64*>(base + COOP_STATUS_BYTES + h_bytes);
    u64* p_l = reinterpret_cast<u64*>(base + COOP_STATUS_BYTES + h_bytes + p_bytes);
    unsigned* p_err = reinterpret_cast<unsigned*>(base);
    const int abl = opts.write_through ? 128 : 0;
    unsigned* p_s = opts.sticky;
    if (precision == GNNPN_PREC_SPLIT)
        GNNPN_FAIL(GNNPN_E_UNSUP, "pointer_decode: the exact-split arithmetic is built for the folded input side with greedy picks (decode_lean.hip)");
    if (shared_cu)
        GNNPN_FAIL(GNNPN_E_UNSUP, "pointer_decode: the 2-per-CU build exists for the folded input side with greedy picks only");
    const int lds_kb = opts.lds_kb;
    bool any_sample = false;
    for (int n = 0; n < n_nets; ++n) any_sample |= args.net[n].sample != 0;
    if (any_sample && !fold)
        GNNPN_FAIL(GNNPN_E_UNSUP, "pointer_decode: the sampling build exists for the folded fp32 input side only");
#define GNNPN_DEC8(FOLD_, SAMPLE_)                                                                                                       \
    hipLaunchKernelGGL((pointer_decode_coop_kernel<FOLD_, false, false, 1, 1, SAMPLE_>), dim3(COOP_OVERSUB * groups * G), dim3(256),          \
                       coop_lds_padding((const void*)pointer_decode_coop_kernel<FOLD_, false, false, 1, 1, SAMPLE_>, lds_kb), s, args, p_h, \
                       p_p, p_l, p_err, p_s, n_nets, groups_per_net, gpx, abl, p_seats)
    if (any_sample) GNNPN_DEC8(true, true);       // folded input side, every pick drawn from the window softmax
    else GNNPN_DEC8(false, false);                // the literal two-stage input side (embedding2, then W_ih), greedy
#undef GNNPN_DEC8
    return GNNPN_OK;
}
