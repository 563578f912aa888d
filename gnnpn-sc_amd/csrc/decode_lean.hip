// Greedy pointer decode, cooperative form — the production build (round 3): folded input side, greedy picks, both pointer
// networks in ONE launch.  Replaces /root/reference/src/models/modelPN.py:204-239 (the decode loop) for the shipped
// configuration; decode_coop.hip keeps the general builds (literal two-stage input side, sampling, diagnostics).
//
// Same ownership, hand-off protocol and ARITHMETIC as decode_coop.hip (read its header): a group of 8 workgroups owns a
// tile of 16 problems, member m keeps W_hh of hidden units [32m, 32m+32) in registers, one exchange per step carries
// h_{k-1}, the 8 members' partial attention dots and (High net) the Low net's window logits.  Results are bit-identical
// to that kernel and to the streaming form (tests/test_gpu_pn.py).  What is different is everything around the arithmetic:
//
//  * every global access is a buffer instruction: a 128-bit resource in SGPRs + ONE 32-bit lane offset per stream (constant
//    for the whole launch) + a scalar step offset — no 64-bit vector address arithmetic, no spilled row pointers (the
//    2-per-CU build of decode_coop.hip carried 96-180 B of scratch and 33 64-bit adds per step).  The hand-off sweeps are
//    ordinary (compiler-visible) `buffer_load_dwordx4 ... sc1`: the compiler tracks their completion itself.  Rows past the
//    end of the batch are "out of range" of the resource and read as zero: no clamping code.
//  * the partial dots land in the lane that owns (row, candidate): the exchange buffer is [row][16 candidates][8 members],
//    so a lane's four 16-byte loads ARE its 8 partials in member order — summed in registers, no LDS round trip.
//  * first-max argmax = four v_max_f32 DPP steps + one ballot (was: a 64-bit key and 4 x (2 DPP moves, compare, 2 selects));
//    the softmax normaliser of pick_prob is not needed to pick: it moved to a follow-up kernel over the stored window
//    logits (pick_prob_kernel below — same DPP summation order, same bits).
//  * the raw 8-feature rows of the step's window are staged in LDS a step ahead, so the decoder input of the pick costs an
//    LDS read after the argmax instead of a global round trip (it used to hide under 128 fp32 MFMAs; the exact-split
//    product is too short for that).
//  * GNNPN_PREC_SPLIT: the exact three-piece product of coop_common.h; h travels already split.
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
constexpr int KW = 16;            // candidate slots per row in the exchange buffer (n_per <= 16)
constexpr unsigned SPIN_LIMIT = 400000;
constexpr int XP_GRANULES = ROWS * KW * G;   // partial-dot granules per parity buffer

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
constexpr int AUX_SC1 = 16;       // cache-policy bit sc1: agent scope (a load reads L2, a store writes through)


__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float dpp_ror(float v, int n) {
    switch (n) {
        case 1: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xF, 0xF, true));
        case 2: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xF, 0xF, true));
        case 4: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, true));
        default: return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true));
    }
}
__device__ __forceinline__ void store_granule(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, u64 g, bool same_xcd) {
    const u32x2 v = {(unsigned)g, (unsigned)(g >> 32)};
    // one 8-byte store per granule.  Same XCD: no sc bits — the store lands in the group's L2, the coherence point of an
    // XCD (its peers read with sc1 = past their L1).  Otherwise: sc1 (write-through, agent scope).
    if (same_xcd) __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, AUX_SC1);
}
// failure record of a timed-out sweep (read by gnnpn_decode_diag; layout: decode_coop.hip) — failure path only
__device__ __forceinline__ void record_failure(unsigned* diag, const unsigned* err, int group, int member, int tile, int k, int wave,
                                               unsigned tag, unsigned h_miss, unsigned p_miss, unsigned l_miss, int gpx) {
    if ((threadIdx.x & 63) != 0 || !diag) return;
    const unsigned n = atomicAdd(diag, 1u);
    if (n >= 31) return;
    unsigned* rec = diag + 16 * (n + 1);
    const unsigned* cnt = err + COOP_XCDCNT_OFFSET / 4;
    unsigned c0 = 0, c1 = 0;
    for (int x = 0; x < 4; ++x) {
        c0 |= (cnt[COOP_XCD_STRIDE * x] & 0xffu) << (8 * x);
        c1 |= (cnt[COOP_XCD_STRIDE * (4 + x)] & 0xffu) << (8 * x);
    }
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    rec[0] = group; rec[1] = member; rec[2] = tile; rec[3] = k; rec[4] = wave; rec[5] = tag;
    rec[6] = h_miss; rec[7] = p_miss; rec[8] = l_miss; rec[9] = c0; rec[10] = c1;
    rec[11] = (unsigned)t; rec[12] = (unsigned)(t >> 32); rec[13] = gpx; rec[14] = blockIdx.x; rec[15] = err[0];
}
}  // namespace

// EVH: lanes per (row, candidate) in the partial dots: 2 (n_per <= 8: 16 floats of the slice each) or 1 (n_per <= 16)
// DIAG: the diagnostic build with phase stamps (tools/stamp_decode.py; lstm_ablate bit 11) — timing only, never a measured run
template <bool SPLIT, int OCC, int EVH, bool DIAG = false>
__global__ __launch_bounds__(256, OCC) void pointer_decode_lean_kernel(DecodeArgs a, u64* xh, u64* xp,
                                                                     u64* xl, unsigned* __restrict__ err,
                                                                     unsigned* __restrict__ sticky, int n_nets, int groups_per_net,
                                                                     int gpx, int write_through, unsigned* __restrict__ seats,
                                                                     unsigned* __restrict__ diag) {
    __shared__ __attribute__((aligned(16))) float hs[SPLIT ? 3 * SPLIT_TILE / 2 : ROWS * LDT];   // fp32 tile (k-quarter-major) | three fp16 piece tiles
    __shared__ __attribute__((aligned(16))) unsigned wts[SPLIT ? SPLIT_WT_DWORDS : 4];
    __shared__ __attribute__((aligned(16))) float hsl[ROWS][UNITS + 4];
    __shared__ __attribute__((aligned(16))) float win[ROWS][KW][8];       // raw rows of the window the NEXT pick comes from
    __shared__ int sel[ROWS];
    __shared__ int abort_flag;
    __shared__ int place[2];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = lane >> 4, c = lane & 15;
    int group, member;
    if (!coop_place<G>(err, gpx, place, group, member, seats, (write_through & 2) != 0, sticky)) return;   // surplus workgroup of the over-subscribed launch (bit 1: opts.paired_start)
    // seat numbers come out of LDS, i.e. in vector registers: say that they are uniform — every base address, resource and
    // scalar offset below then lives in SGPRs (a resource the compiler cannot prove uniform costs a waterfall loop per access)
    group = __builtin_amdgcn_readfirstlane(group);
    member = __builtin_amdgcn_readfirstlane(member);
    const int net_id = group / groups_per_net, gi = group % groups_per_net;
    if (net_id >= n_nets) return;
    const DecodeNet& net = a.net[net_id];
    if (tid == 0) abort_flag = 0;
    __syncthreads();
    const bool same_xcd = !(write_through & 1);
    if (tid == 0 && same_xcd) atomicAdd(err + COOP_PLACED_OFFSET / 4 + COOP_XCD_STRIDE * xcc_id(), 1u);   // statistics: workgroups on the same-XCD fast path (per XCD: its own line)

    const int B = a.B, T = a.T, K = a.K, L = T * K;
    const bool latent_in_launch = net.latent_from >= 0;
    const bool has_lat = latent_in_launch || net.latent_win;
    bool publishes_latent = false;
    for (int n = 0; n < n_nets; ++n) publishes_latent |= (a.net[n].latent_from == net_id);
    const bool first = member == 0;                      // the member that writes the outputs

    // ---- weights of this lane's two gate columns: tile 0 = [i | f], tile 1 = [g | o], 8 units per wave
    const int unit = member * UNITS + wave * 8 + (c & 7);
    const int wrow0 = (0 + (c >> 3)) * H + unit, wrow1 = (2 + (c >> 3)) * H + unit;
    float wBh[SPLIT ? 1 : 2][SPLIT ? 1 : 64];
    f16x8 wH16[SPLIT ? 2 : 1][8], wL16[SPLIT ? 2 : 1][8];
    float winv[2] = {1.0f, 1.0f};
    unsigned* wt_lane = wts + (SPLIT ? (wave * 8 * 64 + lane) * 4 : 0);
    float bh[2], bi[2], sg[2], wXf[2][2];
    const uint4* __restrict__ Wsplit = SPLIT ? static_cast<const uint4*>(net.whh_split) : nullptr;   // uniform: packed once per model, or split here
    if constexpr (SPLIT) {
        if (Wsplit) load_split_weights(Wsplit + (size_t)member * SPLIT_PACK_U4_PER_MEMBER, tid, wH16, wL16, wt_lane, winv);
    }
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        const int wrow = tl ? wrow1 : wrow0, gate = wrow / H, u = wrow % H;
        bh[tl] = net.bhh[wrow];
        if constexpr (SPLIT) {
            if (!Wsplit) winv[tl] = split_weights<H>(net.whh, gate, u, kq, wH16[tl], wL16[tl], wt_lane + 2 * tl);
        } else {
#pragma unroll
            for (int kk = 0; kk < 64; ++kk) wBh[tl][kk] = net.whh[((size_t)(kk * 4 + gate) * H + u) * 4 + kq];
        }
        wXf[tl][0] = net.xw_fold[wrow * 8 + kq];        // B-fragments of (W_ih W_e) [4H,8], its bias, and the step-0 gates
        wXf[tl][1] = net.xw_fold[wrow * 8 + 4 + kq];
        bi[tl] = net.xb_fold[wrow];
        sg[tl] = net.start_fold[wrow];
    }

    // ---- per-lane constants of the step loop (everything else of an address is uniform: resource + scalar offset)
    const int own0 = kq * 4 + (c < 8 ? 0 : 2);           // this lane finishes rows own0, own0+1 of unit `unit`
    const int vo_hpub = (int)(own0 * H + unit) * 8u;
    const int vo_sweep = 16 * lane;                // h pairs: + 1024 j
    // Lanes that have nothing to load load anyway, from an offset past every resource (reads as zero, moves no data): a
    // load under a lane mask is a branch around it, and at the join the compiler's vmcnt bookkeeping — in-order counting —
    // can no longer tell how many operations are in flight: it then waits for the loads of the step's window right where
    // they were issued (seen in the ISA: s_waitcnt vmcnt(2)/(1)/(0) directly behind them).
    constexpr int VO_NONE = (int)0xFFFFFF00u;
    const int rq = kq, r = c;                            // argmax view of the lane: row 4 wave + rq, candidate r
    const bool live = r < K;
    const int vo_p = live ? 64 * lane : VO_NONE;   // lane (rq, r): its 8 partials = 4 pairs, 64 contiguous bytes
    const int vo_lat = live ? (int)(rq * K + r) * 8u : VO_NONE;
    const int vo_wl = (int)((wave * 4 + rq) * T * K + r) * 4u;     // win_logits / latent_win [row][T][K]
    const int vo_idx = (int)((wave * 4 + rq) * T) * 4u;            // idx [row][T]
    const int vo_act = (int)(c * T * 8 + kq) * 4u;                 // actions [row c][T][8], features kq and 4 + kq
    const int vo_q = (int)(own0 * T * H + unit) * 4u;              // queries [row][T][H]
    constexpr int EVN = 8 / EVH;                         // float4 loads of the slice per lane
    const int ppair = tid / EVH, phalf = tid - ppair * EVH;
    const int prow = ppair / K, pcand = ppair - prow * K;
    const bool pdot = ppair < ROWS * K;
    const int vo_ev = pdot ? (int)(((size_t)prow * L + pcand) * H + member * UNITS + phalf * (4 * EVN)) * 4u : VO_NONE;
    const int vo_ppub = (int)((prow * KW + pcand) * G + member) * 8u;
    const int wr_row = tid >> 4, wr_cand = tid & 15;     // window-row staging: thread (row, candidate) moves 32 bytes
    const bool wr_live = wr_cand < K;
    const int vo_win = wr_live ? (int)(((size_t)wr_row * L + wr_cand) * 8) * 4u : VO_NONE;

    __amdgpu_buffer_rsrc_t r_h = make_rsrc(xh + (size_t)group * (2 * ROWS * H), 2 * ROWS * H * 8);
    __amdgpu_buffer_rsrc_t r_p = make_rsrc(xp + (size_t)group * (2 * XP_GRANULES), 2 * XP_GRANULES * 8);

    const int n_tiles = (B + ROWS - 1) / ROWS;
    unsigned step = 0;   // publish counter: tag = step+1, parity = step&1
    unsigned tiles_done = 0;   // proof of work (coop_note_finished)
    for (int tile = gi; tile < n_tiles; tile += groups_per_net) {
        const int b0 = tile * ROWS;
        const int rows_here = min(ROWS, B - b0);
        // tile-relative resources: rows >= B lie past num_records and read as zero / are not written
        __amdgpu_buffer_rsrc_t r_enc = make_rsrc(net.enc_out + (size_t)b0 * L * H, (unsigned)((size_t)rows_here * L * H * 4));
        __amdgpu_buffer_rsrc_t r_in = make_rsrc(a.inputs + (size_t)b0 * L * 8, (unsigned)((size_t)rows_here * L * 32));
        __amdgpu_buffer_rsrc_t r_lat = make_rsrc(xl + (size_t)tile * T * ROWS * K, (unsigned)((size_t)T * ROWS * K * 8));
        // outputs (written by member 0) and the optional stored latent logits, tile-relative as well: a store past
        // num_records (a row >= B) is dropped by the hardware
        __amdgpu_buffer_rsrc_t r_wl = make_rsrc(net.win_logits + (size_t)b0 * T * K, (unsigned)((size_t)rows_here * T * K * 4));
        __amdgpu_buffer_rsrc_t r_idx = make_rsrc(net.idx + (size_t)b0 * T, (unsigned)((size_t)rows_here * T * 4));
        __amdgpu_buffer_rsrc_t r_act = make_rsrc(net.actions + (size_t)b0 * T * 8, (unsigned)((size_t)rows_here * T * 32));
        __amdgpu_buffer_rsrc_t r_lw = make_rsrc(net.latent_win ? net.latent_win + (size_t)b0 * T * K : nullptr,
                                                net.latent_win ? (unsigned)((size_t)rows_here * T * K * 4) : 0u);
        __amdgpu_buffer_rsrc_t r_q = make_rsrc(net.queries ? net.queries + (size_t)b0 * T * H : nullptr,
                                               net.queries ? (unsigned)((size_t)rows_here * T * H * 4) : 0u);
        f32x2 cst, hl = {0.f, 0.f};
        cst.x = b0 + own0 < B ? net.c0[(int64_t)(b0 + own0) * H + unit] : 0.0f;
        cst.y = b0 + own0 + 1 < B ? net.c0[(int64_t)(b0 + own0 + 1) * H + unit] : 0.0f;
        __syncthreads();   // previous tile is completely done with the LDS arrays
        for (int j = 0; j < ROWS; ++j) {
            const float h0v = (b0 + j < B) ? net.h0[(int64_t)(b0 + j) * H + tid] : 0.0f;
            if constexpr (SPLIT) split_store(reinterpret_cast<_Float16*>(hs) + j * LDH16 + tid, h0v);
            else hs[ht_index(j, tid)] = h0v;
        }
        __syncthreads();

        for (int k = 0; k <= T; ++k) {
            float axf0 = 0.0f, axf1 = 0.0f;
            u64 st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if constexpr (DIAG) st[0] = st[1] = st[2] = st[3] = st[4] = st[5] = phase_stamp();
            const unsigned tag = step;                           // publish #(step-1) carries tag `step`
            const unsigned par = (step - 1) & 1;
            const unsigned so_k = (unsigned)k * (unsigned)K;
            // Destinations of the loads requested half-way through the products.  They are "defined" without an instruction and
            // not touched until their real use: a register copy or an initialising move next to the loads makes the compiler
            // wait for them on the spot (vmcnt), in the middle of the chain.  Lanes that do not load never look at them.
            u32x4 ev[EVN];                                       // raw bits of the window's enc_out slice
            u32x4 vp[4];                                         // the lane's 8 partials {value, tag} x 2
            u32x2 vl;                                            // Low's window logit {value, tag}
#pragma unroll
            for (int j = 0; j < EVN; ++j) asm volatile("" : "=v"(ev[j]));
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "=v"(vp[j]));
            asm volatile("" : "=v"(vl));
            f32x4 ah[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

            // This step's window of enc_out (rows of the window against the own 32-unit slice; consumed in (5)).  It does not
            // depend on the picks, only on k; it comes from HBM (enc_out of a launch is tens of MB).  vmcnt counts in order, so a
            // request must sit behind every load whose result is needed EARLIER: the one-per-CU builds request it right behind
            // the sweep of h — it then has the products, the argmax and the cell update to arrive (32 more live registers at
            // most) — and the two-per-CU builds, which have no registers for that, in (4), where the other resident workgroup
            // covers the round trip.
            constexpr bool EV_EARLY = OCC == 1;
            auto request_ev = [&]() {
#pragma unroll
                for (int j = 0; j < EVN; ++j) ev[j] = __builtin_amdgcn_raw_buffer_load_b128(r_enc, vo_ev, so_k * (H * 4) + 16u * j, 0);
            };
            // ---- (1) h_{k-1}: peers published it BEFORE their partial dots, so it is (nearly) there when this member has
            // published its own; the W_hh.h products start on it while the partial dots are still in flight
            if (k < T) {
                if (k > 0) {
                    const unsigned so_h0 = par * (ROWS * H * 8) + wave * (4 * H * 8);
                    u32x4 vh[8];
                    bool ok = false;
                    unsigned h_miss = 0;
                    for (unsigned spins = 0; spins <= SPIN_LIMIT; ++spins) {
                        // The scalar offset passes through an empty asm every pass: the optimiser must not treat a sweep load
                        // as loop-invariant (the first build's ISA had the h loads hoisted ABOVE the polling loop — a pass that
                        // came too early was then repeated over stale registers until the spin bound).
                        unsigned so_h = so_h0;
                        asm volatile("" : "+s"(so_h));
#pragma unroll
                        for (int j = 0; j < 8; ++j) vh[j] = __builtin_amdgcn_raw_buffer_load_b128(r_h, vo_sweep, so_h + 1024u * j, AUX_SC1);   // the constant part rides the SCALAR offset: no per-load address register
                        bool good = true;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            if constexpr (SPLIT) good &= split_pair_tagged(vh[j], tag);
                            else good &= (vh[j].y == tag) & (vh[j].w == tag);
                        }
                        if (__all(good)) {
                            ok = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if constexpr (DIAG) st[1] = phase_stamp();
                    if (!ok) {   // failure path only: whose h is missing (pair j of lane l: member 4 (j & 1) + l / 16)
                        abort_flag = 1;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            bool bad;
                            if constexpr (SPLIT) bad = !split_pair_tagged(vh[j], tag);
                            else bad = (vh[j].y != tag) | (vh[j].w != tag);
                            const unsigned long long bm = __ballot(bad);
                            for (int q = 0; q < 4; ++q)
                                if ((bm >> (16 * q)) & 0xffffull) h_miss |= 1u << (4 * (j & 1) + q);
                        }
                        record_failure(diag, err, group, member, tile, k, wave, tag, h_miss, 0u, 0u, gpx);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int i = 2 * (j * 64 + lane);          // even: units i and i + 1 of one row
                        if constexpr (SPLIT) {
                            split_pair_to_lds(reinterpret_cast<_Float16*>(hs) + (wave * 4 + (i >> 8)) * LDH16 + (i & 255), vh[j]);
                        } else {
                            float* d = &hs[ht_index(wave * 4 + (i >> 8), i & 255)];   // 64 floats apart (k-quarter-major)
                            d[0] = __uint_as_float(vh[j].x);
                            d[64] = __uint_as_float(vh[j].z);
                        }
                    }
                    if constexpr (EV_EARLY) request_ev();
                    __syncthreads();   // the h tile of every wave is complete
                    if (abort_flag) break;
                } else if constexpr (EV_EARLY) {
                    request_ev();
                }
                if constexpr (DIAG) st[2] = phase_stamp();
                // ---- (2) W_hh.h (the folded input side follows the pick, below).  Half-way through, the first pass over the
                // partial dots of publish #(step-1) is requested: they were published a hand-off time ago, and the loads' own
                // round trip (~700 cycles) then lies under the second half of the products instead of behind them.
                constexpr bool EARLY_P = OCC == 1 || !SPLIT;   // the two-per-CU exact-split builds have no 18 registers to spare across the second half
                auto request_p = [&]() {
                    if (EARLY_P && k > 0) {
                        const unsigned so_p = par * (XP_GRANULES * 8) + wave * (4 * KW * G * 8);
#pragma unroll
                        for (int j = 0; j < 4; ++j) vp[j] = __builtin_amdgcn_raw_buffer_load_b128(r_p, vo_p, so_p + 16u * j, AUX_SC1);
                        if (latent_in_launch)
                            vl = __builtin_amdgcn_raw_buffer_load_b64(r_lat, vo_lat, (unsigned)(((k - 1) * ROWS + wave * 4) * K) * 8u, AUX_SC1);
                    }
                };
                if constexpr (SPLIT) split_chain(reinterpret_cast<const _Float16*>(hs) + c * LDH16 + 8 * kq, wH16, wL16, wt_lane, winv, ah, request_p);
                else mfma_chain_pair<LDT, OCC == 2 ? 8 : 16, true>(hs, c, kq, wBh[0], wBh[SPLIT ? 0 : 1], ah[0], ah[1], request_p);
                if constexpr (DIAG) {
                    asm volatile("" ::"v"(ah[0][0]), "v"(ah[1][0]));
                    st[3] = phase_stamp();
                }
            }

            // ---- (3) the partial dots of publish #(step-1) (+ Low's window logits): they crossed while (2) ran
            if (k > 0) {
                const unsigned so_p0 = par * (XP_GRANULES * 8) + wave * (4 * KW * G * 8);
                const unsigned so_l0 = (unsigned)(((k - 1) * ROWS + wave * 4) * K) * 8u;
                bool ok = false;
                for (unsigned spins = 0; spins <= SPIN_LIMIT; ++spins) {
                    // pass 0 looks at what (2) requested half-way through the products (k == T: nothing was requested);
                    // every further pass loads again
                    if (spins > 0 || k == T || (OCC != 1 && SPLIT)) {
                        unsigned so_p = so_p0, so_l = so_l0;
                        asm volatile("" : "+s"(so_p), "+s"(so_l));
#pragma unroll
                        for (int j = 0; j < 4; ++j) vp[j] = __builtin_amdgcn_raw_buffer_load_b128(r_p, vo_p, so_p + 16u * j, AUX_SC1);
                        if (latent_in_launch) vl = __builtin_amdgcn_raw_buffer_load_b64(r_lat, vo_lat, so_l, AUX_SC1);
                    }
                    bool good = true;
                    if (live) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) good &= (vp[j].y == tag) & (vp[j].w == tag);
                        if (latent_in_launch) good &= vl.y == 1u;
                    }
                    if (__all(good)) {
                        ok = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if constexpr (DIAG) st[4] = phase_stamp();
                if (!ok) {
                    abort_flag = 1;
                    unsigned p_miss = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (__ballot(live && vp[j].y != tag)) p_miss |= 1u << (2 * j);
                        if (__ballot(live && vp[j].w != tag)) p_miss |= 1u << (2 * j + 1);
                    }
                    record_failure(diag, err, group, member, tile, k, wave, tag, 0u, p_miss, (unsigned)__popcll(__ballot(live && latent_in_launch && vl.y != 1u)), gpx);
                }

                // ---- logits and first-max argmax: lane (rq, r) = (row 4 wave + rq, candidate r)
                {
                    const int row = wave * 4 + rq;
                    const unsigned so_t = (unsigned)(k - 1) * (unsigned)K * 4u;
                    float dot = 0.0f, lv = 0.0f;
                    if (live) {                                  // the 8 partials in member order
                        dot = __fadd_rn(__fadd_rn(__fadd_rn(__uint_as_float(vp[0].x), __uint_as_float(vp[0].z)), __uint_as_float(vp[1].x)), __uint_as_float(vp[1].z));
                        dot = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(dot, __uint_as_float(vp[2].x)), __uint_as_float(vp[2].z)), __uint_as_float(vp[3].x)), __uint_as_float(vp[3].z));
                        if (latent_in_launch) lv = __uint_as_float(vl.x);
                        else if (net.latent_win) lv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_lw, vo_wl, so_t, 0));
                    }
                    // C*tanh with the device-library tanhf: these values decide the pick
                    float v = a.use_tanh ? __fmul_rn(a.tanh_c, tanhf(dot)) : dot;
                    if (live && first) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r_wl, vo_wl, so_t, 0);
                        if (publishes_latent)   // read by the High net's group, wherever it sits: agent scope
                            __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v), 1u}, r_lat, vo_lat, so_l0, AUX_SC1);
                    }
                    v = (has_lat && live) ? __fadd_rn(v, lv) : v;
                    float m = live ? v : -INFINITY;
                    m = fmaxf(m, dpp_ror(m, 8));
                    m = fmaxf(m, dpp_ror(m, 4));
                    m = fmaxf(m, dpp_ror(m, 2));
                    m = fmaxf(m, dpp_ror(m, 1));
                    // torch.max returns the FIRST maximum: the lowest candidate whose logit equals the row maximum
                    const unsigned hit = (unsigned)(__ballot(live && v == m) >> (16 * rq)) & 0xffffu;
                    const int pick_r = hit ? __ffs(hit) - 1 : 0;
                    if (r == 0) {
                        sel[row] = pick_r;
                        if (first) __builtin_amdgcn_raw_buffer_store_b32((unsigned)((k - 1) * K + pick_r), r_idx, vo_idx, (unsigned)(k - 1) * 4u, 0);
                    }
                }
                __syncthreads();   // sel[] is complete (win[] was written a step ago)
                if (abort_flag) break;
                // raw 8-feature row of the pick as MFMA A-fragments (row c, k = 4 kk2 + kq) out of the staged window
                axf0 = win[c][sel[c]][kq];
                axf1 = win[c][sel[c]][4 + kq];
                if (first && wave == 0) {                        // ... the same registers ARE the action row
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(axf0), r_act, vo_act, (unsigned)(k - 1) * 32u, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(axf1), r_act, vo_act, (unsigned)(k - 1) * 32u + 16u, 0);
                }
                if (k == T) break;
                if constexpr (DIAG) st[5] = phase_stamp();
            }

            // ---- (4) folded input side, cell update, publish h_k
            if constexpr (!EV_EARLY) request_ev();
            // the raw rows of this step's window (the next pick's decoder input comes out of them): requested here — they are
            // consumed at the end of the step — so that they are not live across the products
            u32x4 wv0, wv1;
            wv0 = __builtin_amdgcn_raw_buffer_load_b128(r_in, vo_win, so_k * 32u, 0);
            wv1 = __builtin_amdgcn_raw_buffer_load_b128(r_in, vo_win, so_k * 32u + 16u, 0);
            float gx[2][4];
            if (k > 0) {
                f32x4 ax0 = {0.f, 0.f, 0.f, 0.f}, ax1 = ax0;
                ax0 = __builtin_amdgcn_mfma_f32_16x16x4f32(axf0, wXf[0][0], ax0, 0, 0, 0);
                ax1 = __builtin_amdgcn_mfma_f32_16x16x4f32(axf0, wXf[1][0], ax1, 0, 0, 0);
                ax0 = __builtin_amdgcn_mfma_f32_16x16x4f32(axf1, wXf[0][1], ax0, 0, 0, 0);
                ax1 = __builtin_amdgcn_mfma_f32_16x16x4f32(axf1, wXf[1][1], ax1, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    gx[0][q] = __fadd_rn(ax0[q], bi[0]);
                    gx[1][q] = __fadd_rn(ax1[q], bi[1]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    gx[0][q] = sg[0];
                    gx[1][q] = sg[1];
                }
            }
            {   // one cell update per lane for the two rows it finishes (coop_common.h: cell_update_split)
                f32x2 g0[2], g1[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    g0[q] = (f32x2{ah[0][2 * q], ah[0][2 * q + 1]} + pk_set(bh[0])) + f32x2{gx[0][2 * q], gx[0][2 * q + 1]};
                    g1[q] = (f32x2{ah[1][2 * q], ah[1][2 * q + 1]} + pk_set(bh[1])) + f32x2{gx[1][2 * q], gx[1][2 * q + 1]};
                }
                cell_update_split(g0, g1, c < 8, cst, hl);
            }
            {
                const unsigned so_pub = (step & 1) * (ROWS * H * 8);
                if constexpr (SPLIT) {
                    store_granule(r_h, vo_hpub, so_pub, split_granule(step + 1, hl.x), same_xcd);
                    store_granule(r_h, vo_hpub, so_pub + H * 8, split_granule(step + 1, hl.y), same_xcd);
                } else {
                    store_granule(r_h, vo_hpub, so_pub, ((u64)(step + 1) << 32) | __float_as_uint(hl.x), same_xcd);
                    store_granule(r_h, vo_hpub, so_pub + H * 8, ((u64)(step + 1) << 32) | __float_as_uint(hl.y), same_xcd);
                }
                hsl[own0][wave * 8 + (c & 7)] = hl.x;
                hsl[own0 + 1][wave * 8 + (c & 7)] = hl.y;
                if (net.queries) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hl.x), r_q, vo_q, (unsigned)k * (H * 4), 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hl.y), r_q, vo_q, (unsigned)(k + T) * (H * 4), 0);
                }
            }
            if constexpr (DIAG) {
                asm volatile("" ::"v"(hl.y));
                st[6] = phase_stamp();
            }
            __syncthreads();   // hsl is complete; every wave has read its picks' rows out of win[]
            // ---- (5) partial attention dots of the step-k window against the own h_k slice
            {
                float p = 0.0f;
                if (pdot) {
#pragma unroll
                    for (int j = 0; j < EVN; ++j) {
                        const float4 hv = *reinterpret_cast<const float4*>(&hsl[prow][phalf * (4 * EVN) + 4 * j]);
                        p = fmaf(__uint_as_float(ev[j].x), hv.x, p);
                        p = fmaf(__uint_as_float(ev[j].y), hv.y, p);
                        p = fmaf(__uint_as_float(ev[j].z), hv.z, p);
                        p = fmaf(__uint_as_float(ev[j].w), hv.w, p);
                    }
                }
                if constexpr (EVH == 2)   // lanes 2p, 2p+1 hold the two halves of the slice: low half + high half
                    p = __fadd_rn(p, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false)));
                if (pdot && phalf == 0)
                    store_granule(r_p, vo_ppub, (step & 1) * (XP_GRANULES * 8), ((u64)(step + 1) << 32) | __float_as_uint(p), same_xcd);
            }
            if (wr_live) {
                *reinterpret_cast<u32x4*>(&win[wr_row][wr_cand][0]) = wv0;
                *reinterpret_cast<u32x4*>(&win[wr_row][wr_cand][4]) = wv1;
            }
            if constexpr (DIAG) {
                st[7] = phase_stamp();
                if (blockIdx.x == 0 && tid == 0 && k > 0) {   // sums live in the status area (u64 words 4..)
                    u64* prof = reinterpret_cast<u64*>(err) + 4;
                    prof[0] += st[1] - st[0];   // sweep of h: wait for the tags
                    prof[1] += st[2] - st[1];   // LDS fill of h + barrier
                    prof[2] += st[3] - st[2];   // window loads issued + W_hh.h products
                    prof[3] += st[4] - st[3];   // sweep of the partial dots: wait
                    prof[4] += st[5] - st[4];   // logits, argmax, barrier, pick's row
                    prof[5] += st[6] - st[5];   // input side, cell, publish h
                    prof[6] += st[7] - st[6];   // barrier, partial dots + publish, window rows to LDS
                    prof[7] += 1;
                }
            }
            ++step;
        }
        if (abort_flag) break;
        ++tiles_done;
    }
    if (tid == 0) {
        if (abort_flag) coop_raise(err, sticky, 2u, seats);
        coop_note_finished(sticky, GNNPN_STATUS_DEC_FINISHED, tiles_done);
    }
}

// pick_prob[b][t] = softmax of the step's window logits (+ the latent logits) at the pick = 1 / sum_j exp(v_j - v_pick):
// the normaliser the decode kernel no longer forms per step (modelPN.py:224-226, 297-299 gather exactly this).  16 lanes
// per (problem, step), the exponentials summed by the DPP rotations 8, 4, 2, 1 — the order decode_coop.hip uses in lane 0.
struct PickProbArgs {
    const float* win[2];
    const float* lat[2];
    const int32_t* idx[2];
    float* out[2];
};
// blockIdx.y = net: both nets' normalisers in ONE launch.  One LANE per (problem, step): its <= 16 logits in registers, the sum
// formed by the same tree the 16-lane DPP form walks (rotations 8, 4, 2, 1 seen from lane 0: p_i = e_i + e_{i+8}, q_i = p_i + p_{i+4},
// r_i = q_i + q_{i+2}, r_0 + r_1; absent entries are exact zeros) — the same bits with a sixteenth of the waves.
__global__ __launch_bounds__(256) void pick_prob_kernel(PickProbArgs a, int64_t rows, int T, int K) {
    const int net = blockIdx.y;
    const float* __restrict__ win = a.win[net];
    const float* __restrict__ lat = a.lat[net];
    const int32_t* __restrict__ idx = a.idx[net];
    float* __restrict__ out = a.out[net];
    const int64_t row = blockIdx.x * 256ll + threadIdx.x;
    if (row >= rows) return;
    // the step of this row: one 64-bit remainder per workgroup (scalar), a 32-bit one per lane
    const unsigned t_base = (unsigned)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256ll) % T));
    const int t = (int)((t_base + threadIdx.x) % (unsigned)T);
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        v[j] = 0.0f;
        if (j < K) {
            v[j] = win[row * K + j];
            if (lat) v[j] = __fadd_rn(v[j], lat[row * K + j]);
        }
    }
    const int pick_r = min(max(idx[row] - t * K, 0), 15);
    float best = v[0];
#pragma unroll
    for (int j = 1; j < 16; ++j) best = j == pick_r ? v[j] : best;
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = j < K ? expf(__fsub_rn(v[j], best)) : 0.0f;
#pragma unroll
    for (int n = 8; n >= 1; n >>= 1)
#pragma unroll
        for (int j = 0; j < n; ++j) v[j] = __fadd_rn(v[j], v[j + n]);
    out[row] = 1.0f / v[0];
}

int64_t gnnpn_decode_lean_workspace_bytes(int32_t B, int32_t T, int32_t n_per) {
    const int64_t groups = 64, tiles = (B + ROWS - 1) / ROWS;
    return COOP_STATUS_BYTES + groups * 2 * ROWS * H * 8 + groups * 2 * XP_GRANULES * 8 + tiles * T * ROWS * (int64_t)n_per * 8 +
           COOP_OVERREAD_BYTES;
}

// folded, greedy nets only (the caller checked); shared_cu: the 256-register build (two workgroups per CU)
int gnnpn_launch_decode_lean(const DecodeArgs& args, int n_nets, int precision, bool shared_cu, const CoopOpts& opts,
                             void* workspace, int64_t workspace_bytes, hipStream_t s) {
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        GNNPN_FAIL(GNNPN_E_LAUNCH, "pointer_decode: cannot query the device");
    if (args.K > KW) GNNPN_FAIL(GNNPN_E_UNSUP, "pointer_decode: the cooperative form is built for n_per <= %d", KW);
    const int n_tiles = (args.B + ROWS - 1) / ROWS;
    int gpx = n_cu / (8 * G);
    if (gpx > 8) gpx = 8;
    while (gpx > 1 && (gpx - 1) * 8 >= n_nets * n_tiles) --gpx;
    if (gpx < 1 || n_cu < 256) GNNPN_FAIL(GNNPN_E_UNSUP, "pointer_decode: device has %d CUs, the cooperative form is built for 8 XCDs x 32 CUs", n_cu);
    const int groups = gpx * 8;
    if (groups < n_nets) GNNPN_FAIL(GNNPN_E_UNSUP, "pointer_decode: %d groups for %d nets", groups, n_nets);
    const int groups_per_net = groups / n_nets;
    // the tile-relative 32-bit offsets of the buffer resources
    if ((int64_t)ROWS * args.T * args.K * H * 4 >= (1ll << 32) || (int64_t)args.T * ROWS * args.K * 8 >= (1ll << 32))
        GNNPN_FAIL(GNNPN_E_UNSUP, "pointer_decode: a tile of 16 problems must stay below 4 GB of encoder output");
    const int64_t h_bytes = (int64_t)groups * 2 * ROWS * H * 8;
    const int64_t p_bytes = (int64_t)groups * 2 * XP_GRANULES * 8;
    const int64_t l_bytes = (int64_t)n_tiles * args.T * ROWS * args.K * 8;
    const int64_t need = COOP_STATUS_BYTES + h_bytes + p_bytes + l_bytes + COOP_OVERREAD_BYTES;
    if (!workspace || workspace_bytes < need || !gnnpn_aligned(workspace, 256))
        GNNPN_FAIL(GNNPN_E_ARG, "pointer_decode: workspace of %lld B (256-B aligned) required", (long long)need);
    if (coop_zero_workspace(workspace, (size_t)need, s, opts.sticky, GNNPN_STATUS_DEC_EXPECTED, (unsigned)(G * n_nets * n_tiles),
                            (gnnpn_option_lstm_ablate() & 0x2000) != 0) != hipSuccess)
        GNNPN_FAIL(GNNPN_E_LAUNCH, "pointer_decode: workspace memset failed");
    g_gnnpn_last_units = opts.sticky ? (int64_t)G * n_nets * n_tiles : 0;
    unsigned* p_seats = gnnpn_cu_seat_table();
    if (!p_seats) GNNPN_FAIL(GNNPN_E_LAUNCH, "pointer_decode: cannot allocate the seat table");
    char* base = static_cast<char*>(workspace);
    u64* p_h = reinterpret_cast<u64*>(base + COOP_STATUS_BYTES);
    u64* p_p = reinterpret_cast<u64*>(base + COOP_STATUS_BYTES + h_bytes);
    u64* p_l = reinterpret_cast<u64*>(base + COOP_STATUS_BYTES + h_bytes + p_bytes);
    unsigned* p_err = reinterpret_cast<unsigned*>(base);
    const bool split = precision == GNNPN_PREC_SPLIT;
    static unsigned* p_diag = gnnpn_decode_diag_buffer();   // failure record (written on a timed-out sweep only)
    const int wt = (opts.write_through ? 1 : 0) | (opts.paired_start ? 2 : 0);
    const int lds_kb = opts.lds_kb;
#define GNNPN_LEAN(SPLIT_, OCC_, EVH_, ...)                                                                                  \
    hipLaunchKernelGGL((pointer_decode_lean_kernel<SPLIT_, OCC_, EVH_, ##__VA_ARGS__>), dim3(COOP_OVERSUB * groups * G), dim3(256), \
                       coop_lds_padding((const void*)pointer_decode_lean_kernel<SPLIT_, OCC_, EVH_, ##__VA_ARGS__>, lds_kb), s, args, p_h, p_p, \
                       p_l, p_err, opts.sticky, n_nets, groups_per_net, gpx, wt, p_seats, p_diag)
    const bool wide = args.K > 8;
    if (gnnpn_option_lstm_ablate() & 0x800) {       // phase stamps (1-per-CU builds, n_per <= 8): a decoder-only bit
        if (wide) GNNPN_FAIL(GNNPN_E_UNSUP, "pointer_decode: the stamped build exists for n_per <= 8");
        if (split) GNNPN_LEAN(true, 1, 2, true);
        else GNNPN_LEAN(false, 1, 2, true);
    } else if (split && shared_cu && wide) GNNPN_LEAN(true, 2, 1);
    else if (split && shared_cu) GNNPN_LEAN(true, 2, 2);
    else if (split && wide) GNNPN_LEAN(true, 1, 1);
    else if (split) GNNPN_LEAN(true, 1, 2);
    else if (shared_cu && wide) GNNPN_LEAN(false, 2, 1);
    else if (shared_cu) GNNPN_LEAN(false, 2, 2);
    else if (wide) GNNPN_LEAN(false, 1, 1);
    else GNNPN_LEAN(false, 1, 2);
#undef GNNPN_LEAN
    if (hipGetLastError() != hipSuccess) GNNPN_FAIL(GNNPN_E_LAUNCH, "pointer_decode: launch of the cooperative kernel failed");
    // the softmax normaliser at the picks, from the stored window logits (Low's are the High net's latent logits)
    const int64_t rows = (int64_t)args.B * args.T;
    for (int n0 = 0; n0 < n_nets; n0 += 2) {
        PickProbArgs pa = {};
        const int cnt = n_nets - n0 < 2 ? n_nets - n0 : 2;
        for (int n = 0; n < cnt; ++n) {
            const DecodeNet& d = args.net[n0 + n];
            pa.win[n] = d.win_logits;
            pa.lat[n] = d.latent_from >= 0 ? args.net[d.latent_from].win_logits : d.latent_win;
            pa.idx[n] = d.idx;
            pa.out[n] = d.pick_prob;
        }
        hipLaunchKernelGGL(pick_prob_kernel, dim3((unsigned)((rows + 255) / 256), (unsigned)cnt), dim3(256), 0, s, pa, rows, args.T, args.K);
    }
    return GNNPN_OK;
}
