// Greedy pointer decode, cooperative form for CO-RESIDENCY with the encoder (v3).
//
// Same algorithm and hand-off as decode_coop.hip (read its header), different ownership: a group is
// 16 workgroups, member m owns hidden units [16m, 16m+16); each of the 4 waves owns ONE 16-column MFMA
// tile [i(4) | f(4) | g(4) | o(4)] of 4 units = 64 B-fragment VGPRs (instead of 128), so the kernel fits
// 256 registers per lane and runs at one wave per SIMD BESIDE a wave of the cooperative encoder
// (253 registers) — the decoder of one batch then overlaps the encoder of the next batch that is in
// flight on another stream, which is what its step latency cannot hide on its own.
// Folded input side only (decode_coop.hip handles the literal two-stage order).  The partial
// attention dots are summed over 16 members of 16 units each (member order).
#include "common.h"
#include "recurrent.h"
#include "decode_shared.h"
#include "coop_common.h"
#include "lstm_shared.h"

namespace {
constexpr int H = 256;
constexpr int G = 16;
constexpr int ROWS = 16;
constexpr int UNITS = H / G;      // 16
constexpr int LDH = 258;
constexpr int KMAX = 16;
constexpr unsigned SPIN_LIMIT = 400000;
}  // namespace

__device__ __forceinline__ int ror16i(int v, int n) {
    switch (n) {
        case 1: return __builtin_amdgcn_update_dpp(0, v, 0x121, 0xF, 0xF, false);
        case 2: return __builtin_amdgcn_update_dpp(0, v, 0x122, 0xF, 0xF, false);
        case 4: return __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, false);
        case 8: return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, false);
        default: return __builtin_amdgcn_update_dpp(0, v, 0x12C, 0xF, 0xF, false);
    }
}
__device__ __forceinline__ float ror16v(float v, int n) { return __int_as_float(ror16i(__float_as_int(v), n)); }

// NP = granule loads per lane for the partial dots (= smallest built size >= n_per): sizing the sweep's
// registers by the actual K keeps the kernel inside 256 registers without spills.
template <int NP>
__global__ __launch_bounds__(256, 2) void pointer_decode_coop2_kernel(DecodeArgs a, u64* __restrict__ xh,
                                                                      u64* __restrict__ xp, u64* __restrict__ xl,
                                                                      unsigned* __restrict__ err, unsigned* __restrict__ sticky,
                                                                      int n_nets, int groups_per_net, int gpx, int ablate, unsigned* __restrict__ seats) {
    __shared__ __attribute__((aligned(16))) float hs[ROWS * LDH];     // fp32 h tile (stride LDH)
    __shared__ __attribute__((aligned(16))) float hsl[ROWS][UNITS + 4];
    __shared__ __attribute__((aligned(16))) float part_lin[4][G * 4 * KMAX];   // per wave: the partial dots of its 4 rows, [row%4][cand][member]
    __shared__ float lat[ROWS][KMAX];
    __shared__ int sel[ROWS];
    __shared__ int abort_flag;
    __shared__ int place[2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane >> 4, c = lane & 15, gate = c >> 2;
    int group, member;
    if (!coop_place<G>(err, gpx, place, group, member, seats, false, sticky)) return;   // surplus workgroup of the over-subscribed launch (coop_common.h)
    const int net_id = group / groups_per_net, gi = group % groups_per_net;
    if (net_id >= n_nets) return;
    const DecodeNet& net = a.net[net_id];
    if (tid == 0) abort_flag = 0;
    __syncthreads();
    const bool same_xcd = !(ablate & 128);
    if (tid == 0 && same_xcd) atomicAdd(err + 1, 1u);   // statistics: workgroups on the same-XCD fast path

    const int B = a.B, T = a.T, K = a.K, L = T * K;
    const bool latent_in_launch = net.latent_from >= 0;
    const bool has_lat = latent_in_launch || net.latent_win;
    bool publishes_latent = false;
    for (int n = 0; n < n_nets; ++n) publishes_latent |= (a.net[n].latent_from == net_id);

    u64* xh_g = xh + (size_t)group * (2 * ROWS * H);
    u64* xp_g = xp + (size_t)group * (2 * G * ROWS * K);

    // this lane's gate column: unit = 16m + 4w + (c&3), gate = c>>2
    const int unit = member * UNITS + wave * 4 + (c & 3);
    const int wrow = gate * H + unit;
    float wBh[64];
#pragma unroll
    for (int kk = 0; kk < 64; ++kk) wBh[kk] = net.whh[((size_t)(kk * 4 + gate) * H + unit) * 4 + kq];
    const float bh = net.bhh[wrow], bi = net.xb_fold[wrow], sg = net.start_fold[wrow];
    const float wXf0 = net.xw_fold[wrow * 8 + kq], wXf1 = net.xw_fold[wrow * 8 + 4 + kq];

    const int n_tiles = (B + ROWS - 1) / ROWS;
    unsigned step = 0;   // publish counter: tag = step+1, parity = step&1
    for (int tile = gi; tile < n_tiles; tile += groups_per_net) {
        const int b0 = tile * ROWS;
        float cst[4], hl[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = b0 + kq * 4 + r;
            cst[r] = b < B ? net.c0[(int64_t)b * H + unit] : 0.0f;
            hl[r] = 0.0f;
        }
        __syncthreads();   // previous tile is completely done with the LDS arrays
        for (int j = 0; j < ROWS; ++j) {
            const float h0v = (b0 + j < B) ? net.h0[(int64_t)(b0 + j) * H + tid] : 0.0f;
            hs[j * LDH + tid] = h0v;
        }
        __syncthreads();

        for (int k = 0; k <= T; ++k) {
            float axf0 = 0.0f, axf1 = 0.0f;
            if (k > 0) {
                // ---- ONE combined sweep of publish #(step-1): h_{k-1}, partial dots, Low's window logits
                const unsigned tag = step;
                const int par = (step - 1) & 1;
                {
                    const u64* src_h = xh_g + par * (ROWS * H) + wave * 4 * H;
                    const u64* src_p = xp_g + par * (G * ROWS * K) + wave * (G * 4 * K);   // [quarter][row%4][cand][member]
                    const u64* src_l = xl + ((size_t)tile * T + (k - 1)) * ROWS * K + wave * 4 * K;
                    const int n_p = G * 4 * K;            // this wave's rows 4w..4w+3 from all 16 members
                    const int n_l = 4 * K;
                    unsigned vh[16], vp[NP], vl = 0;
                    bool ok = false;
                    for (unsigned spins = 0; spins <= SPIN_LIMIT; ++spins) {
                        bool good = true;
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const u64 x = granule_load(src_h + j * 64 + lane);
                            vh[j] = (unsigned)x;
                            good &= (unsigned)(x >> 32) == tag;
                        }
#pragma unroll
                        for (int j = 0; j < NP; ++j) {
                            if (lane + 64 * j < n_p) {
                                const u64 x = granule_load(src_p + lane + 64 * j);
                                vp[j] = (unsigned)x;
                                good &= (unsigned)(x >> 32) == tag;
                            }
                        }
                        if (latent_in_launch && lane < n_l) {
                            const u64 x = granule_load(src_l + lane);
                            vl = (unsigned)x;
                            good &= (unsigned)(x >> 32) == 1u;
                        }
                        if (__all(good)) {
                            ok = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    if (!ok) abort_flag = 1;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int i = j * 64 + lane;
                        hs[(wave * 4 + (i >> 8)) * LDH + (i & 255)] = __uint_as_float(vh[j]);
                    }
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        if (lane + 64 * j < n_p) part_lin[wave][lane + 64 * j] = __uint_as_float(vp[j]);
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
                __syncthreads();
                if (abort_flag) break;

                // ---- logits, softmax and first-max argmax inside the wave (16 lanes per row, lane = candidate)
                {
                    const int row = wave * 4 + kq, b = b0 + row, r = c;
                    const bool live = r < K;
                    float dot = 0.0f;
                    if (live) {
                        // the G members' partial dots of (row, candidate) are contiguous: four 16-byte LDS reads
                        const float4* pp = reinterpret_cast<const float4*>(&part_lin[wave][(kq * K + r) * G]);
                        const float4 p0 = pp[0], p1 = pp[1], p2 = pp[2], p3 = pp[3];
                        dot = __fadd_rn(__fadd_rn(__fadd_rn(p0.x, p0.y), p0.z), p0.w);                      // member order
                        dot = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(dot, p1.x), p1.y), p1.z), p1.w);
                        dot = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(dot, p2.x), p2.y), p2.z), p2.w);
                        dot = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(dot, p3.x), p3.y), p3.z), p3.w);
                    }
                    float v = a.use_tanh ? __fmul_rn(a.tanh_c, tanhf(dot)) : dot;   // device-library tanhf: decides the pick
                    if (live && member == 0) {
                        if (b < B) net.win_logits[((int64_t)b * T + (k - 1)) * K + r] = v;
                        if (publishes_latent)
                            granule_store(xl + (((size_t)tile * T + (k - 1)) * ROWS + row) * K + r, 1u, v);
                    }
                    if (has_lat && live) v = __fadd_rn(v, lat[row][r]);
                    const unsigned hi = live ? float_order_key(v) : 0u;
                    unsigned long long key = ((unsigned long long)hi << 32) | (unsigned)(15 - r);
#pragma unroll
                    for (int n = 1; n <= 8; n <<= 1) {
                        const unsigned olo = (unsigned)ror16i((int)(unsigned)key, n);
                        const unsigned ohi = (unsigned)ror16i((int)(unsigned)(key >> 32), n);
                        const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
                        key = o > key ? o : key;
                    }
                    const int best_r = 15 - (int)(key & 0xffffffffu);
                    float best = live ? v : -INFINITY;
#pragma unroll
                    for (int n = 1; n <= 8; n <<= 1) best = fmaxf(best, ror16v(best, n));
                    float e = live ? expf(__fsub_rn(v, best)) : 0.0f;
#pragma unroll
                    for (int n = 8; n >= 1; n >>= 1) e = __fadd_rn(e, ror16v(e, n));
                    if (r == 0) {
                        sel[row] = (k - 1) * K + best_r;
                        if (member == 0 && b < B) {
                            net.pick_prob[(int64_t)b * T + (k - 1)] = 1.0f / e;
                            net.idx[(int64_t)b * T + (k - 1)] = (k - 1) * K + best_r;
                        }
                    }
                }
                __syncthreads();
                {   // raw 8-feature row of the pick as MFMA A-fragments (row c), in flight under the MFMAs; loaded
                    // unconditionally (row clamped) so that no join waits for it, and stored as the action row by
                    // wave 0 of member 0 once it has arrived (see decode_coop.hip)
                    const int bc = min(b0 + c, B - 1);
                    const float* rowp = a.inputs + ((int64_t)bc * L + sel[c]) * 8;
                    axf0 = rowp[kq];
                    axf1 = rowp[4 + kq];
                }
                if (k == T) {
                    if (member == 0 && wave == 0 && b0 + c < B) {
                        float* act = net.actions + ((int64_t)(b0 + c) * T + (k - 1)) * 8;
                        act[kq] = axf0;
                        act[4 + kq] = axf1;
                    }
                    break;
                }
            }

            // this step's window rows, own 16-unit slice: thread (row, cand) holds 16 floats
            float4 ev[4];
            const int prow = tid / K, pcand = tid - prow * K;
            const bool pdot = tid < ROWS * K;
            if (pdot) {
                const int b = b0 + prow;
                const float* src = net.enc_out + ((int64_t)b * L + (int64_t)k * K + pcand) * H + member * UNITS;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    ev[j] = b < B ? *reinterpret_cast<const float4*>(src + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
            }

            // ---- decoder LSTM cell: W_hh.h as one k-ordered fma chain per gate column, folded input side
            f32x4 ah = {0.f, 0.f, 0.f, 0.f};
            {
                const float* base = hs + c * LDH + kq;
                float av[2][8];   // A-fragments fetched 8 k-steps ahead (one chain: 8 MFMAs = 256 cycles of cover)
#pragma unroll
                for (int i = 0; i < 8; ++i) av[0][i] = base[4 * i];
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) {
                    if (ch < 7) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) av[(ch + 1) & 1][i] = base[4 * (8 * (ch + 1) + i)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        ah = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ch & 1][i], wBh[8 * ch + i], ah, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            float gx[4];
            if (k > 0) {
                f32x4 ax = {0.f, 0.f, 0.f, 0.f};
                ax = __builtin_amdgcn_mfma_f32_16x16x4f32(axf0, wXf0, ax, 0, 0, 0);
                ax = __builtin_amdgcn_mfma_f32_16x16x4f32(axf1, wXf1, ax, 0, 0, 0);
                if (member == 0 && wave == 0 && b0 + c < B) {   // the action row of pick k-1
                    float* act = net.actions + ((int64_t)(b0 + c) * T + (k - 1)) * 8;
                    act[kq] = axf0;
                    act[4 + kq] = axf1;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) gx[r] = __fadd_rn(ax[r], bi);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) gx[r] = sg;
            }
            u64* out_h = xh_g + (step & 1) * (ROWS * H);
#pragma unroll
            for (int r = 0; r < 4; r += 2) {   // two rows per pass: packed fp32 arithmetic, no stores in between
                const f32x2 gpre = (f32x2{ah[r], ah[r + 1]} + pk_set(bh)) + f32x2{gx[r], gx[r + 1]};
                const f32x2 a0 = cell_act2(gpre, gate == 2);
                const f32x2 fg = {ror16v(a0.x, 12), ror16v(a0.y, 12)}, gg = {ror16v(a0.x, 8), ror16v(a0.y, 8)},
                            og = {ror16v(a0.x, 4), ror16v(a0.y, 4)};   // valid in the lanes of gate 0 (ig = a0)
                const f32x2 cs = fg * f32x2{cst[r], cst[r + 1]} + a0 * gg;
                const f32x2 hh = og * cell_act2(cs, true);
                cst[r] = cs.x;
                cst[r + 1] = cs.y;
                hl[r] = hh.x;
                hl[r + 1] = hh.y;
            }
            if (gate == 0) {
                u64* dst = out_h + (kq * 4) * H + unit;
                if (same_xcd) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) granule_store_l2(dst + r * H, step + 1, hl[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) granule_store(dst + r * H, step + 1, hl[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = kq * 4 + r;
                    hsl[row][wave * 4 + (c & 3)] = hl[r];
                    if (net.queries && b0 + row < B) net.queries[((int64_t)(b0 + row) * T + k) * H + unit] = hl[r];
                }
            }
            __syncthreads();
            // ---- partial attention dots of the step-k window against the own h_k slice (16 units)
            if (pdot) {
                float p = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 hv = *reinterpret_cast<const float4*>(&hsl[prow][4 * j]);
                    p = fmaf(ev[j].x, hv.x, p);
                    p = fmaf(ev[j].y, hv.y, p);
                    p = fmaf(ev[j].z, hv.z, p);
                    p = fmaf(ev[j].w, hv.w, p);
                }
                granule_publish(xp_g + (step & 1) * (G * ROWS * K) + ((size_t)prow * K + pcand) * G + member, step + 1, p, same_xcd);
            }
            ++step;
        }
        if (abort_flag) break;
    }
    if (abort_flag && tid == 0) coop_raise(err, sticky, 2u, seats);
}

int64_t gnnpn_decode_coop2_workspace_bytes(int32_t B, int32_t T, int32_t n_per) {
    const int64_t groups = 16, tiles = (B + ROWS - 1) / ROWS;
    return COOP_STATUS_BYTES + groups * 2 * ROWS * H * 8 + groups * 2 * G * ROWS * (int64_t)n_per * 8 +
           tiles * T * ROWS * (int64_t)n_per * 8;
}

// GNNPN_E_UNSUP (error text untouched) when this form does not fit the call: the caller then uses
// decode_coop.hip
int gnnpn_launch_decode_coop2(const DecodeArgs& args, int n_nets, int precision, const CoopOpts& opts, void* workspace,
                              int64_t workspace_bytes, hipStream_t s) {
    for (int n = 0; n < n_nets; ++n)
        if (!args.net[n].xw_fold) return GNNPN_E_UNSUP;
    if (args.K > KMAX) return GNNPN_E_UNSUP;
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return GNNPN_E_UNSUP;
    const int n_tiles = (args.B + ROWS - 1) / ROWS;
    int gpx = n_cu / (8 * G);
    if (gpx > 2) gpx = 2;
    while (gpx > 1 && (gpx - 1) * 8 >= n_nets * n_tiles) --gpx;
    if (gpx < 1 || n_cu < 256) return GNNPN_E_UNSUP;   // built for 8 XCDs x 32 CUs (coop_common.h: coop_place)
    const int groups = gpx * 8;
    if (groups < n_nets) return GNNPN_E_UNSUP;
    const int groups_per_net = groups / n_nets;
    const int64_t h_bytes = (int64_t)groups * 2 * ROWS * H * 8;
    const int64_t p_bytes = (int64_t)groups * 2 * G * ROWS * args.K * 8;
    const int64_t l_bytes = (int64_t)n_tiles * args.T * ROWS * args.K * 8;
    const int64_t need = COOP_STATUS_BYTES + h_bytes + p_bytes + l_bytes;
    if (!workspace || workspace_bytes < need || !gnnpn_aligned(workspace, 256)) return GNNPN_E_UNSUP;
    if (coop_zero_workspace(workspace, (size_t)need, s) != hipSuccess) return GNNPN_E_UNSUP;
    unsigned* p_seats = gnnpn_cu_seat_table();
    if (!p_seats) return GNNPN_E_UNSUP;
    char* base = static_cast<char*>(workspace);
    u64* p_h = reinterpret_cast<u64*>(base + COOP_STATUS_BYTES);
    u64* p_p = reinterpret_cast<u64*>(base + COOP_STATUS_BYTES + h_bytes);
    u64* p_l = reinterpret_cast<u64*>(base + COOP_STATUS_BYTES + h_bytes + p_bytes);
    unsigned* p_e = reinterpret_cast<unsigned*>(base);
    const int abl = (gnnpn_option_lstm_ablate() & ~(64 | 0x800)) | (opts.write_through ? 128 : 0);
    unsigned* p_s = opts.sticky;
    if (precision == GNNPN_PREC_SPLIT) return GNNPN_E_UNSUP;   // the exact-split product lives in the 8-member builds (decode_coop.hip); callers fall back to them
    const int lds_kb = opts.lds_kb;
#define GNNPN_DEC2(NP_)                                                                                          \
    hipLaunchKernelGGL((pointer_decode_coop2_kernel<NP_>), dim3(COOP_OVERSUB * groups * G), dim3(256),           \
                       coop_lds_padding((const void*)pointer_decode_coop2_kernel<NP_>, lds_kb), s, args,         \
                       p_h, p_p, p_l, p_e, p_s, n_nets, groups_per_net, gpx, abl, p_seats)
    if (args.K <= 5) GNNPN_DEC2(5);
    else if (args.K <= 8) GNNPN_DEC2(8);
    else if (args.K <= 10) GNNPN_DEC2(10);
    else GNNPN_DEC2(16);
#undef GNNPN_DEC2
    return GNNPN_OK;
}
