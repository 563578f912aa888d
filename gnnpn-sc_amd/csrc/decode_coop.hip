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

namespace {
constexpr int H = 256;
constexpr int G = 8;
constexpr int ROWS = 16;
constexpr int UNITS = H / G;      // 32
constexpr int LDH = 258;
constexpr int KMAX = 16;          // candidates per category the cooperative form is built for
constexpr unsigned SPIN_LIMIT = 400000;
}  // namespace

__global__ __launch_bounds__(256, 1) void pointer_decode_coop_kernel(DecodeArgs a, u64* __restrict__ xh,
                                                                     u64* __restrict__ xp, u64* __restrict__ xl,
                                                                     unsigned* __restrict__ err, int n_nets,
                                                                     int groups_per_net) {
    __shared__ float hs[ROWS * LDH];
    __shared__ float xs[ROWS * LDH];
    __shared__ float hsl[ROWS][UNITS + 1];
    __shared__ float part[ROWS][KMAX][G];
    __shared__ float lat[ROWS][KMAX];
    __shared__ int sel[ROWS];
    __shared__ float xin[ROWS][8];
    __shared__ int abort_flag;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kq = lane >> 4, c = lane & 15;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int gpx = (gridDim.x >> 3) / G;
    const int group = xcd * gpx + slot / G, member = slot % G;
    const int net_id = group / groups_per_net, gi = group % groups_per_net;
    if (net_id >= n_nets) return;
    const DecodeNet& net = a.net[net_id];
    if (tid == 0) abort_flag = 0;

    const int B = a.B, T = a.T, K = a.K, L = T * K;
    const bool latent_in_launch = net.latent_from >= 0;
    bool publishes_latent = false;
    for (int n = 0; n < n_nets; ++n) publishes_latent |= (a.net[n].latent_from == net_id);

    u64* xh_g = xh + (size_t)group * (2 * ROWS * H);
    u64* xp_g = xp + (size_t)group * (2 * G * ROWS * K);

    const int unit = member * UNITS + wave * 8 + (c & 7);
    int wrow[2];
    wrow[0] = (0 + (c >> 3)) * H + unit;
    wrow[1] = (2 + (c >> 3)) * H + unit;
    float wBh[2][64], wBx[2][64], bh[2], bi[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        const int gate = wrow[tl] / H, u = wrow[tl] % H;
        bh[tl] = net.bhh[wrow[tl]];
        bi[tl] = net.bih[wrow[tl]];
#pragma unroll
        for (int kk = 0; kk < 64; ++kk) {
            wBh[tl][kk] = net.whh[((size_t)(kk * 4 + gate) * H + u) * 4 + kq];
            wBx[tl][kk] = net.wih[((size_t)(kk * 4 + gate) * H + u) * 4 + kq];
        }
    }

    // in-kernel embedding2 of the picked row (net.embedded == nullptr): column tid of emb_w
    float ew[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, eb = 0.f;
    if (!net.embedded) {
#pragma unroll
        for (int i = 0; i < 8; ++i) ew[i] = net.emb_w[tid * 8 + i];
        eb = net.emb_b[tid];
    }

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
            hs[j * LDH + tid] = (b0 + j < B) ? net.h0[(int64_t)(b0 + j) * H + tid] : 0.0f;
            xs[j * LDH + tid] = net.start[tid];
        }
        __syncthreads();

        for (int k = 0; k <= T; ++k) {
            float4 xg[4];
            float xraw = 0.0f;
            if (k > 0) {
                // ---- hand-off of publish #(step-1): h_{k-1}, partial dots, Low's window logits — ONE
                // combined sweep (all loads issued, then all tags checked): one round trip, not three
                const unsigned tag = step;
                const int par = (step - 1) & 1;
                const bool has_lat = latent_in_launch || net.latent_win;
                {
                    const u64* src_h = xh_g + par * (ROWS * H) + wave * 4 * H;
                    const u64* src_p = xp_g + par * (G * ROWS * K);
                    const u64* src_l = xl + ((size_t)tile * T + (k - 1)) * ROWS * K + wave * 4 * K;
                    const int n_p = G * 4 * K;            // this wave's rows 4w..4w+3 from all members
                    const int n_l = 4 * K;
                    int p_at[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int q = lane + 64 * j;
                        const int m = q / (4 * K), rem = q - m * 4 * K;
                        p_at[j] = (m * ROWS + wave * 4 + rem / K) * K + rem % K;
                    }
                    unsigned vh[16], vp[8], vl = 0;
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
                        for (int j = 0; j < 8; ++j) {
                            if (lane + 64 * j < n_p) {
                                const u64 x = granule_load(src_p + p_at[j]);
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
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (!ok) abort_flag = 1;
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const int i = j * 64 + lane;
                        hs[(wave * 4 + (i >> 8)) * LDH + (i & 255)] = __uint_as_float(vh[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int q = lane + 64 * j;
                        if (q < n_p) {
                            const int m = q / (4 * K), rem = q - m * 4 * K;
                            part[wave * 4 + rem / K][rem % K][m] = __uint_as_float(vp[j]);
                        }
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

                // ---- logits: one thread per (row, candidate): sum of the 8 partial dots in member
                // order, C*tanh (device-library tanhf: these values decide the pick), + latent
                if (tid < ROWS * K) {
                    const int row = tid / K, r = tid - row * K, b = b0 + row;
                    float dot = part[row][r][0];
#pragma unroll
                    for (int m = 1; m < G; ++m) dot = __fadd_rn(dot, part[row][r][m]);
                    float v = a.use_tanh ? __fmul_rn(a.tanh_c, tanhf(dot)) : dot;
                    if (member == 0) {
                        if (b < B) net.win_logits[((int64_t)b * T + (k - 1)) * K + r] = v;
                        if (publishes_latent)
                            granule_store(xl + (((size_t)tile * T + (k - 1)) * ROWS + row) * K + r, 1u, v);
                    }
                    if (has_lat) v = __fadd_rn(v, lat[row][r]);
                    lat[row][r] = v;                    // biased logit
                }
                __syncthreads();
                // ---- softmax denominator + first-max argmax: one thread per row
                if (tid < ROWS) {
                    const int row = tid, b = b0 + row;
                    float best = lat[row][0];
                    int best_r = 0;
                    for (int r = 1; r < K; ++r) {
                        const float v = lat[row][r];
                        if (v > best) {                 // strict '>' keeps the first maximum
                            best = v;
                            best_r = r;
                        }
                    }
                    float denom = 0.0f;
                    for (int r = 0; r < K; ++r) denom = __fadd_rn(denom, expf(__fsub_rn(lat[row][r], best)));
                    sel[row] = (k - 1) * K + best_r;
                    if (member == 0 && b < B) {
                        net.pick_prob[(int64_t)b * T + (k - 1)] = 1.0f / denom;
                        net.idx[(int64_t)b * T + (k - 1)] = (k - 1) * K + best_r;
                    }
                }
                __syncthreads();
                if (member == 0 && tid < ROWS * 8) {
                    const int row = tid >> 3, b = b0 + row;
                    if (b < B)
                        net.actions[((int64_t)b * T + (k - 1)) * 8 + (tid & 7)] =
                            a.inputs[((int64_t)b * L + sel[row]) * 8 + (tid & 7)];
                }
                if (k == T) break;
                // x_k = embedded[idx_{k-1}] : in flight under the W_hh.h MFMAs
                if (net.embedded) {
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
            float4 ev[8];
            const int prow = tid / K, pcand = tid - prow * K;
            const bool pdot = tid < ROWS * K;
            if (pdot) {
                const int b = b0 + prow;
                const float* src = net.enc_out + ((int64_t)b * L + (int64_t)k * K + pcand) * H + member * UNITS;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    ev[j] = b < B ? *reinterpret_cast<const float4*>(src + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
            }

            // ---- decoder LSTM cell: two independent fma chains per gate column
            f32x4 ah0 = {0.f, 0.f, 0.f, 0.f}, ah1 = ah0, ax0 = ah0, ax1 = ah0;
            mfma_chain_pair<LDH>(hs, c, kq, wBh[0], wBh[1], ah0, ah1);
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
            mfma_chain_pair<LDH>(xs, c, kq, wBx[0], wBx[1], ax0, ax1);
            u64* out_h = xh_g + (step & 1) * (ROWS * H);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float g0 = __fadd_rn(__fadd_rn(ah0[r], bh[0]), __fadd_rn(ax0[r], bi[0]));
                const float g1 = __fadd_rn(__fadd_rn(ah1[r], bh[1]), __fadd_rn(ax1[r], bi[1]));
                cell_update_pair(g0, g1, c < 8, cst[r], hl[r]);
                if (c < 8) {
                    const int row = kq * 4 + r;
                    granule_store(out_h + row * H + unit, step + 1, hl[r]);
                    hsl[row][wave * 8 + (c & 7)] = hl[r];
                    if (net.queries && b0 + row < B) net.queries[((int64_t)(b0 + row) * T + k) * H + unit] = hl[r];
                }
            }
            __syncthreads();
            // ---- partial attention dots of the step-k window against the own h_k slice
            if (pdot) {
                float p = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    p = fmaf(ev[j].x, hsl[prow][4 * j + 0], p);
                    p = fmaf(ev[j].y, hsl[prow][4 * j + 1], p);
                    p = fmaf(ev[j].z, hsl[prow][4 * j + 2], p);
                    p = fmaf(ev[j].w, hsl[prow][4 * j + 3], p);
                }
                granule_store(xp_g + (step & 1) * (G * ROWS * K) + (member * ROWS + prow) * K + pcand, step + 1, p);
            }
            ++step;
        }
        if (abort_flag) break;
    }
    if (abort_flag && tid == 0) atomicOr(err, 2u);
}

extern "C" int64_t gnnpn_pointer_decode_workspace_bytes(int32_t B, int32_t T, int32_t n_per) {
    const int64_t groups = 64, tiles = (B + ROWS - 1) / ROWS;
    return 256 + groups * 2 * ROWS * H * 8 + groups * 2 * G * ROWS * (int64_t)n_per * 8 +
           tiles * T * ROWS * (int64_t)n_per * 8;
}

bool gnnpn_decode_coop_supported(int32_t H_, int32_t n_per) { return H_ == H && n_per <= KMAX; }

int gnnpn_launch_decode_coop(const DecodeArgs& args, int n_nets, void* workspace, int64_t workspace_bytes,
                             hipStream_t s) {
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        GNNPN_FAIL(GNNPN_E_LAUNCH, "pointer_decode: cannot query the device");
    const int n_tiles = (args.B + ROWS - 1) / ROWS;
    int gpx = n_cu / (8 * G);
    if (gpx > 8) gpx = 8;
    while (gpx > 1 && (gpx - 1) * 8 >= n_nets * n_tiles) --gpx;
    if (gpx < 1) GNNPN_FAIL(GNNPN_E_UNSUP, "pointer_decode: device has %d CUs, cooperative form needs >= 64", n_cu);
    const int groups = gpx * 8;
    if (groups < n_nets) GNNPN_FAIL(GNNPN_E_UNSUP, "pointer_decode: %d groups for %d nets", groups, n_nets);
    const int groups_per_net = groups / n_nets;
    const int64_t h_bytes = (int64_t)groups * 2 * ROWS * H * 8;
    const int64_t p_bytes = (int64_t)groups * 2 * G * ROWS * args.K * 8;
    const int64_t l_bytes = (int64_t)n_tiles * args.T * ROWS * args.K * 8;
    const int64_t need = 256 + h_bytes + p_bytes + l_bytes;
    if (!workspace || workspace_bytes < need || !gnnpn_aligned(workspace, 256))
        GNNPN_FAIL(GNNPN_E_ARG, "pointer_decode: workspace of %lld B (256-B aligned) required", (long long)need);
    if (hipMemsetAsync(workspace, 0, (size_t)need, s) != hipSuccess)
        GNNPN_FAIL(GNNPN_E_LAUNCH, "pointer_decode: workspace memset failed");
    char* base = static_cast<char*>(workspace);
    hipLaunchKernelGGL(pointer_decode_coop_kernel, dim3(groups * G), dim3(256), 0, s, args,
                       reinterpret_cast<u64*>(base + 256), reinterpret_cast<u64*>(base + 256 + h_bytes),
                       reinterpret_cast<u64*>(base + 256 + h_bytes + p_bytes), reinterpret_cast<unsigned*>(base),
                       n_nets, groups_per_net);
    return GNNPN_OK;
}
