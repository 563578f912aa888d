// Pointer-network LSTM encoder recurrence, cooperative form with TWO independent recurrences per
// workgroup (v3).
//
// lstm_coop.hip runs one (net, 16-problem tile) recurrence per group of 8 CUs; its 4 waves move in
// lock step, so while they wait for the hand-off or run the cell update the CU's matrix pipes idle
// (MFMA busy ~40 % of a step).  Here a group is 16 workgroups of 8 waves; waves 0-3 ("half 0") and
// waves 4-7 ("half 1") of every workgroup run two DIFFERENT recurrences — normally the Low and the
// High net of the same 16 problems — that never synchronise with each other: each half has its own
// LDS tile (double-buffered by step parity), its own 4-wave barrier (an LDS counter), its own
// hand-off buffers.  One half's MFMA phase therefore overlaps the other half's hand-off wait and
// cell update on the same SIMDs (half 1 starts half a step late to seed the stagger).
//
// Ownership: member m of the group keeps the W_hh rows of hidden units [16m, 16m+16) of BOTH halves'
// nets in registers; wave w of a half owns units 16m+4w..+3 = one 16-column MFMA tile
// [i(4) | f(4) | g(4) | o(4)] = 64 B-fragment VGPRs, 64 MFMAs per step as ONE k-ordered fp32 chain per
// column (bit-identical to the other implementations).  The four gates of a unit sit 4 lanes apart:
// three DPP row rotations bring them together for the cell update.
// Hand-off, tags, parity double buffering, sentinel-hinted waiting, bounded spins, same-XCD fast
// path: exactly as lstm_coop.hip (coop_common.h).
#include "common.h"
#include "recurrent.h"
#include "lstm_shared.h"
#include "coop_common.h"

namespace {
constexpr int H = 256;
constexpr int G2 = 16;            // workgroups per group
constexpr int ROWS = 16;          // problems per tile (MFMA M)
constexpr int U2 = H / G2;        // hidden units per member (16)
constexpr int LDH = 258;
constexpr unsigned SPIN_LIMIT = 400000;
constexpr int HALF_GRANULES = 2 * ROWS * H + 2 * 4 * G2;   // h granules (2 parities) + sentinels
constexpr int STAGGER_CYCLES = 2600;
}  // namespace

__device__ __forceinline__ float ror16f(float v, int n) {   // n in {4, 8, 12}
    const int i = __float_as_int(v);
    int r;
    if (n == 4) r = __builtin_amdgcn_update_dpp(0, i, 0x124, 0xF, 0xF, false);
    else if (n == 8) r = __builtin_amdgcn_update_dpp(0, i, 0x128, 0xF, 0xF, false);
    else r = __builtin_amdgcn_update_dpp(0, i, 0x12C, 0xF, 0xF, false);
    return __int_as_float(r);
}

// barrier among the 4 waves of one half: arrive on an LDS counter, wait until it reaches `target`
__device__ __forceinline__ bool half_barrier(unsigned* cnt, unsigned target, int* abort_flag, int lane) {
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (unsigned spins = 0;; ++spins) {
        if (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) >= target) return true;
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) || spins > 4 * SPIN_LIMIT)
            return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// one wave's quarter of the tile (rows 4w..4w+3): see sweep_quarter of lstm_coop.hip
__device__ __forceinline__ bool sweep_quarter2(const u64* buf, const u64* sentinels, unsigned tag, float* hs, int w4,
                                               int lane, bool keep) {
    const u64* src = buf + w4 * 4 * H;
    unsigned v[16];
    unsigned spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const u64 x = granule_load(src + j * 64 + lane);
            v[j] = (unsigned)x;
            ok &= (unsigned)(x >> 32) == tag;
        }
        if (__all(ok)) break;
        int nap = 1;
        for (;;) {   // wait for the hint: 4*G2 = 64 sentinels, one per lane
            const bool seen = (unsigned)(granule_load(sentinels + lane) >> 32) >= tag;
            if (__all(seen)) break;
            if (++spins > SPIN_LIMIT) return false;
            for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(2);
            if (nap < 16) nap <<= 1;
        }
        if (++spins > SPIN_LIMIT) return false;
    }
    if (keep) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = j * 64 + lane;
            hs[(w4 * 4 + (i >> 8)) * LDH + (i & 255)] = __uint_as_float(v[j]);
        }
    }
    return true;
}

__global__ __launch_bounds__(512) void lstm_encode_coop2_kernel(LstmNets nets, u64* __restrict__ xchg,
                                                                unsigned* __restrict__ err, int32_t B, int32_t L,
                                                                int n_nets, int groups, int ablate) {
    __shared__ float hs_all[2][2][ROWS * LDH];                           // [half][parity]
    __shared__ __attribute__((aligned(16))) float hst_all[2][2][ROWS][U2];   // [half][parity] own h slices, staged for 64-B stores
    __shared__ unsigned bar[2];
    __shared__ int abort_flag;
    __shared__ int xcd_flag;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = wave >> 2, w4 = wave & 3;
    const int kq = lane >> 4, c = lane & 15, gate = c >> 2;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int gpx = (gridDim.x >> 3) / G2;
    const int group = xcd * gpx + slot / G2, member = slot % G2;
    if (threadIdx.x == 0) {
        abort_flag = 0;
        bar[0] = bar[1] = 0;
    }
    const int same = group_same_xcd<G2>(err, group, member, &xcd_flag, SPIN_LIMIT);   // contains __syncthreads()
    if (same < 0) {
        if (threadIdx.x == 0) atomicOr(err, 4u);
        return;
    }
    const bool same_xcd = same == 1 && !(ablate & 128);
    if (threadIdx.x == 0 && same_xcd) atomicAdd(err + 1, 1u);
    // ---- from here on the two halves never meet again: no __syncthreads() below -------------------

    const int n_tiles = (B + ROWS - 1) / ROWS;
    const int n_jobs = n_nets * n_tiles;                 // job j: net j % n_nets, tile j / n_nets
    const int job0 = 2 * group + half;
    if (job0 >= n_jobs) return;
    const int net = job0 % n_nets;                       // constant for this half: (2*groups) % n_nets == 0

    const float* __restrict__ pre = nets.pregates[net];
    const float* __restrict__ xin = nets.inputs[net];
    const float* __restrict__ Wp = nets.whh[net];
    float* __restrict__ enc = nets.enc_out[net];
    u64* xg = xchg + (size_t)(group * 2 + half) * HALF_GRANULES;
    u64* sent = xg + 2 * ROWS * H;                       // [parity][member*4 + w4]
    unsigned* my_bar = &bar[half];

    // this lane's gate column: unit = 16m + 4w + (c&3), gate = c>>2  (tile = [i | f | g | o], 4 units each)
    const int unit = member * U2 + w4 * 4 + (c & 3);
    const int wrow = gate * H + unit;
    float wB[64], bh, wX[2] = {0.f, 0.f}, bx = 0.f;
    bh = nets.bhh[net][wrow];
    if (!pre) {
        wX[0] = nets.w_in[net][wrow * 8 + kq];
        wX[1] = nets.w_in[net][wrow * 8 + 4 + kq];
        bx = nets.b_in[net][wrow];
    }
#pragma unroll
    for (int kk = 0; kk < 64; ++kk) wB[kk] = Wp[((size_t)(kk * 4 + gate) * H + unit) * 4 + kq];

    if (half == 1) {   // seed the stagger: half 1 starts about half a step after half 0
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < STAGGER_CYCLES) __builtin_amdgcn_s_sleep(4);
    }

    unsigned step = 0;       // running step counter of this half: tag = step+1, parity = step&1
    unsigned n_bar = 0;      // barriers passed by this half
    bool first_job = true, alive = true;
    for (int job = job0; job < n_jobs && alive; job += 2 * groups) {
        const int tile = job / n_nets, b0 = tile * ROWS;
        float cst[4] = {0.f, 0.f, 0.f, 0.f}, hlast[4] = {0.f, 0.f, 0.f, 0.f};
        float pg[4], pg_next[4] = {0.f, 0.f, 0.f, 0.f}, ax[2] = {0.f, 0.f}, ax_next[2] = {0.f, 0.f};
        auto load_input = [&](int t, float (&pgv)[4], float (&axv)[2]) {
            if (pre) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = b0 + kq * 4 + r;
                    pgv[r] = b < B ? pre[((int64_t)b * L + t) * (4 * H) + wrow] : 0.0f;
                }
            } else if (b0 + c < B) {
                const float* row = xin + ((int64_t)(b0 + c) * L + t) * 8;
                axv[0] = row[kq];
                axv[1] = row[4 + kq];
            }
        };
        load_input(0, pg_next, ax_next);
        for (int t = 0; t < L; ++t, ++step) {
            float* hs = hs_all[half][step & 1];
            const bool stamps = ablate & 32;
            u64 s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
            if (stamps) s0 = phase_stamp();
            ax[0] = ax_next[0];
            ax[1] = ax_next[1];
#pragma unroll
            for (int r = 0; r < 4; ++r) pg[r] = pg_next[r];

            bool ok = true;
            if (t == 0) {
                for (int i = (threadIdx.x & 255); i < ROWS * LDH; i += 256) hs[i] = 0.0f;
                if (!first_job)   // proves every peer is done with the buffer about to be overwritten
                    ok = sweep_quarter2(xg + ((step - 1) & 1) * (ROWS * H), sent + ((step - 1) & 1) * (4 * G2), step, hs,
                                        w4, lane, false);
            } else {
                ok = sweep_quarter2(xg + ((step - 1) & 1) * (ROWS * H), sent + ((step - 1) & 1) * (4 * G2), step, hs, w4,
                                    lane, true);
            }
            if (!ok) __hip_atomic_store(&abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (stamps) s1 = phase_stamp();
            n_bar += 4;
            if (!half_barrier(my_bar, n_bar, &abort_flag, lane) || !ok) {
                alive = false;
                break;
            }
            if (stamps) s2 = phase_stamp();
            if (t + 1 < L) load_input(t + 1, pg_next, ax_next);
            // enc_out of the previous step leaves after the hand-off wait, 64 B per problem row
            if (t > 0 && (threadIdx.x & 255) < ROWS * 4) {
                const int row = (threadIdx.x & 255) >> 2, q = threadIdx.x & 3;
                if (b0 + row < B)
                    *reinterpret_cast<float4*>(enc + ((int64_t)(b0 + row) * L + (t - 1)) * H + member * U2 + 4 * q) =
                        *reinterpret_cast<const float4*>(&hst_all[half][(step - 1) & 1][row][4 * q]);
            }

            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (t > 0) {
                const float* base = hs + c * LDH + kq;
                float a[2][16];
#pragma unroll
                for (int i = 0; i < 16; ++i) a[0][i] = base[4 * i];
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    if (ch < 3) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) a[(ch + 1) & 1][i] = base[4 * (16 * (ch + 1) + i)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ch & 1][i], wB[16 * ch + i], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (stamps) {
                asm volatile("" ::"v"(acc[0]));
                s3 = phase_stamp();
            }
            if (!pre) {
                f32x4 px = {0.f, 0.f, 0.f, 0.f};
                px = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[0], wX[0], px, 0, 0, 0);
                px = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[1], wX[1], px, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) pg[r] = __fadd_rn(px[r], bx);
            }

            u64* out_buf = xg + (step & 1) * (ROWS * H);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // gates = (h.W_hh^T + b_hh) + (x.W_ih^T + b_ih); this lane holds gate `gate` of unit (c&3)
                const float gpre = __fadd_rn(__fadd_rn(acc[r], bh), pg[r]);
                const float a0 = cell_act(gpre, gate == 2);
                // lanes of gate 0 collect f, g, o of the same unit from 4, 8, 12 lanes up the row
                const float r4 = ror16f(a0, 4), r8 = ror16f(a0, 8), r12 = ror16f(a0, 12);
                const float ig = a0, fg = r12, gg = r8, og = r4;
                cst[r] = __fadd_rn(__fmul_rn(fg, cst[r]), __fmul_rn(ig, gg));
                hlast[r] = __fmul_rn(og, cell_act(cst[r], true));
                if (gate == 0) {
                    const int row = kq * 4 + r;
                    granule_publish(out_buf + row * H + unit, step + 1, hlast[r], same_xcd);
                    hst_all[half][step & 1][row][w4 * 4 + (c & 3)] = hlast[r];
                }
            }
            if (lane == 0) granule_publish(sent + (step & 1) * (4 * G2) + member * 4 + w4, step + 1, 0.0f, same_xcd);
            if (stamps) {
                asm volatile("" ::"v"(hlast[3]));
                s4 = phase_stamp();
                if (blockIdx.x == 0 && w4 == 0 && lane == 0 && t > 0) {
                    u64* prof = reinterpret_cast<u64*>(err) + 4 + half * 8;
                    prof[0] += s1 - s0;   // sweep + LDS fill
                    prof[1] += s2 - s1;   // 4-wave barrier
                    prof[2] += s3 - s2;   // flush + A-frag reads + 64 MFMAs
                    prof[3] += s4 - s3;   // projection + cell + publish
                    prof[4] += 1;
                }
            }
        }
        if (!alive) break;
        // last step's slice: wait for the siblings' hst writes, then store; also h_n / c_n
        n_bar += 4;
        if (!half_barrier(my_bar, n_bar, &abort_flag, lane)) {
            alive = false;
            break;
        }
        if ((threadIdx.x & 255) < ROWS * 4) {
            const int row = (threadIdx.x & 255) >> 2, q = threadIdx.x & 3;
            if (b0 + row < B)
                *reinterpret_cast<float4*>(enc + ((int64_t)(b0 + row) * L + (L - 1)) * H + member * U2 + 4 * q) =
                    *reinterpret_cast<const float4*>(&hst_all[half][(step - 1) & 1][row][4 * q]);
        }
        if (gate == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = b0 + kq * 4 + r;
                if (b < B) {
                    nets.h_n[net][(int64_t)b * H + unit] = hlast[r];
                    nets.c_n[net][(int64_t)b * H + unit] = cst[r];
                }
            }
        }
        // the siblings must have read hst before the next job's first cell update rewrites it
        n_bar += 4;
        if (!half_barrier(my_bar, n_bar, &abort_flag, lane)) {
            alive = false;
            break;
        }
        first_job = false;
    }
    if (!alive && lane == 0) atomicOr(err, 1u);
}

int64_t gnnpn_encode_coop2_workspace_bytes() {
    return COOP_STATUS_BYTES + (int64_t)16 * 2 * HALF_GRANULES * sizeof(u64);
}

// returns GNNPN_E_UNSUP (without touching the error text) when this form does not fit the call, so
// that the dispatcher can use lstm_coop.hip instead
int gnnpn_launch_encode_coop2(const LstmNets& nets, int n_nets, int32_t B, int32_t L, void* workspace,
                              int64_t workspace_bytes, hipStream_t s) {
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return GNNPN_E_UNSUP;
    const int n_tiles = (B + ROWS - 1) / ROWS;
    const int n_jobs = n_nets * n_tiles;
    int gpx = n_cu / (8 * G2);                       // groups per XCD (2 on 256 CUs)
    if (gpx > 2) gpx = 2;
    while (gpx > 1 && (gpx - 1) * 8 * 2 >= n_jobs) --gpx;
    if (gpx < 1) return GNNPN_E_UNSUP;
    const int groups = gpx * 8;
    if ((2 * groups) % n_nets != 0) return GNNPN_E_UNSUP;
    const int64_t need = COOP_STATUS_BYTES + (int64_t)groups * 2 * HALF_GRANULES * sizeof(u64);
    if (!workspace || workspace_bytes < need || !gnnpn_aligned(workspace, 256)) return GNNPN_E_UNSUP;
    if (hipMemsetAsync(workspace, 0, (size_t)need, s) != hipSuccess) return GNNPN_E_UNSUP;
    hipLaunchKernelGGL(lstm_encode_coop2_kernel, dim3(groups * G2), dim3(512), 0, s, nets,
                       reinterpret_cast<u64*>(static_cast<char*>(workspace) + COOP_STATUS_BYTES),
                       reinterpret_cast<unsigned*>(workspace), B, L, n_nets, groups, gnnpn_option_lstm_ablate());
    return GNNPN_OK;
}
