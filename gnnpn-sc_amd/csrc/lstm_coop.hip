// Pointer-network LSTM encoder recurrence, cooperative form (v2).
//
// v1 (lstm.hip) gives each workgroup whole problems and re-streams the 1 MiB W_hh from L2 every
// step (~11 us/step measured).  Here the recurrent weights never move: a GROUP of G = 8
// workgroups (one per CU) owns a tile of 16 problems; member m keeps the W_hh rows of hidden
// units [32m, 32m+32) (128 gate columns) in REGISTERS as fp32 MFMA B-fragments (128 VGPRs/lane)
// for the whole launch.  Per step a member
//   1. gathers the tile's full h_{t-1} [16 x 256] from the group's exchange buffer into LDS,
//   2. runs 128 v_mfma_f32_16x16x4_f32 per wave (exact fp32, k-ordered fma chain: bit-identical
//      to v1's per-thread fmaf chain),
//   3. adds bias + pre-gates, applies the cell update in registers (c never leaves them),
//   4. writes its h_t slice to enc_out and PUBLISHES it to the 7 peers.
// Hand-off = 8-byte {tag = step+1, value} granules written with ONE sc1 (write-through, agent
// scope) store each and swept by sc1 loads until every tag matches: the data is the flag, no
// fences, no separate flag round trip, placement independent (CDNA4 guide, Guideline 16 / R2).
// Buffers are double-buffered by step parity; a member can run at most one step ahead of any
// peer, so a slot is never overwritten before every peer has read it.  Every spin is bounded:
// on timeout the workgroup raises the launch's error word and leaves, so the grid always drains.
// Groups are placed on one XCD (blockIdx % 8 equal) for speed only.
#include "common.h"
#include "recurrent.h"
#include "lstm_shared.h"
#include "coop_common.h"
#include "decode_shared.h"   // gnnpn_decode_diag_buffer: the failure record both recurrent kernels write

namespace {
constexpr int H = 256;
constexpr int G = 8;              // workgroups per group
constexpr int ROWS = 16;          // problems per tile (MFMA M)
constexpr int UNITS = H / G;      // hidden units per member
constexpr unsigned SPIN_LIMIT = 400000;   // sweep passes before giving up (~0.3 s)
constexpr int GROUP_GRANULES = 2 * ROWS * H + 2 * 4 * G;   // h granules (2 parities) + 64 spare
constexpr int ENC_LOOP_PAD_NOPS = 13;      // see the anchor in front of the step loop
}  // namespace

// Sweep this wave's quarter (rows 4w..4w+3, all 256 units) of one parity buffer until every
// granule carries `tag`; values go to LDS.  Returns false on timeout.  Eight 16-byte loads per lane (two granules each).
template <int PREC>
__device__ __forceinline__ bool sweep_quarter(const u64* buf, unsigned tag, float* hs, int wave, int lane, bool keep,
                                              bool nowait = false) {
    const u64* src = uniform_ptr(buf + wave * 4 * H);      // group, parity, wave: uniform over the wave
    const unsigned voff = 16u * lane;
    u32x4 v[8];                                              // granules 2*(64 j + lane), +1: {value, tag, value, tag}
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "=v"(v[j]));   // "defined" without an instruction: the loop below owns the registers
    // ONE load statement in ONE loop (its operands tied in place): with a first-pass load and a separate re-read statement the
    // register allocator copied all 32 registers twice per step to merge the two.  The first pass is optimistic (peers run in
    // lock step, so it normally succeeds).  If it does not, only the LANES that miss something re-read (a late member's 32
    // units sit in 16 lanes of the wave), between naps — not the whole quarter (with two launches sharing the CUs, full
    // re-sweeps swamp the L2 and slow the very peers it waits for).
    bool bad = true;                                         // this lane is still missing a tag
    unsigned spins = 0;
    int nap = 1;
    for (;;) {
        granule_reload2_x8_masked(v, src, voff, __ballot(bad));
        bad = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (PREC == 2) bad |= !split_pair_tagged(v[j], tag);   // {p0 | tag16, p1 | p2} granules (coop_common.h)
            else bad |= (v[j].y != tag) | (v[j].w != tag);
        }
        if (!__any(bad) || nowait) break;
        if (++spins > SPIN_LIMIT) return false;
        for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(2);
        if (nap < 16) nap <<= 1;
    }
    if (keep) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = 2 * (j * 64 + lane);      // 0..1022 (even) within the quarter; i and i+1 share a row
            if constexpr (PREC == 0) {
                float* d = &hs[ht_index(wave * 4 + (i >> 8), i & 255)];       // units i, i+1: 64 floats apart (coop_common.h)
                d[0] = __uint_as_float(v[j].x);
                d[64] = __uint_as_float(v[j].z);
            } else {
                _Float16* h16 = reinterpret_cast<_Float16*>(hs) + (wave * 4 + (i >> 8)) * LDH16 + (i & 255);
                if constexpr (PREC == 2) {
                    split_pair_to_lds(h16, v[j]);          // the publisher split the value: three packed stores, no arithmetic
                } else {
                    h16[0] = (_Float16)__uint_as_float(v[j].x);
                    h16[1] = (_Float16)__uint_as_float(v[j].z);
                }
            }
        }
    }
    return true;
}


// F16 = the reduced-precision encoder of BASELINE configs[4]: W_hh and h_{t-1} enter the recurrent
// product as fp16 (v_mfma_f32_16x16x32_f16, fp32 accumulate: 16 MFMAs per step instead of 128); bias,
// input projection, cell update, state c, the published h and enc_out stay fp32.  Opt-in, NOT
// parity-exact: callers report an index-agreement rate against the fp32 path.
// PRE: the input side arrives as stored pre-gates (else: folded, from the raw rows).  DIAG: the diagnostic
// build (phase stamps / ablation switches of tools/ablate_encode.py); the production instantiation carries
// none of that code (`ablate` is then the constant 0).
// PREC 2 = exact-split operands (coop_common.h): W_hh and h_{t-1} each as THREE fp16 pieces that reproduce the fp32 value
// bit for bit, every cross term that can reach 2^-24 of a product kept (6 products on the fp16 matrix cores, fp32
// accumulation in three magnitude classes: 96 MFMAs of 16 cycles per step instead of 128 of 32).  h travels between the
// members already split (the publisher splits its own value once).  Error bound and measurements: DESIGN.md section 5; profiles/LOG_r01_r04.md section 12.
template <int PREC, bool PRE, bool DIAG>
__global__ __launch_bounds__(256, 2) void lstm_encode_coop_kernel(LstmNets nets, u64* __restrict__ xchg,
                                                                  unsigned* __restrict__ err, unsigned* __restrict__ sticky,
                                                                  int32_t B, int32_t L, int n_nets, int groups_per_net,
                                                                  int gpx, int ablate_arg, unsigned* __restrict__ seats,
                                                                  unsigned* __restrict__ diag) {
    const int ablate = DIAG ? (ablate_arg & ~0x1000) : (ablate_arg & 128);   // bit 7 = gnnpn_launch_opts_t.write_through (a tested mode)
    constexpr bool F16 = PREC != 0, SPLIT = PREC == 2;
    constexpr int HS_FLOATS = PREC == 2 ? 3 * SPLIT_TILE / 2 : ROWS * LDH16;
    __shared__ __attribute__((aligned(16))) float hs[HS_FLOATS];      // fp32 tile (k-quarter-major, stride LDT) | fp16 tile | three fp16 piece tiles (stride LDH16 halfs)
    __shared__ __attribute__((aligned(16))) float hst[ROWS][UNITS];   // own h slice, staged for whole-line stores
    __shared__ int abort_flag;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kq = lane >> 4, c = lane & 15;
    // placement by claim (coop_common.h): one member per CU, a group's members on one XCD
    __shared__ int place[2];
    int group, member;
    u64 p_entry = 0, p_placed = 0, p_weights = 0;          // diagnostic build: stamps of the launch's fixed part (prof[7..9] below)
    if constexpr (DIAG) p_entry = phase_stamp();
    if (!coop_place<G>(err, gpx, place, group, member, seats, (ablate_arg & 0x1000) != 0, sticky)) return;   // surplus workgroup of the over-subscribed launch (bit 12: opts.paired_start)
    if constexpr (DIAG) p_placed = phase_stamp();
    const int net = group / groups_per_net, gi = group % groups_per_net;
    if (net >= n_nets) return;                            // spare group: takes part in no exchange
    if (threadIdx.x == 0) abort_flag = 0;
    __syncthreads();
    const bool same_xcd = !(ablate & 128);
    if (threadIdx.x == 0 && same_xcd) atomicAdd(err + COOP_PLACED_OFFSET / 4 + COOP_XCD_STRIDE * xcc_id(), 1u);   // statistics: workgroups on the fast path (per XCD: its own line)

    const float* __restrict__ pre = PRE ? nets.pregates[net] : nullptr;
    const float* __restrict__ xin = nets.inputs[net];     // used when pre == nullptr (F = 8)
    const float* __restrict__ Wp = nets.whh[net];
    float* __restrict__ enc = nets.enc_out[net];
    u64* xg = xchg + (size_t)group * GROUP_GRANULES;

    // this lane's two gate columns: tile 0 = [i | f], tile 1 = [g | o], 8 units per wave
    const int unit = member * UNITS + wave * 8 + (c & 7);
    int wrow[2];
    wrow[0] = (0 + (c >> 3)) * H + unit;
    wrow[1] = (2 + (c >> 3)) * H + unit;
    float wB[F16 ? 1 : 2][F16 ? 1 : 64], bh[2], wX[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, bx[2] = {0.f, 0.f};
    f16x8 wB16[F16 ? 2 : 1][8];   // fp16 B-fragments: lane (c, kq) holds W[col c][32kk + 8kq + j], j = 0..7
    f16x8 wL16[SPLIT ? 2 : 1][8];                            // exact split: the second pieces (the third: LDS, coop_common.h)
    float winv[2] = {1.0f, 1.0f};                            // ... and the columns' un-scaling factors
    __shared__ __attribute__((aligned(16))) unsigned wts[SPLIT ? SPLIT_WT_DWORDS : 4];
    unsigned* wt_lane = wts + (SPLIT ? (wave * 8 * 64 + lane) * 4 : 0);
    const uint4* __restrict__ Wsplit = SPLIT ? static_cast<const uint4*>(nets.whh_split[net]) : nullptr;   // uniform: packed once per model, or split here
    if constexpr (SPLIT) {
        if (Wsplit) load_split_weights(Wsplit + (size_t)member * SPLIT_PACK_U4_PER_MEMBER, threadIdx.x, wB16, wL16, wt_lane, winv);
    }
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        const int gate = wrow[tl] / H, u = wrow[tl] % H;
        bh[tl] = nets.bhh[net][wrow[tl]];
        if constexpr (SPLIT) {
            if (!Wsplit) winv[tl] = split_weights<H>(Wp, gate, u, kq, wB16[tl], wL16[tl], wt_lane + 2 * tl);
        } else if constexpr (F16) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    wB16[tl][kk][j] = (_Float16)Wp[((size_t)((8 * kk + 2 * kq + (j >> 2)) * 4 + gate) * H + u) * 4 + (j & 3)];
        }
        if constexpr (!PRE) {   // B-fragments of the folded input projection: w_in[wrow][4*kk2 + kq], kk2 = 0,1
            wX[tl][0] = nets.w_in[net][wrow[tl] * 8 + kq];
            wX[tl][1] = nets.w_in[net][wrow[tl] * 8 + 4 + kq];
            bx[tl] = nets.b_in[net][wrow[tl]];
        }
        if constexpr (!F16) {
#pragma unroll
            for (int kk = 0; kk < 64; ++kk)   // packed layout: W[g*H+u][4kk+kq] = Wp[((kk*4+g)*H+u)*4+kq]
                wB[tl][kk] = Wp[((size_t)(kk * 4 + gate) * H + u) * 4 + kq];
        }
    }

    if constexpr (DIAG) {
        asm volatile("" ::"v"(winv[0]), "v"(winv[1]), "v"(bh[0]), "v"(bh[1]));
        p_weights = phase_stamp();
    }
    const int n_tiles = (B + ROWS - 1) / ROWS;
    unsigned step = 0;                                     // running step counter: tag = step+1, parity = step&1
    unsigned tiles_done = 0;                               // proof of work (coop_note_finished)
    bool first_tile = true;
    for (int tile = gi; tile < n_tiles; tile += groups_per_net) {
        const int b0 = tile * ROWS;
        // cell state / output of the two rows this lane finishes: rows kq*4 + {0,1} (c < 8) or kq*4 + {2,3} (c >= 8)
        f32x2 cst = {0.f, 0.f};
        f32x2 hlast = {0.f, 0.f};
        // Input side, software-pipelined by one step: the loads of step t+1 are issued AFTER step t's
        // hand-off wait (vector-memory operations retire in issue order, so an HBM-latency load issued
        // before the sweep would be waited for by the sweep) and are consumed a whole step later.
        float pg[2][4], pg_next[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        float ax[2] = {0.f, 0.f}, ax_next[2] = {0.f, 0.f};
        auto load_input = [&](int t, float (&pgv)[2][4], float (&axv)[2]) {
            if constexpr (PRE) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = b0 + kq * 4 + r;
#pragma unroll
                    for (int tl = 0; tl < 2; ++tl)
                        pgv[tl][r] = b < B ? pre[((int64_t)b * L + t) * (4 * H) + wrow[tl]] : 0.0f;
                }
            } else if (b0 + c < B) {   // raw 8-feature row as MFMA A-fragments (row c, k = 4*kk2 + kq)
                const float* row = xin + ((int64_t)(b0 + c) * L + t) * 8;
                axv[0] = row[kq];
                axv[1] = row[4 + kq];
            }
        };
        load_input(0, pg_next, ax_next);
        // Where the step loop lies in the instruction stream decides 5 % of a solo step (round 6, profiles/LOG_r06.md section 17): the
        // same loop, moved in 4-byte steps through a 64-byte line, takes 559-560 us per QWS launch at two positions 32 bytes apart and
        // 578-590 us at the other fourteen (instruction fetch of a latency-bound wave: one wavefront per SIMD, nothing to hide a
        // fetch bubble behind) — and every change to the placement code in front of it used to move it.  So the loop is anchored:
        // aligned to 64 bytes here, then ENC_LOOP_PAD_NOPS no-ops put its product phase (the first v_mfma_f32_16x16x32_f16 of
        // the disassembly) at 44 bytes into a 64-byte line — 12 and 44 are the two good positions solo, and with two slots in flight
        // (which care little: +-0.3 %) 44 measured a shade better (tools/r06/loop_alignment_scan.sh).  Re-tune when the loop body
        // changes: tools/r06/loop_anchor_check.sh prints the offset.
#ifndef GNNPN_ENC_PAD_NOPS
#define GNNPN_ENC_PAD_NOPS ENC_LOOP_PAD_NOPS
#endif
        if constexpr (SPLIT && !DIAG) asm volatile(".p2align 6\n\t.rept %0\n\ts_nop 0\n\t.endr" ::"n"(GNNPN_ENC_PAD_NOPS) : "memory");
        for (int t = 0; t < L; ++t, ++step) {
            const bool stamps = ablate & 32;
            u64 s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
            if (stamps) s0 = phase_stamp();
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                ax[tl] = ax_next[tl];
#pragma unroll
                for (int r = 0; r < 4; ++r) pg[tl][r] = pg_next[tl][r];
            }
            // x_t . w_in^T as its own k-ordered chain: issued HERE, in front of the sweep — it depends on nothing of this step,
            // and the matrix pipe works it off while the sweep's loads are in flight (it used to sit behind the main chain,
            // ~130 cycles on the critical path in front of the cell update)
            f32x4 px0 = {0.f, 0.f, 0.f, 0.f}, px1 = {0.f, 0.f, 0.f, 0.f};
            if constexpr (!PRE) {
                px0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[0], wX[0][0], px0, 0, 0, 0);
                px1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[0], wX[1][0], px1, 0, 0, 0);
                px0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[1], wX[0][1], px0, 0, 0, 0);
                px1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[1], wX[1][1], px1, 0, 0, 0);
            }
            // h_{t-1}: zeros at t == 0, else the peers' published slices.  At a tile switch the
            // sweep still runs (values dropped): it proves every peer is done with the buffer
            // this member is about to overwrite.
            bool ok = true;
            if (t == 0) {
                for (int i = threadIdx.x; i < HS_FLOATS; i += 256) hs[i] = 0.0f;
                if (!first_tile) ok = sweep_quarter<PREC>(xg + ((step - 1) & 1) * (ROWS * H), step, hs, wave, lane, false);
            } else if (!(ablate & 8)) {
                ok = sweep_quarter<PREC>(xg + ((step - 1) & 1) * (ROWS * H), step, hs, wave, lane, true, ablate & 4);
            }
            if (!ok) abort_flag = 1;
            if constexpr (DIAG) {   // failure record of the diagnostic build (shared with the decoder's, gnnpn_decode_diag; p_missing =
                if (!ok && lane == 0 && diag) {   // 0xE0C marks an encoder entry): seats taken and workgroups ARRIVED per XCD so far
                    const unsigned n = atomicAdd(diag, 1u);
                    if (n < 31) {
                        unsigned* rec = diag + 16 * (n + 1);
                        const unsigned* cnt = err + COOP_XCDCNT_OFFSET / 4;
                        const unsigned* arr = err + COOP_ARRIVE_OFFSET / 4;
                        unsigned c0 = 0, c1 = 0, a0 = 0, a1 = 0;
                        for (int x = 0; x < 4; ++x) {
                            c0 |= (cnt[COOP_XCD_STRIDE * x] & 0xffu) << (8 * x);
                            c1 |= (cnt[COOP_XCD_STRIDE * (4 + x)] & 0xffu) << (8 * x);
                            a0 |= (arr[COOP_XCD_STRIDE * x] & 0xffu) << (8 * x);
                            a1 |= (arr[COOP_XCD_STRIDE * (4 + x)] & 0xffu) << (8 * x);
                        }
                        rec[0] = group; rec[1] = member; rec[2] = tile; rec[3] = t; rec[4] = wave; rec[5] = step;
                        rec[6] = 0; rec[7] = 0xE0Cu; rec[8] = 0; rec[9] = c0; rec[10] = c1; rec[11] = a0; rec[12] = a1;
                        rec[13] = gpx; rec[14] = blockIdx.x; rec[15] = err[0];
                    }
                }
            }
            if (stamps) s1 = phase_stamp();
            __syncthreads();
            if (abort_flag) break;
            if (stamps) s2 = phase_stamp();
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            u64 s2a = 0;
            if (stamps) s2a = phase_stamp();
            if constexpr (SPLIT) {
                if (t > 0) {
                    f32x4 acc[2] = {acc0, acc1};
                    split_chain(reinterpret_cast<const _Float16*>(hs) + c * LDH16 + 8 * kq, wB16, wL16, wt_lane, winv, acc);
                    acc0 = acc[0];
                    acc1 = acc[1];
                }
            } else if constexpr (F16) {
                if (t > 0) {
                    const _Float16* base = reinterpret_cast<const _Float16*>(hs) + c * LDH16 + 8 * kq;
                    f16x8 a16[8];
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) a16[kk] = *reinterpret_cast<const f16x8*>(base + 32 * kk);
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[kk], wB16[0][kk], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[kk], wB16[1][kk], acc1, 0, 0, 0);
                    }
                }
            } else {
                if (t > 0 && !(ablate & 1)) mfma_chain_pair<LDT, 16, true>(hs, c, kq, wB[0], wB[F16 ? 0 : 1], acc0, acc1);
            }
            if (t + 1 < L && !(ablate & 0x400)) load_input(t + 1, pg_next, ax_next);
            // Issued BEHIND the MFMA stream (they would otherwise sit in front of it: measured 1.3k cycles per
            // step): the next step's input loads, and enc_out of the PREVIOUS step — after the hand-off wait
            // (stores retire in issue order with the sweep's loads, so storing before the sweep would lengthen
            // it), as whole 128-B lines (one per problem row) out of the LDS staging tile.
            if (t > 0 && threadIdx.x < ROWS * 8 && !(ablate & 0x200)) {
                const int row = threadIdx.x >> 3, q = threadIdx.x & 7;
                if (b0 + row < B)
                    *reinterpret_cast<float4*>(enc + ((int64_t)(b0 + row) * L + (t - 1)) * H + member * UNITS + 4 * q) =
                        *reinterpret_cast<const float4*>(&hst[row][4 * q]);
            }

            if (stamps) {
                asm volatile("" ::"v"(acc0[0]), "v"(acc1[0]));   // the MFMA chains have retired
                s3 = phase_stamp();
            }
            if constexpr (!PRE) {   // ... + b_in (as gnnpn_linear_f32 would)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pg[0][r] = __fadd_rn(px0[r], bx[0]);
                    pg[1][r] = __fadd_rn(px1[r], bx[1]);
                }
            }
            __syncthreads();   // everyone is done reading hs before the next step's sweep rewrites it
            if (stamps) s4 = phase_stamp();

            u64* out_buf = xg + (step & 1) * (ROWS * H);
            // the four rows' gate sums, then ONE cell update per lane for the two rows it finishes (cell_update_split)
            {
                // gates = (h.W_hh^T + b_hh) + (x.W_ih^T + b_ih); lanes c<8 hold (i,g), c>=8 hold (f,o)
                f32x2 g0[2], g1[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    g0[q] = (f32x2{acc0[2 * q], acc0[2 * q + 1]} + pk_set(bh[0])) + f32x2{pg[0][2 * q], pg[0][2 * q + 1]};
                    g1[q] = (f32x2{acc1[2 * q], acc1[2 * q + 1]} + pk_set(bh[1])) + f32x2{pg[1][2 * q], pg[1][2 * q + 1]};
                }
                cell_update_split(g0, g1, c < 8, cst, hlast);
            }
            {
                const int r0 = kq * 4 + (c < 8 ? 0 : 2);
                u64* dst = out_buf + r0 * H + unit;
                if constexpr (SPLIT) {
                    split_granule_store(dst, step + 1, hlast.x, same_xcd);
                    split_granule_store(dst + H, step + 1, hlast.y, same_xcd);
                } else if (!(ablate & 16)) {
                    if (same_xcd) {
                        granule_store_l2(dst, step + 1, hlast.x);
                        granule_store_l2(dst + H, step + 1, hlast.y);
                    } else {
                        granule_store(dst, step + 1, hlast.x);
                        granule_store(dst + H, step + 1, hlast.y);
                    }
                }
                hst[r0][wave * 8 + (c & 7)] = hlast.x;
                hst[r0 + 1][wave * 8 + (c & 7)] = hlast.y;
            }
            if (stamps) {
                asm volatile("" ::"v"(hlast.y));
                s5 = phase_stamp();
                if (blockIdx.x == 0 && threadIdx.x == 0 && t > 0) {   // sums live in the status area (words 8..)
                    u64* prof = reinterpret_cast<u64*>(err) + 4;
                    prof[0] += s1 - s0;   // input copy + hand-off sweep + LDS fill
                    prof[1] += s2 - s1;   // barrier
                    prof[2] += s3 - s2;   // input prefetch issue + enc_out flush + A-fragment reads + 128 MFMAs
                    prof[6] += s2a - s2;  // ... of which: input prefetch issue + enc_out flush
                    prof[3] += s4 - s3;   // input projection + barrier
                    prof[4] += s5 - s4;   // cell update + publish
                    prof[5] += 1;
                    if (t == 1 && first_tile) {           // the launch's fixed part, once: placement, weights into registers, the first step
                        prof[7] = p_placed - p_entry;
                        prof[8] = p_weights - p_placed;
                        prof[9] = s0 - p_weights;         // (step 0 and what precedes step 1's first stamp)
                    }
                }
            }
        }
        if (abort_flag) break;
        __syncthreads();                                   // last step's slice is complete in hst
        if (threadIdx.x < ROWS * 8) {
            const int row = threadIdx.x >> 3, q = threadIdx.x & 7;
            if (b0 + row < B)
                *reinterpret_cast<float4*>(enc + ((int64_t)(b0 + row) * L + (L - 1)) * H + member * UNITS + 4 * q) =
                    *reinterpret_cast<const float4*>(&hst[row][4 * q]);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int b = b0 + kq * 4 + (c < 8 ? 0 : 2) + r;
            if (b < B) {
                nets.h_n[net][(int64_t)b * H + unit] = r ? hlast.y : hlast.x;
                nets.c_n[net][(int64_t)b * H + unit] = r ? cst.y : cst.x;
            }
        }
        first_tile = false;
        ++tiles_done;
    }
    if (threadIdx.x == 0) {
        if (abort_flag) coop_raise(err, sticky, 1u, seats);
        coop_note_finished(sticky, GNNPN_STATUS_ENC_FINISHED, tiles_done);
    }
}


// ---- the exact split of a recurrent weight matrix, once per model (coop_common.h: load_split_weights).  One workgroup per member;
// every lane runs the SAME split_weights the cooperative kernels run for themselves and writes out what it leaves behind.
__global__ __launch_bounds__(256) void lstm_pack_split_kernel(const float* __restrict__ Wp, uint4* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned wts[SPLIT_WT_DWORDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, c = lane & 15;
    const int member = blockIdx.x;
    const int unit = member * UNITS + wave * 8 + (c & 7);
    f16x8 w0[2][8], w1[2][8];
    float inv[2];
    unsigned* wt_lane = wts + (wave * 8 * 64 + lane) * 4;
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) inv[tl] = split_weights<H>(Wp, 2 * tl + (c >> 3), unit, kq, w0[tl], w1[tl], wt_lane + 2 * tl);
    uint4* pk = out + (size_t)member * SPLIT_PACK_U4_PER_MEMBER;
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            pk[((tl * 2 + 0) * 8 + kk) * 256 + tid] = __builtin_bit_cast(uint4, w0[tl][kk]);
            pk[((tl * 2 + 1) * 8 + kk) * 256 + tid] = __builtin_bit_cast(uint4, w1[tl][kk]);
        }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) pk[8192 + kk * 256 + tid] = *reinterpret_cast<const uint4*>(wt_lane + 4 * 64 * kk);   // (this lane's own slot: no barrier needed)
    float* fi = reinterpret_cast<float*>(pk + 10240);
    fi[tid] = inv[0];
    fi[256 + tid] = inv[1];
}

extern "C" int64_t gnnpn_lstm_split_weights_bytes(void) { return (int64_t)SPLIT_PACK_MEMBERS * SPLIT_PACK_U4_PER_MEMBER * 16; }

extern "C" int gnnpn_lstm_pack_split_weights_f32(const float* whh_packed, void* split, void* stream) {
    GNNPN_REQUIRE(whh_packed && split, "lstm_pack_split_weights: null operand");
    GNNPN_REQUIRE(gnnpn_aligned(whh_packed, 16) && gnnpn_aligned(split, 16), "lstm_pack_split_weights: operands must be 16-byte aligned");
    static_assert(G == SPLIT_PACK_MEMBERS, "one packed block per group member");
    hipLaunchKernelGGL(lstm_pack_split_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, whh_packed, static_cast<uint4*>(split));
    GNNPN_CHECK_LAUNCH("lstm_pack_split_weights");
    return GNNPN_OK;
}

// workspace: COOP_STATUS_BYTES of status (word 0 = error; per-XCD counters on lines of their own: coop_common.h;
// stamps, hello granules), then the exchange buffers
extern "C" int64_t gnnpn_lstm_encode_workspace_bytes(void) {
    (void)gnnpn_cu_seat_table();   // callers size their workspace before the first launch and outside any capture: create the seat table here
    return COOP_STATUS_BYTES + (int64_t)64 * GROUP_GRANULES * sizeof(u64);   // up to 64 groups
}

int gnnpn_launch_encode_coop(const LstmNets& nets, int n_nets, int32_t B, int32_t L, int precision, const CoopOpts& opts,
                             void* workspace, int64_t workspace_bytes, hipStream_t s) {
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        GNNPN_FAIL(GNNPN_E_LAUNCH, "lstm_encode: cannot query the device");
    const int n_tiles = (B + ROWS - 1) / ROWS;
    // groups per XCD: every workgroup must be resident at once (one per CU), grid = 8 * G * gpx
    int gpx = n_cu / (8 * G);
    if (gpx > 8) gpx = 8;
    while (gpx > 1 && (gpx - 1) * 8 >= n_nets * n_tiles) --gpx;
    // coop_place assumes 8 XCDs with workgroup ids dealt round-robin over them (MI355X in SPX mode: 256 CUs); a partitioned
    // device (CPX / DPX / QPX: fewer XCDs) would never finish staffing
    if (gpx < 1 || n_cu < 256) GNNPN_FAIL(GNNPN_E_UNSUP, "lstm_encode: device has %d CUs, the cooperative form is built for 8 XCDs x 32 CUs", n_cu);
    const int groups = gpx * 8;
    if (groups < n_nets) GNNPN_FAIL(GNNPN_E_UNSUP, "lstm_encode: %d groups for %d nets", groups, n_nets);
    const int groups_per_net = groups / n_nets;
    const int64_t need = COOP_STATUS_BYTES + (int64_t)groups * GROUP_GRANULES * sizeof(u64);
    if (!workspace || workspace_bytes < need || !gnnpn_aligned(workspace, 256))
        GNNPN_FAIL(GNNPN_E_ARG, "lstm_encode: workspace of %lld B (256-B aligned) required", (long long)need);
    // zero the status word and every tag before each launch (tags start at 1).  Test hook (lstm_ablate bit 13, tests only): leave
    // the workspace as the previous launch left it — every workgroup must then notice (status code 8) instead of running on it
    if (coop_zero_workspace(workspace, (size_t)need, s, opts.sticky, GNNPN_STATUS_ENC_EXPECTED, (unsigned)(G * n_nets * n_tiles),
                            (gnnpn_option_lstm_ablate() & 0x2000) != 0) != hipSuccess)
        GNNPN_FAIL(GNNPN_E_LAUNCH, "lstm_encode: workspace memset failed");
    g_gnnpn_last_units = opts.sticky ? (int64_t)G * n_nets * n_tiles : 0;
    unsigned* p_seats = gnnpn_cu_seat_table();
    if (!p_seats) GNNPN_FAIL(GNNPN_E_LAUNCH, "%s: cannot allocate the seat table", "lstm_encode");
    u64* p_x = reinterpret_cast<u64*>(static_cast<char*>(workspace) + COOP_STATUS_BYTES);
    unsigned* p_e = reinterpret_cast<unsigned*>(workspace);
    bool pre = nets.pregates[0] != nullptr;
    for (int n = 0; n < n_nets; ++n)
        if ((nets.pregates[n] != nullptr) != pre)
            GNNPN_FAIL(GNNPN_E_ARG, "lstm_encode: all nets of a call must use the same input-side form");
    const int prec = precision;   // GNNPN_PREC_*: 0 fp32, 1 fp16 operands, 2 fp16-split operands
    const int abl = (gnnpn_option_lstm_ablate() & ~(0x800 | 0x1000 | 0x2000)) | (opts.write_through ? 128 : 0);   // bit 11 belongs to the decoder (phase stamps), 13 to the launch (above)
    const int abl_arg = abl | (opts.paired_start ? 0x1000 : 0);   // bit 12 rides the kernel argument only (placement, coop_place)
    const int lds_kb = opts.lds_kb;
    unsigned* p_s = opts.sticky;
    static unsigned* p_diag = gnnpn_decode_diag_buffer();   // failure record (diagnostic build, timed-out sweep only)
#define GNNPN_ENC(PREC_, PRE_, DIAG_)                                                                            \
    hipLaunchKernelGGL((lstm_encode_coop_kernel<PREC_, PRE_, DIAG_>), dim3(COOP_OVERSUB * groups * G), dim3(256),              \
                       coop_lds_padding((const void*)lstm_encode_coop_kernel<PREC_, PRE_, DIAG_>, lds_kb), s, nets, p_x, \
                       p_e, p_s, B, L, n_nets, groups_per_net, gpx, abl_arg, p_seats, p_diag)
    if ((abl & ~128) != 0) {   // diagnostic build (folded form; fp32 with every switch, exact split with the phase stamps)
        if (prec == 1 || pre || (prec == 2 && (abl & ~(128 | 32)) != 0))
            GNNPN_FAIL(GNNPN_E_UNSUP, "lstm_encode: diagnostics are built for the folded form (fp32: all switches; split: stamps only)");
        if (prec == 2) GNNPN_ENC(2, false, true);
        else GNNPN_ENC(0, false, true);
    } else if (prec == 2 && pre) GNNPN_ENC(2, true, false);
    else if (prec == 2) GNNPN_ENC(2, false, false);
    else if (prec == 1 && pre) GNNPN_ENC(1, true, false);
    else if (prec == 1) GNNPN_ENC(1, false, false);
    else if (pre) GNNPN_ENC(0, true, false);
    else GNNPN_ENC(0, false, false);
#undef GNNPN_ENC
    return GNNPN_OK;
}
