// Helpers shared by the cooperative recurrent kernels (lstm_coop.hip, decode_coop.hip).
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

// ---- granule hand-off (CDNA4 guide, Guideline 16 / R2): 8-byte {tag, value}, ONE sc1 store / load
__device__ __forceinline__ u64 granule_load(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_load_dwordx2 sc1
}
__device__ __forceinline__ void granule_store(u64* p, unsigned tag, float v) {
    __hip_atomic_store(p, ((u64)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);                                // global_store_dwordx2 sc1
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// Two adjacent granules with one 16-byte load.  Every granule is written by ONE aligned 8-byte store, and a load of any
// width reads its cache line at one instant, so each 8-byte half is still seen whole — tag and value of a half always
// belong together (the two halves may come from different publishes, which is why each carries its own tag).
// The loads of a sweep and the wait for them are ONE asm statement: the compiler knows nothing about the asynchronous
// register writes of a hand-written load, and a spill or a copy of a destination register placed between the issue and
// the s_waitcnt would save garbage (seen: a 12-byte spill there made a wave wait for tags it had already been sent).
// Pair j of the lane is at p + 128 j granules (1 KB apart: its row quarter is 64 lanes x 16 B per pair index).
// The base address of a sweep is uniform over the wave (group, parity and wave index): it travels in SGPRs (`uniform_ptr`)
// and the lanes add a 32-bit byte offset — one VGPR instead of two 64-bit VGPR pointers, and no 64-bit vector adds per step.
#define GNNPN_LD2(dst, voff, sbase, off) "global_load_dwordx4 " dst ", " voff ", " sbase " offset:" off " sc1\n\t"
__device__ __forceinline__ const u64* uniform_ptr(const u64* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const u64*>(((unsigned long long)hi << 32) | lo);
}
// pair j of lane l: base + 16 l + 1024 j bytes (`voff` = 16 l)
__device__ __forceinline__ void granule_load2_x8(u32x4 (&v)[8], const u64* base, unsigned voff) {
    const u64* q = base + 512;                               // + 4 KB: the 13-bit immediate reaches 4095
    asm volatile(GNNPN_LD2("%0", "%8", "%9", "0") GNNPN_LD2("%1", "%8", "%9", "1024") GNNPN_LD2("%2", "%8", "%9", "2048")
                 GNNPN_LD2("%3", "%8", "%9", "3072") GNNPN_LD2("%4", "%8", "%10", "0") GNNPN_LD2("%5", "%8", "%10", "1024")
                 GNNPN_LD2("%6", "%8", "%10", "2048") GNNPN_LD2("%7", "%8", "%10", "3072")
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(voff), "s"(base), "s"(q)
                 : "memory");
}
// the same for the lanes that still miss something (call under `if (lane_is_missing)`): the others keep what they have
__device__ __forceinline__ void granule_reload2_x8(u32x4 (&v)[8], const u64* base, unsigned voff) {
    const u64* q = base + 512;
    asm volatile(GNNPN_LD2("%0", "%8", "%9", "0") GNNPN_LD2("%1", "%8", "%9", "1024") GNNPN_LD2("%2", "%8", "%9", "2048")
                 GNNPN_LD2("%3", "%8", "%9", "3072") GNNPN_LD2("%4", "%8", "%10", "0") GNNPN_LD2("%5", "%8", "%10", "1024")
                 GNNPN_LD2("%6", "%8", "%10", "2048") GNNPN_LD2("%7", "%8", "%10", "3072")
                 "s_waitcnt vmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 : "v"(voff), "s"(base), "s"(q)
                 : "memory");
}
// the same with the lane mask applied INSIDE the statement (s_and_saveexec ... restore): unconditional at the source level,
// operands tied in place — a sweep loop built on this one statement keeps its 32 registers where they are (a separate
// first-pass load, or this load under an `if`, made the register allocator copy all of them every step)
__device__ __forceinline__ void granule_reload2_x8_masked(u32x4 (&v)[8], const u64* base, unsigned voff, unsigned long long lanes) {
    const u64* q = base + 512;
    unsigned long long saved;
    asm volatile("s_and_saveexec_b64 %8, %12\n\t"
                 GNNPN_LD2("%0", "%9", "%10", "0") GNNPN_LD2("%1", "%9", "%10", "1024") GNNPN_LD2("%2", "%9", "%10", "2048")
                 GNNPN_LD2("%3", "%9", "%10", "3072") GNNPN_LD2("%4", "%9", "%11", "0") GNNPN_LD2("%5", "%9", "%11", "1024")
                 GNNPN_LD2("%6", "%9", "%11", "2048") GNNPN_LD2("%7", "%9", "%11", "3072")
                 "s_waitcnt vmcnt(0)\n\t"
                 "s_mov_b64 exec, %8"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "=&s"(saved)
                 : "v"(voff), "s"(base), "s"(q), "s"(lanes)
                 : "memory", "scc");
}
// 8 pairs at `base` (h quarter) and NP pairs at `pbase` (partial dots), one wait
__device__ __forceinline__ void granule_load2_x8_x2(u32x4 (&v)[8], u32x4 (&w)[2], const u64* base, const u64* pbase, unsigned voff) {
    const u64* q = base + 512;
    asm volatile(GNNPN_LD2("%0", "%10", "%11", "0") GNNPN_LD2("%1", "%10", "%11", "1024") GNNPN_LD2("%2", "%10", "%11", "2048")
                 GNNPN_LD2("%3", "%10", "%11", "3072") GNNPN_LD2("%4", "%10", "%12", "0") GNNPN_LD2("%5", "%10", "%12", "1024")
                 GNNPN_LD2("%6", "%10", "%12", "2048") GNNPN_LD2("%7", "%10", "%12", "3072")
                 GNNPN_LD2("%8", "%10", "%13", "0") GNNPN_LD2("%9", "%10", "%13", "1024")
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                   "=&v"(w[0]), "=&v"(w[1])
                 : "v"(voff), "s"(base), "s"(q), "s"(pbase)
                 : "memory");
}
__device__ __forceinline__ void granule_load2_x8_x4(u32x4 (&v)[8], u32x4 (&w)[4], const u64* base, const u64* pbase, unsigned voff) {
    const u64* q = base + 512;
    asm volatile(GNNPN_LD2("%0", "%12", "%13", "0") GNNPN_LD2("%1", "%12", "%13", "1024") GNNPN_LD2("%2", "%12", "%13", "2048")
                 GNNPN_LD2("%3", "%12", "%13", "3072") GNNPN_LD2("%4", "%12", "%14", "0") GNNPN_LD2("%5", "%12", "%14", "1024")
                 GNNPN_LD2("%6", "%12", "%14", "2048") GNNPN_LD2("%7", "%12", "%14", "3072")
                 GNNPN_LD2("%8", "%12", "%15", "0") GNNPN_LD2("%9", "%12", "%15", "1024") GNNPN_LD2("%10", "%12", "%15", "2048")
                 GNNPN_LD2("%11", "%12", "%15", "3072")
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                   "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
                 : "v"(voff), "s"(base), "s"(q), "s"(pbase)
                 : "memory");
}

// ... and ONE more granule per lane at `lbase` + 8 * lane (the Low net's window logits, read by the High net): in the same
// statement, so that it shares the sweep's round trip — as a separate load behind the statement's wait it cost the High net
// a second, dependent L2 round trip every step.  Lanes beyond the window read on (inside the workspace: COOP_OVERREAD_BYTES).
__device__ __forceinline__ void granule_load2_x8_x2_lat(u32x4 (&v)[8], u32x4 (&w)[2], u64& lat, const u64* base, const u64* pbase,
                                                        const u64* lbase, unsigned voff) {
    const u64* q = base + 512;
    asm volatile(GNNPN_LD2("%0", "%11", "%13", "0") GNNPN_LD2("%1", "%11", "%13", "1024") GNNPN_LD2("%2", "%11", "%13", "2048")
                 GNNPN_LD2("%3", "%11", "%13", "3072") GNNPN_LD2("%4", "%11", "%14", "0") GNNPN_LD2("%5", "%11", "%14", "1024")
                 GNNPN_LD2("%6", "%11", "%14", "2048") GNNPN_LD2("%7", "%11", "%14", "3072")
                 GNNPN_LD2("%8", "%11", "%15", "0") GNNPN_LD2("%9", "%11", "%15", "1024")
                 "global_load_dwordx2 %10, %12, %16 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                   "=&v"(w[0]), "=&v"(w[1]), "=&v"(lat)
                 : "v"(voff), "v"(voff >> 1), "s"(base), "s"(q), "s"(pbase), "s"(lbase)
                 : "memory");
}
__device__ __forceinline__ void granule_load2_x8_x4_lat(u32x4 (&v)[8], u32x4 (&w)[4], u64& lat, const u64* base, const u64* pbase,
                                                        const u64* lbase, unsigned voff) {
    const u64* q = base + 512;
    asm volatile(GNNPN_LD2("%0", "%13", "%15", "0") GNNPN_LD2("%1", "%13", "%15", "1024") GNNPN_LD2("%2", "%13", "%15", "2048")
                 GNNPN_LD2("%3", "%13", "%15", "3072") GNNPN_LD2("%4", "%13", "%16", "0") GNNPN_LD2("%5", "%13", "%16", "1024")
                 GNNPN_LD2("%6", "%13", "%16", "2048") GNNPN_LD2("%7", "%13", "%16", "3072")
                 GNNPN_LD2("%8", "%13", "%17", "0") GNNPN_LD2("%9", "%13", "%17", "1024") GNNPN_LD2("%10", "%13", "%17", "2048")
                 GNNPN_LD2("%11", "%13", "%17", "3072")
                 "global_load_dwordx2 %12, %14, %18 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]),
                   "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(lat)
                 : "v"(voff), "v"(voff >> 1), "s"(base), "s"(q), "s"(pbase), "s"(lbase)
                 : "memory");
}

// A failed bounded wait: the launch's own status word (zeroed by every launch) and the caller's sticky word
// (never cleared by the library), see gnnpn_launch_opts_t.sticky_status.
constexpr int COOP_STAFFING_WORD = 8 * 256 + 8;   // in the per-device seat table (api.hip): cooperative launches that are staffing right now
// `seats` (the per-device seat table): a launch that ends in a time-out may never take its last seat, so it leaves the
// count of staffing launches HERE (status word 4: 1 counted -> 2 left; coop_note_staffed's own 1 -> 2 then fails, so the
// count is decremented once whichever comes first).  Without this a failed launch stayed in the count until the host
// polled the status, every later launch saw "another launch is staffing" and its early arrivals kept declining their
// seats (measured with unconditional declines: 430 k -> 255-300 k problems/s) — and C-ABI / graph-replay callers never
// got the reset at all.
__device__ __forceinline__ void coop_raise(unsigned* err, unsigned* sticky, unsigned code, unsigned* seats = nullptr) {
    atomicOr(err, code);
    if (sticky) atomicOr(sticky, code);
    if (seats && atomicCAS(err + 4, 1u, 2u) == 1u) atomicSub(seats + COOP_STAFFING_WORD, 1u);
}

// ---- placement by CLAIM: one workgroup per CU, groups inside one XCD, whatever the dispatcher does ------------------
// A group's members exchange data every step, so all of them must be resident at once, on different CUs (two members
// on one CU run at half speed and every peer waits for them) and — for the L2-resident hand-off — on ONE XCD.  HIP promises
// nothing about where a workgroup lands, and a CU has room for two of these 256-register workgroups: launched on an idle
// chip, or beside another cooperative launch, the dispatcher puts two workgroups of the SAME launch on some CUs and none on
// others (round 1 steered it with LDS-footprint padding, which works only while the larger-footprint launch is already
// resident: a 56 KB launch that starts first still doubles up, and the 100 KB launch after it then waits for CUs that
// stay full for the whole run of the first — measured: bounded-spin aborts at kernels longer than the spin bound).
// So placement is ESTABLISHED at run time instead: the launch is over-subscribed (COOP_OVERSUB x the workgroups it needs);
// every workgroup reads its XCC id and the CU bits of HW_REG_HW_ID (bits [8,16): CU, SH and SE ids — 256 distinct
// (xcc, bits) keys on the 256 CUs, tools/probes/hwid_census.hip) and claims that CU for this launch with one atomic; a
// workgroup that finds its CU already claimed, or its XCD already fully staffed, exits at once and frees the slot for the
// next one.  The k-th claimant of an XCD becomes member k % G of the XCD's group k / G: groups sit on one XCD by
// construction and no CU holds two members of a launch.  Two such launches on two streams then share every CU one
// workgroup each.  The claim words live in the status area and are zeroed by the launch's own memset.
constexpr int COOP_STATUS_BYTES = 16384;     // [0,256) status + stamps, [2048,10240) CU claims, [10240,12288) seat flags, [12288,13312) per-XCD seat counters, [13312,14336) arrivals, [14336,15360) seated workgroups
constexpr unsigned COOP_LDS_UNITS = 640;    // a CU's 160 KB of LDS in the 256-byte units of HW_REG_LDS_ALLOC
// The per-XCD counters sit on cache lines of their OWN (round 6; they used to be eight adjacent words): every one of a launch's 768
// workgroups adds to its XCD's arrival counter on entry, and atomics on one line are worked off one after the other by the line's
// L2 channel (~88 per us): eight counters on one line made the arrivals of all eight XCDs one queue of 768.
constexpr int COOP_XCD_STRIDE = 32;          // words between two XCDs' counters (128 bytes)
constexpr int COOP_XCDCNT_OFFSET = 12288;    // word COOP_XCDCNT_OFFSET / 4 + COOP_XCD_STRIDE * xcd: seats taken on that XCD
constexpr int COOP_ARRIVE_OFFSET = 13312;    // likewise: workgroups of the launch that have arrived on that XCD
constexpr int COOP_PLACED_OFFSET = 14336;    // likewise: seated workgroups that run the L2-resident hand-off (statistics; used to be status word 1)
constexpr int COOP_CLAIM_OFFSET = 2048;
constexpr int COOP_TAKEN_OFFSET = 10240;     // [10240,12288) per-XCD seat flags (64 per XCD)
constexpr int COOP_OVERREAD_BYTES = 4096;    // slack at the end of the decoder workspace: fixed-shape 16-byte sweeps may read (never use) that far past their data
constexpr int COOP_OVERSUB = 3;              // launched workgroups per needed workgroup
constexpr unsigned COOP_SURPLUS_WAIT_TICKS = 1600;   // 16 us of s_memrealtime (100 MHz): how long an early surplus workgroup keeps its slot
constexpr unsigned COOP_RESERVE_WAIT_TICKS = 200000; // 2 ms: how long the reserve (the last arrivals) waits before it takes the open seats

__device__ __forceinline__ unsigned xcc_id() {
    return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xfu;   // hwreg(HW_REG_XCC_ID, 0, 4)
}

// The L2-resident publish, WRITTEN as the instruction it is (round 6): one global_store_dwordx2 without cache-policy bits — the
// granule goes to the XCD's L2, the point the group's peers read at with sc1 loads (a group sits on one XCD by placement).  Rounds
// 1-5 asked for it as a workgroup-scope relaxed atomic store and relied on this toolchain lowering that to exactly this instruction
// (build.py used to warn under another hipcc); the form is a statement about the hardware — which level a plain store lands in,
// MI355X_MICROARCH.md "stores of each flavour" — not about the HIP memory model, so the source now says which instruction it means.
// No destination register: nothing for the compiler to track (the sweeps' own asm statements wait vmcnt(0) before they read).
__device__ __forceinline__ void granule_store_l2_bits(u64* p, u64 g) {
    asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(g) : "memory");
}
__device__ __forceinline__ void granule_store_l2(u64* p, unsigned tag, float v) {
    granule_store_l2_bits(p, ((u64)tag << 32) | __float_as_uint(v));
}
__device__ __forceinline__ void granule_publish(u64* p, unsigned tag, float v, bool same_xcd) {
    if (same_xcd) granule_store_l2(p, tag, v);
    else granule_store(p, tag, v);
}

// Host side: bytes of (unused) dynamic LDS that bring `func`'s footprint to gnnpn_launch_opts_t.lds_kb (an extra placement
// constraint callers may still ask for; not needed for correctness any more).
inline unsigned coop_lds_padding(const void* func, int target_kb) {
    if (target_kb <= 0) return 0;
    hipFuncAttributes a;
    if (hipFuncGetAttributes(&a, func) != hipSuccess) return 0;
    const long dyn = (long)target_kb * 1024 - (long)a.sharedSizeBytes;
    if (dyn <= 0) return 0;
    if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) return 0;
    return (unsigned)dyn;
}

// Zeroing of the hand-off workspace before every launch (status words, claim table, every granule tag).  A kernel of our
// own rather than hipMemsetAsync: with two captured graphs replaying on two streams, the memset NODES of one graph were
// observed to fill with a 16-byte pattern made of another launch's kernel arguments (tools/soak_pipeline.py: persistent
// garbage in the status area after a few hundred steps, results wrong from then on) — the fill pattern of a captured
// memset lives in runtime-managed memory that gets recycled; a kernel node carries its arguments by value.
namespace {   // one copy per translation unit
// The zeros are written with AGENT-scope stores (sc1: write-through to the level the cooperative kernel's atomics and sc1 loads
// work on), not left dirty in the L2 of whichever XCD ran the block: the words of the status area are modified by device-scope
// atomics of the next kernel, and a failure record of round 4 showed a launch that started on seat counters of 32 + 3 and 32 + 20
// on two XCDs — the previous launch's totals, i.e. zeros that had not arrived (profiles/LOG_r01_r04.md section 13.3).
// The same kernel books the work the launch behind it is EXPECTED to do (`expected` workgroup-tiles, added to word `word` of the
// caller's status block: gnnpn_launch_opts_t.sticky_status, GNNPN_STATUS_*): every seated workgroup of the cooperative kernel adds
// the tiles it FINISHED to the word next to it when it leaves (coop_note_finished), and the host compares the two after a
// synchronisation — a launch whose workgroups all left without working (stale seat counters: section "robustness" of DESIGN.md)
// raises nothing and times nothing out, but it cannot make finished == expected.
__global__ __launch_bounds__(256) void coop_zero_kernel(unsigned long long* __restrict__ p, size_t n8, unsigned* __restrict__ sticky,
                                                        int word, unsigned expected, int skip_zeroing) {
    if (sticky && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(sticky + word, expected);
    if (skip_zeroing) return;                              // test hook (lstm_ablate bit 13): the workspace stays as the test prepared it
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x)
        __hip_atomic_store(p + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace
// bytes: a multiple of 16.  sticky / word / expected: see coop_zero_kernel (sticky may be NULL: nothing is booked).
inline hipError_t coop_zero_workspace(void* workspace, size_t bytes, hipStream_t s, unsigned* sticky, int word, unsigned expected,
                                      bool skip_zeroing = false) {
    const size_t n8 = bytes / 8;
    unsigned blocks = skip_zeroing ? 1u : (unsigned)((n8 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(coop_zero_kernel, dim3(blocks), dim3(256), 0, s, static_cast<unsigned long long*>(workspace), n8, sticky, word,
                       expected, skip_zeroing ? 1 : 0);
    return hipGetLastError();
}
// one thread of a seated workgroup, when it leaves the kernel: the tiles it took to the end (an aborted tile is not one)
__device__ __forceinline__ void coop_note_finished(unsigned* sticky, int word, unsigned tiles_done) {
    if (sticky && tiles_done) atomicAdd(sticky + word, tiles_done);
}

// Called by every thread of the workgroup.  false: surplus workgroup (leave at once).  `slot` is two ints of LDS.
//
// Seats are CANONICAL: the first cooperative launch of the process that reaches a CU writes that CU's seat (the next
// free number of its XCD) into a per-device table (`seats`, api.hip) and every later launch puts the same CU in the same
// seat.  The groups of two launches that share the chip (PipelinedRunner's two slots) then share their CUs group by
// group: all 8 members of a group see the same partner group, i.e. the same contention at the same moment, instead of
// eight partners in eight different phases — measured: the time a group waits for its slowest member falls from 6.7 %
// to 4.8 % of the two-slot step, 278.4 k -> 284.0 k problems/s (Normal B=1024: 145.4 k -> 150.8 k).
//
// Workgroup ids go round-robin over the XCDs, so each XCD receives per_xcd = gridDim.x / 8 = COOP_OVERSUB * target
// workgroups of the launch and needs `target` of them seated.  Every arrival takes an arrival index a (per XCD):
//   * an EARLY arrival (a < per_xcd - target) whose LDS allocation sits where it would leave no contiguous hole for a
//     second workgroup of the same footprint leaves at once, seat untouched (see the note at HW_REG_LDS_ALLOC below);
//   * the CU's canonical seat s is below `target` and still open -> seated (one compare-and-swap; member s % G of the
//     XCD's group s / G);
//   * otherwise it is surplus, and what it does with its CU slot decides whether the launch gets staffed:
//     - the early ones (a < per_xcd - target) keep the slot for up to 16 us, or until staffing completes, and exit: while
//       one sits there the dispatcher can only place the launch's next workgroups on CUs that still have room — the
//       unclaimed ones, as soon as whatever fills them has gone.  The bound keeps two launches that staff at the same
//       time from holding each other's slots.
//     - the LAST `target` arrivals of the XCD are the reserve and are not burned: they keep their slots until staffing
//       completes (at most one of them fits on each claimed CU, so the ones not yet dispatched stay in the dispatcher's
//       queue — at least as many as there are open seats — and go to the unclaimed CUs when those free up), and after
//       COOP_RESERVE_WAIT_TICKS (2 ms) they take the open seats themselves (lowest free seat first), on whatever CU they
//       are: a group with two members on one CU is slower, not wrong.
//   Measured need for the reserve: an ordinary kernel of the other stream whose workgroups fill a CU for longer than
//   ~50 us (the front end at 5000 candidates x 512 problems) outlasted 64 surplus workgroups of 16 us each, the launch
//   stayed one or two members short on some XCDs and its groups timed out (tools/repro_synth4.sh, record in DESIGN.md).
// one lane, right after it has taken a seat: was that the launch's last one?  Then the launch leaves the per-device
// count of launches that are staffing (status word 4: 0 not started, 1 counted, 2 staffed).  The seats are counted in ONE
// word (status word 5) so that exactly one workgroup sees the total: two last seat-takers on two XCDs comparing the
// per-XCD counters could each miss the other's increment, and the launch would stay in the count for good.
// Round 6: counted in two levels — the seat-taker adds to its XCD's counter (the one `staffed()` reads; 32-64 adds per word), the
// XCD's LAST seat-taker (its add returned target - 1) adds the XCD to status word 5, and the one that makes it 8 closes the launch:
// still exactly one workgroup sees the total, and no word takes an add from all 256 seat-takers any more.
__device__ __forceinline__ void coop_note_staffed(unsigned* status, unsigned* count_xcc, unsigned target, unsigned* seats) {
    // (an exchange, not a compare-and-swap: should the launch be complete before the first arrival of any XCD has marked it, that
    // workgroup finds 2, not 0, and adds nothing)
    if (atomicAdd(count_xcc, 1u) + 1u == target && atomicAdd(status + 5, 1u) + 1u == 8u && atomicExch(status + 4, 2u) == 1u)
        atomicSub(seats + COOP_STAFFING_WORD, 1u);
}
template <int G>
__device__ __forceinline__ bool coop_place(unsigned* status, int gpx, int* slot, int& group, int& member, unsigned* seats,
                                           bool paired_start = false, unsigned* sticky = nullptr) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const unsigned xcc = xcc_id();
        unsigned* count = status + COOP_XCDCNT_OFFSET / 4;
        unsigned* arrive = status + COOP_ARRIVE_OFFSET / 4;
        unsigned* taken = status + COOP_TAKEN_OFFSET / 4 + xcc * 64;
        const unsigned target = (unsigned)(gpx * G);
        // what a waiting surplus workgroup polls: ONE word — status word 5, the number of XCDs whose last seat has been taken
        // (coop_note_staffed) — not the eight per-XCD counters: those sit on eight lines now, and up to 500 waiting workgroups
        // re-reading eight lines each every 0.2 us is traffic in front of the very atomics the launch is waiting for (measured:
        // the every-step interferer soak lost 15-32 % with the eight-line poll against 6-13 % before the counters were spread)
        auto staffed = [&]() {
            unsigned v = 0;
            if (lane == 0) v = __hip_atomic_load(status + 5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return __shfl(v, 0, 64) >= 8u;
        };
        int g = -1, m = 0;
        // Every workgroup takes an arrival index, also the ones that find the launch staffed — and with it checks that the status
        // area was CLEAN when the launch began: a launch is staffed only after `target` arrivals on every XCD, and an XCD receives
        // gridDim.x / 8 workgroups, so "staffed" seen by one of an XCD's first `target` arrivals, or an arrival index beyond the
        // XCD's share, means counters left over from the previous launch on this workspace (the zeroing kernel's stores not seen:
        // every workgroup would then leave as surplus and the outputs would be garbage WITHOUT anyone timing out).  Code 8.
#ifndef GNNPN_COOP_NO_ENTRY_ACQUIRE
        // agent-scope acquire before the first look at the status area: what the kernel boundary behind the zeroing kernel is
        // supposed to give anyway (clean lines of another kernel's data dropped from this XCD's L2) — said explicitly, once per
        // workgroup (this wavefront), because the stale counters above are what a missing invalidate looks like
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        // ONE round trip for everything a workgroup has to learn on arrival, and no word that all 768 workgroups of a launch hammer
        // with atomics (round 6).  It used to be a chain of five dependent device-scope accesses — and every workgroup's
        // compare-and-swap on status word 4 plus every seat-taker's add to status word 5: a single word takes ~88 atomics per us
        // (MI355X_MICROARCH.md, price list: dequeue), so 768 of them alone are 9 us; the placement stamp of workgroup 0 read 25-28 k
        // cycles with the chain and, unchanged, with the chain folded into one returning atomic of twelve lanes (which made the
        // eight counter reads atomics too).  Now: one 10-lane sc1 LOAD — lanes 0..7 the eight XCDs' seat counters, lane 9 the
        // launch's mark, lane 11 this CU's canonical seat — beside one atomic add of lane 8 (the arrival index: 96 per word); the
        // launch is marked as started (status word 4: 0 -> 1, and with it the per-device count of launches that are staffing) by
        // the FIRST arrival of each XCD only (8 contenders instead of 768, the ones that already see the mark do not even try),
        // the launch's first arrival being one of them.
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
        const unsigned key = (xcc << 8) | ((hw >> 8) & 0xffu);
        unsigned* staffing = seats + COOP_STAFFING_WORD;           // per device: launches that have started and are not staffed yet
        const unsigned* word = lane < 8 ? count + COOP_XCD_STRIDE * lane : lane == 9 ? status + 4 : seats + key;
        unsigned got = 0;
        if (lane < 12 && lane != 8 && lane != 10) got = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 8) got = atomicAdd(arrive + COOP_XCD_STRIDE * xcc, 1u);
        const bool staffed_at_entry = __all(lane >= 8 || got >= target);
        unsigned arrival = __shfl(got, 8, 64);
        const bool marked = __shfl(got, 9, 64) != 0u;             // the launch was in the per-device count when this workgroup looked
        const unsigned seat_seen = __shfl(got, 11, 64);
        if (lane == 0) {
            // an XCD's first arrival puts the launch into the per-device count: mark (0 -> 1), and the ONE that wins the mark adds.
            // (Not "add, mark, take the add back if beaten", as this round first had it: eight first arrivals then hold the
            // per-device count up to eight too high for the microsecond in which most of the launch arrives, every workgroup that
            // reads it takes ANOTHER launch to be staffing, the badly placed ones decline their seats one after the other until only the
            // reserve is left, which takes seats off their CUs 2 ms later — and from the first group with two members on one CU on,
            // the partner launch finds that CU full.  Found with slot 1 held back by 400 us (tools/probes/stagger_probe.py): a third of
            // the 100-step rounds ended in a hand-off time-out; profiles/LOG_r06.md section 16.)
            if (arrival == 0u && !marked && atomicCAS(status + 4, 0u, 1u) == 0u) atomicAdd(staffing, 1u);
            if (arrival >= gridDim.x / 8 || (staffed_at_entry && arrival < target)) coop_raise(status, sticky, 8u);
        }
        if (!staffed_at_entry) {
            if (lane == 0) {
                atomicAdd(status + COOP_CLAIM_OFFSET / 4 + key, 1u);      // statistics: workgroups of this launch that reached the CU
                // Where in the CU's LDS did the dispatcher put this workgroup?  LDS is handed out in contiguous ranges, and a
                // workgroup that lands ABOVE a short-lived neighbour (an ordinary kernel's few KB) stays there when the neighbour
                // has gone: a 78 KB footprint at, say, [25 KB, 103 KB) leaves two holes of 25 and 57 KB, and the partner launch's
                // workgroup for this CU — same footprint — cannot be placed until this whole kernel has finished.  Two launches
                // that start together can each hold a few CUs that way: both stay under-staffed, both wait, and the bounded spins
                // end it 0.46 s later (seen: two slots started together at the 1000- and 2000-task shapes, 10 % of the steps;
                // failure record: 33 of 96 workgroups of an XCD dispatched, 29 of 32 seats).  HW_REG_LDS_ALLOC holds base [11:0]
                // and size [23:12] of the allocation in 256-byte units (tools/probes/lds_alloc_probe.hip).  An early arrival that
                // would fragment the CU does not take the seat and leaves at once — a later one finds the neighbour gone and
                // starts at 0; the reserve (the last arrivals) takes the seat wherever it is, so staffing still always completes.
                // All of this ONLY while another cooperative launch of the process is staffing too (a per-device count of
                // launches between their first arrival and their last seat): a badly placed workgroup of the ONLY launch that
                // is short of members can at worst delay a launch that arrives later — which is complete, runs and ends — and
                // declining there costs dearly where ordinary kernels hold LDS for long (the RCCL all-gather of the multi-GPU
                // path: 430 k -> 255-300 k problems/s with unconditional declines).
                const unsigned la = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11));     // HW_REG_LDS_ALLOC
                const unsigned lds_base = la & 0xfffu, lds_size = (la >> 12) & 0xfffu;
                // strict (every badly placed early arrival leaves) where the caller says that a partner launch starts TOGETHER with
                // this one (gnnpn_launch_opts_t.paired_start: half-batches, slots started in pairs — nothing but the tails of the
                // front halves is around then); otherwise only while another launch is staffing at this moment
                const bool badly_placed = lds_base != 0u && lds_base < lds_size && lds_base + 2u * lds_size > COOP_LDS_UNITS;
                // "another launch is staffing" is asked only by the few workgroups it decides something for, and exactly, as rounds
                // 3-5 asked it: this launch is marked first (whoever wins the mark adds it to the count), then the count is read
                // afresh — more than this launch's own entry means another one.  (The snapshot every workgroup takes on arrival is
                // too early for this question: see the note at the mark above.)
                bool another_staffing = false;
                if (badly_placed && !paired_start) {
                    if (atomicCAS(status + 4, 0u, 1u) == 0u) atomicAdd(staffing, 1u);
                    another_staffing = __hip_atomic_load(staffing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 1u;
                }
                const bool fragments = (paired_start || another_staffing) && badly_placed;
                const bool reserve_arrival = arrival + target >= gridDim.x / 8;
                if (fragments && !reserve_arrival) {
                    atomicAdd(status + 3, 1u);                                // statistics: arrivals that left because of their LDS position
                    if (sticky) atomicAdd(sticky + GNNPN_STATUS_DECLINED_SEATS, 1u);   // ... cumulative over the caller's launches (ABI 9)
                    g = -2;                                                   // leave without holding the slot; the seat stays open
                } else if (!fragments) {                         // (a badly placed RESERVE arrival does not rush for the seat either: it waits
                    //                                              with the reserve below and takes an open seat only if nobody better placed has)
                    // the CU's canonical seat (process-wide table: the same CU sits in the same seat in every launch); whoever of
                    // this launch gets there first takes it (one compare-and-swap), every later arrival on the CU is surplus
                    unsigned s = seat_seen;
                    if (s == 0u) {
                        if (atomicCAS(seats + key, 0u, 0xffffffffu) == 0u) {
                            s = atomicAdd(seats + 8 * 256 + xcc, 1u) + 1u;
                            __hip_atomic_store(seats + key, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else {
                            s = 0xffffffffu;
                        }
                    }
                    for (unsigned spin = 0; s == 0xffffffffu && spin < 100000u; ++spin)     // another launch is writing it
                        s = __hip_atomic_load(seats + key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned seat = s - 1u;
                    if (seat < target && atomicCAS(taken + seat, 0u, 1u) == 0u) {
                        g = (int)xcc * gpx + (int)(seat / G);
                        m = (int)(seat % G);
                        coop_note_staffed(status, count + COOP_XCD_STRIDE * xcc, target, seats);
                    }
                }
            }
            g = __shfl(g, 0, 64);
            m = __shfl(m, 0, 64);
            arrival = __shfl(arrival, 0, 64);
            if (g == -2) {                                       // declined its seat: the slot (and its LDS range) is freed at once
                g = -1;
            } else if (g < 0) {                                  // surplus (see above)
                const unsigned per_xcd = gridDim.x / 8;
                const bool reserve = arrival + target >= per_xcd;
                const unsigned long long patience = reserve ? COOP_RESERVE_WAIT_TICKS : COOP_SURPLUS_WAIT_TICKS;
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                bool done = staffed();
                while (!done && __builtin_amdgcn_s_memrealtime() - t0 < patience) {
                    __builtin_amdgcn_s_sleep(8);
                    done = staffed();
                }
                if (!done && reserve) {                          // take an open seat, whichever CU this is
                    if (lane == 0) {
                        for (unsigned seat = 0; seat < target && g < 0; ++seat) {
                            if (atomicCAS(taken + seat, 0u, 1u) == 0u) {
                                g = (int)xcc * gpx + (int)(seat / G);
                                m = (int)(seat % G);
                                coop_note_staffed(status, count + COOP_XCD_STRIDE * xcc, target, seats);
                                atomicAdd(status + 2, 1u);       // statistics: seats taken off the canonical CU
                                if (sticky) atomicAdd(sticky + GNNPN_STATUS_OFF_CANONICAL_SEATS, 1u);
                            }
                        }
                    }
                    g = __shfl(g, 0, 64);
                    m = __shfl(m, 0, 64);
                }
            }
        }
        if (lane == 0) {
            slot[0] = g;
            slot[1] = m;
        }
    }
    __syncthreads();
    group = slot[0];
    member = slot[1];
    return group >= 0;
}

// value of lane (l ^ 8) within each row of 16 lanes, as a DPP row rotate (no LDS round trip)
__device__ __forceinline__ float swap8(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /* row_ror:8 */, 0xF, 0xF, false));
}

// Diagnostic stamps (diagnostic option only; never in a measured run)
__device__ __forceinline__ u64 phase_stamp() {
    u64 t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

// fp32 h tile in k-quarter-major order: unit u = 4*kk + kq of row r sits at r*LDT + kq*64 + kk, so the A-fragments of lane
// (c, kq) for the k-steps kk = 0..63 are 64 CONSECUTIVE floats: 16 ds_read_b128 per chain pass instead of 64 ds_read_b32
// (their issue cost sits in the MFMA phase — fp32 MFMAs share the SIMD's datapath with every other instruction the wave
// issues, tools/probes/mfma_valu_coissue.hip).  LDT = 260: the 16 lanes of a ds_read_b128 group hold 16 different rows c,
// whose 4-float reads start 4*c banks apart — conflict-free; a sweep's two stores per granule pair (units u, u+1: 64 floats
// apart) are 2-way on 32 banks, which costs a ds_write_b32 nothing.
constexpr int LDT = 260;
__device__ __forceinline__ int ht_index(int row, int unit) { return row * LDT + (unit & 3) * 64 + (unit >> 2); }

// Two k-ordered fp32 MFMA chains (one per 16-column tile) over K = 256 against A-fragments read from
// an LDS tile `src`: row-major (row c, stride LD, element 4*kk + kq) or, KQ, the k-quarter-major h tile above.
// The A-fragments are fetched CH k-steps ahead of the MFMAs that use them: left to itself hipcc issues each
// ds_read right before the MFMAs that need it and waits ~70 cycles per 4 MFMAs (measured 6.3k instead of 4.1k
// cycles per step).
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
// `mid` runs once, half-way through the chain (the decoder requests the partial-dot granules there: their round trip then
// lies under the second half of the products)
template <int LD, int CH = 16, bool KQ = false, typename Mid = NoHook>   // CH: prefetch depth in k-steps (8 where registers are short: 2 x CH fragment registers)
__device__ __forceinline__ void mfma_chain_pair(const float* src, int c, int kq, const float (&w0)[64],
                                                const float (&w1)[64], f32x4& acc0, f32x4& acc1, Mid mid = Mid()) {
    const float* base = KQ ? src + c * LD + kq * 64 : src + c * LD + kq;
    const float4* base4 = reinterpret_cast<const float4*>(base);       // KQ: 16-byte aligned (LD * 4 and kq * 256 are)
    float a[2][CH];
    auto fetch = [&](int ch, float (&dst)[CH]) {
        if constexpr (KQ) {
#pragma unroll
            for (int q = 0; q < CH / 4; ++q) {
                const float4 v = base4[(CH / 4) * ch + q];
                dst[4 * q] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < CH; ++i) dst[i] = base[4 * (CH * ch + i)];
        }
    };
    fetch(0, a[0]);
#pragma unroll
    for (int ch = 0; ch < 64 / CH; ++ch) {
        if (ch < 64 / CH - 1) fetch(ch + 1, a[(ch + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ch & 1][i], w0[CH * ch + i], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ch & 1][i], w1[CH * ch + i], acc1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ch == 64 / CH / 2 - 1) mid();
    }
}

// ---- exact-split recurrent product ("split" precision, rebuilt in round 3) -------------------------------------------
// Every fp32 operand x of W_hh.h is decomposed into THREE fp16 pieces that reproduce it bit for bit,
//     x * 2^s = p0 + p1 / 2^11 + p2 / 2^22,     p0 = fp16(x 2^s), p1 = fp16((x 2^s - p0) 2^11), p2 = fp16(((x 2^s - p0) 2^11 - p1) 2^11)
// (each residual is exact in fp32; p2 is the <= 1 bit the first two 11-bit significands cannot hold), s a power-of-two
// scale that parks the operand high in fp16's range: 2^15 for h (|h| <= 1), per gate column for W (column maximum in
// [2^14, 2^15)) — exact for 2^-38 <= |x| / max <= 1 and for 0, to 2^-62 * max absolutely below that (tests/test_split3.py).
// The product keeps every cross term that can reach 2^-24 of |x y|:
//     x y 2^(s+t) = p0 q0 + (p0 q1 + p1 q0) / 2^11 + (p1 q1 + p0 q2 + p2 q0) / 2^22  [+ terms <= 2^-32 |x y|, dropped]
// = 6 v_mfma_f32_16x16x32_f16 per 32 k-steps and column tile instead of 16 v_mfma_f32_16x16x4_f32: 96 x 16 cycles per
// recurrent step against 128 x 32.  The three magnitude classes accumulate in three SEPARATE fp32 accumulators (the matrix
// core aligns the products of a group to its largest term and drops what falls below 2^-24 of it — measured,
// tools/probes/mfma_accum_model.hip — so small terms must not share an accumulator with large ones) and are combined once,
// by two fused multiply-adds.  Error bound and measurements: DESIGN.md section 5; profiles/LOG_r01_r04.md section 12.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int LDH16 = 264;                 // row stride in halfs (528 B: 16-B aligned, bank slots spread)
constexpr int SPLIT_TILE = 16 * LDH16;     // halfs per piece tile; the LDS h tile is [3 pieces][16 rows][LDH16]
constexpr float SPLIT_SCALE = 2048.0f, SPLIT_INV = 1.0f / 2048.0f;
constexpr float SPLIT_H_SCALE = 32768.0f, SPLIT_H_INV = 1.0f / 32768.0f;   // |h| <= 1 -> |h 2^15| <= 2^15 < 65504

__device__ __forceinline__ void split3(float xs, _Float16& p0, _Float16& p1, _Float16& p2) {   // xs: the SCALED operand
    p0 = (_Float16)xs;
    const float r1 = __fmul_rn(__fsub_rn(xs, (float)p0), SPLIT_SCALE);      // exact: residual of an 11-bit rounding, times 2^11
    p1 = (_Float16)r1;
    p2 = (_Float16)__fmul_rn(__fsub_rn(r1, (float)p1), SPLIT_SCALE);        // exact, and representable: <= 1 significant bit
}
__device__ __forceinline__ unsigned short f16_bits(_Float16 v) { return __builtin_bit_cast(unsigned short, v); }

// h granule of the split precision: the publisher splits its own value ONCE and every receiver only unpacks (the fp32
// value is never needed by a receiver) — {p0 | tag16 << 16, p1 | p2 << 16}.  tag16 = 0x8000 | (tag & 0x7fff): never 0 (a
// zeroed granule is "empty"), and a parity buffer's previous content is two publishes old, so 15 bits cannot alias.
__device__ __forceinline__ unsigned split_tag16(unsigned tag) { return 0x8000u | (tag & 0x7fffu); }
__device__ __forceinline__ u64 split_granule(unsigned tag, float h) {
    _Float16 p0, p1, p2;
    split3(__fmul_rn(h, SPLIT_H_SCALE), p0, p1, p2);
    const unsigned d0 = (unsigned)f16_bits(p0) | (split_tag16(tag) << 16);
    const unsigned d1 = (unsigned)f16_bits(p1) | ((unsigned)f16_bits(p2) << 16);
    return ((u64)d1 << 32) | d0;
}
__device__ __forceinline__ void split_granule_store(u64* p, unsigned tag, float h, bool same_xcd) {
    const u64 g = split_granule(tag, h);
    if (same_xcd) granule_store_l2_bits(p, g);
    else __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// both granules of a 16-byte pair {A.d0, A.d1, B.d0, B.d1} carry `tag`?
__device__ __forceinline__ bool split_pair_tagged(u32x4 v, unsigned tag) {
    const unsigned t = split_tag16(tag);
    return __builtin_amdgcn_perm(v.z, v.x, 0x07060302u) == (t | (t << 16));
}
// unpack a pair (units u, u+1 of one row; u even) into the three piece tiles: one packed 32-bit LDS store per piece
__device__ __forceinline__ void split_pair_to_lds(_Float16* at, u32x4 v) {   // at: element (row, u) of the p0 tile
    unsigned* d = reinterpret_cast<unsigned*>(at);
    d[0] = __builtin_amdgcn_perm(v.z, v.x, 0x05040100u);                     // p0(u) | p0(u+1) << 16
    d[SPLIT_TILE / 2] = __builtin_amdgcn_perm(v.w, v.y, 0x05040100u);        // p1
    d[SPLIT_TILE] = __builtin_amdgcn_perm(v.w, v.y, 0x07060302u);            // p2
}
// one fp32 value (the tile's initial state h0) into the three tiles
__device__ __forceinline__ void split_store(_Float16* at, float h) {        // at: element address inside the p0 tile
    _Float16 p0, p1, p2;
    split3(__fmul_rn(h, SPLIT_H_SCALE), p0, p1, p2);
    at[0] = p0;
    at[SPLIT_TILE] = p1;
    at[2 * SPLIT_TILE] = p2;
}

// B-fragments of one 16-column tile from the packed weight layout Wp[((k/4*4 + gate)*H + u)*4 + k%4]: lane (c, kq) holds
// W[col c][32kk + 8kq + j], j = 0..7, of ITS column, scaled by 2^s with s chosen from the column's largest |w| (the four
// lanes c, c+16, c+32, c+48 hold the column between them).  Pieces 0 and 1 stay in registers (64 VGPRs per tile each).
// Piece 2 is zero or a single power of two (the one bit two 11-bit significands cannot hold), so its upper byte IS the
// fp16 (= an e5m2 number): it lives in LDS as bytes — 192 weight registers would not leave room for two workgroups per CU
// (measured: 136 B of scratch in the encoder, 450 B in the decoder) — `wt` = this lane's slot of the tile's byte image,
// two dwords (elements 0..3, 4..7) per k-block kk at wt[4 * 64 * kk] (see split_chain).  An element more than 2^29 below its
// column's largest (p0 subnormal in fp16) would need more than that byte: its third piece is truncated (|error| < 2^-52
// of the column maximum; tests/test_split3.py walks that boundary).
// Returns the factor 2^-(15+s) that un-scales an accumulated product of this column (the 2^15 of h included).
constexpr int SPLIT_WT_DWORDS = 4 * 8 * 64 * 4;   // per workgroup: [wave][kk][lane]{tile0 lo, tile0 hi, tile1 lo, tile1 hi}
template <int HD>
__device__ __forceinline__ float split_weights(const float* __restrict__ Wp, int gate, int u, int kq, f16x8 (&w0)[8],
                                               f16x8 (&w1)[8], unsigned* wt) {
    float m = 0.0f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            m = fmaxf(m, fabsf(Wp[((size_t)((8 * kk + 2 * kq + (j >> 2)) * 4 + gate) * HD + u) * 4 + (j & 3)]));
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    int e = 0;
    if (m > 0.0f && m < 3.0e38f) (void)frexpf(m, &e);                        // m = f 2^e, f in [0.5, 1): m < 2^e
    int s = m > 0.0f ? 15 - e : 0;                                           // m 2^s in [2^14, 2^15)
    s = s > 96 ? 96 : (s < -96 ? -96 : s);                                   // 2^-(15+s) stays a normal fp32
    const float up = ldexpf(1.0f, s);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        unsigned bytes[2] = {0u, 0u};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float w = Wp[((size_t)((8 * kk + 2 * kq + (j >> 2)) * 4 + gate) * HD + u) * 4 + (j & 3)];
            _Float16 p0, p1, p2;
            split3(__fmul_rn(w, up), p0, p1, p2);
            w0[kk][j] = p0;
            w1[kk][j] = p1;
            bytes[j >> 2] |= (unsigned)(f16_bits(p2) >> 8) << (8 * (j & 3));
        }
        wt[4 * 64 * kk] = bytes[0];
        wt[4 * 64 * kk + 1] = bytes[1];
    }
    return ldexpf(1.0f, -(15 + s));
}
// ---- the same split, made ONCE per model (round 6: gnnpn_lstm_pack_split_weights_f32).  Splitting inside the kernel costs every
// cooperative launch 9-13 us of its fixed part (profiles/LOG_r05.md section 10: each lane reads its 128 weights twice — column maximum, then the
// pieces: ~2 k vector instructions); the packed image holds exactly what split_weights leaves in a lane's registers and LDS slot, in
// the order the lanes load it (64 lanes x 16 B contiguous per load instruction).  Per member m (8 per weight matrix), in uint4:
//   [((tile * 2 + piece) * 8 + kk) * 256 + tid]      pieces 0 / 1 of k-block kk, tid = 64 * wave + lane          (8192 entries)
//   [8192 + kk * 256 + tid]                          the third pieces' bytes {tile 0 lo, hi, tile 1 lo, hi}      (2048 entries)
//   [10240 ...] as float[tile * 256 + tid]           the columns' un-scaling factors                             (128 entries)
// Made by split_weights itself (lstm_pack_split_kernel, lstm_coop.hip): bit-identical to splitting in the kernel by construction.
constexpr int SPLIT_PACK_U4_PER_MEMBER = 2 * 2 * 8 * 256 + 8 * 256 + 2 * 256 / 4;   // 10368 x 16 B = 165,888 B
constexpr int SPLIT_PACK_MEMBERS = 8;
__device__ __forceinline__ void load_split_weights(const uint4* __restrict__ pk, int tid, f16x8 (&w0)[2][8], f16x8 (&w1)[2][8],
                                                   unsigned* wt_lane, float (&inv)[2]) {
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            w0[tl][kk] = __builtin_bit_cast(f16x8, pk[((tl * 2 + 0) * 8 + kk) * 256 + tid]);
            w1[tl][kk] = __builtin_bit_cast(f16x8, pk[((tl * 2 + 1) * 8 + kk) * 256 + tid]);
        }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) *reinterpret_cast<uint4*>(wt_lane + 4 * 64 * kk) = pk[8192 + kk * 256 + tid];
    const float* fi = reinterpret_cast<const float*>(pk + 10240);
    inv[0] = fi[tid];
    inv[1] = fi[256 + tid];
}

// four third-piece bytes -> two packed fp16 pairs (byte b becomes the half b << 8)
__device__ __forceinline__ f16x8 split_expand(unsigned lo, unsigned hi) {
    const u32x4 r = {__builtin_amdgcn_perm(0u, lo, 0x010c000cu), __builtin_amdgcn_perm(0u, lo, 0x030c020cu),
                     __builtin_amdgcn_perm(0u, hi, 0x010c000cu), __builtin_amdgcn_perm(0u, hi, 0x030c020cu)};
    return __builtin_bit_cast(f16x8, r);
}
// base = p0 tile + c * LDH16 + 8 * kq; wt = this lane's third-piece slot (uint4 per k-block: both tiles).  Two column tiles
// sharing the A-fragments; acc[n] receives W.h of tile n (un-scaled by inv[n]).
template <typename Mid = NoHook>
__device__ __forceinline__ void split_chain(const _Float16* base, const f16x8 (&w0)[2][8], const f16x8 (&w1)[2][8],
                                            const unsigned* wt, const float (&inv)[2], f32x4 (&acc)[2], Mid mid = Mid()) {
    constexpr int NT = 2;
    // a0 = p0.q0 in TWO accumulators (k-blocks 0..3 / 4..7): the matrix core works through a 16x16x32 product in four groups
    // of 8 k's, aligning a group's 8 products and the accumulator to the largest of them, dropping what falls below 2^-24 of
    // it and rounding once (tools/probes/mfma_accum_model.hip) — 10 error events of <= 2^-24 of the running magnitude per
    // group.  16 groups per accumulator keep the worst-case constant at 10 x 16 + 2 = 162 units of 2^-24 sum|h w|, below the
    // 256 of a 256-term fp32 fma chain (32 groups in one accumulator would be 320); profiles/LOG_r01_r04.md section 12.
    f32x4 a0[NT], a0b[NT], a1[NT], a2[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) a0[n] = a0b[n] = a1[n] = a2[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Fragment schedule (registers are short: two workgroups per CU).  Per k-block the twelve products run in the order
    //   h1.w1 -> a2 | h1.w0 -> a1 | h0.w2 -> a2 | h0.w0 -> a0 | h2.w0 -> a2 | h0.w1 -> a1        (each for both tiles)
    // one per line below, every line ending in a scheduling fence.  A 16x16x32 product occupies the matrix pipe for 16
    // cycles and the SIMD's vector issue for 8 of them (MI355X_MICROARCH.md, cycle constants): the eight v_perm that expand
    // the block's third-piece bytes are dealt two per gap to the first four gaps, and every single-buffered piece (h1, h2,
    // the bytes) is re-requested in the gap behind its last use, >= 7 products (112 cycles) ahead of its next one; h0, used
    // three times per block, is the one double-buffered piece.  Left alone the compiler sinks every LDS read down to its
    // first use (fewest live registers) and the wave waits out a full LDS latency several times per block — the first build
    // of this function spent 3.5 k cycles on 1.5 k cycles of MFMAs (tools/stamp_decode.py).  Measured: the bare chain runs
    // at 17.4-18.1 ticks per product (tools/probes/mfma_chain_rate.hip: registers only / with these LDS reads and v_perm);
    // inside the kernels the phase takes 2.4 k ticks for 96 products.  Dealing the v_perm out instead of one burst of eight
    // per block changed nothing for a lone workgroup (2440 vs 2420) and gave +2 % on the two-slot pipeline (the partner
    // wave's vector instructions find the issue slots).
    f16x8 h0[2];
    h0[0] = *reinterpret_cast<const f16x8*>(base);
    f16x8 h1 = *reinterpret_cast<const f16x8*>(base + SPLIT_TILE);
    f16x8 h2 = *reinterpret_cast<const f16x8*>(base + 2 * SPLIT_TILE);
    u32x4 t = *reinterpret_cast<const u32x4*>(wt);
#define GNNPN_FENCE __builtin_amdgcn_sched_barrier(0)
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int nx = 32 * (kk + 1), cur = kk & 1;
        u32x4 e0, e1;                                                       // the expanded third pieces of tiles 0 and 1
        if (kk < 7) h0[cur ^ 1] = *reinterpret_cast<const f16x8*>(base + nx);
        GNNPN_FENCE;
        a2[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, w1[0][kk], a2[0], 0, 0, 0);
        e0.x = __builtin_amdgcn_perm(0u, t.x, 0x010c000cu);
        e0.y = __builtin_amdgcn_perm(0u, t.x, 0x030c020cu);
        GNNPN_FENCE;
        a2[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, w1[1][kk], a2[1], 0, 0, 0);
        e0.z = __builtin_amdgcn_perm(0u, t.y, 0x010c000cu);
        e0.w = __builtin_amdgcn_perm(0u, t.y, 0x030c020cu);
        GNNPN_FENCE;
        a1[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, w0[0][kk], a1[0], 0, 0, 0);
        e1.x = __builtin_amdgcn_perm(0u, t.z, 0x010c000cu);
        e1.y = __builtin_amdgcn_perm(0u, t.z, 0x030c020cu);
        GNNPN_FENCE;
        a1[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, w0[1][kk], a1[1], 0, 0, 0);
        e1.z = __builtin_amdgcn_perm(0u, t.w, 0x010c000cu);
        e1.w = __builtin_amdgcn_perm(0u, t.w, 0x030c020cu);
        GNNPN_FENCE;
        if (kk < 7) {
            h1 = *reinterpret_cast<const f16x8*>(base + SPLIT_TILE + nx);
            t = *reinterpret_cast<const u32x4*>(wt + 4 * 64 * (kk + 1));
        }
        a2[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0[cur], __builtin_bit_cast(f16x8, e0), a2[0], 0, 0, 0);
        GNNPN_FENCE;
        a2[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0[cur], __builtin_bit_cast(f16x8, e1), a2[1], 0, 0, 0);
        GNNPN_FENCE;
        if (kk < 4) a0[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0[cur], w0[0][kk], a0[0], 0, 0, 0);
        else a0b[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0[cur], w0[0][kk], a0b[0], 0, 0, 0);
        GNNPN_FENCE;
        if (kk < 4) a0[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0[cur], w0[1][kk], a0[1], 0, 0, 0);
        else a0b[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0[cur], w0[1][kk], a0b[1], 0, 0, 0);
        GNNPN_FENCE;
        a2[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h2, w0[0][kk], a2[0], 0, 0, 0);
        GNNPN_FENCE;
        a2[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h2, w0[1][kk], a2[1], 0, 0, 0);
        GNNPN_FENCE;
        if (kk < 7) h2 = *reinterpret_cast<const f16x8*>(base + 2 * SPLIT_TILE + nx);
        a1[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0[cur], w1[0][kk], a1[0], 0, 0, 0);
        GNNPN_FENCE;
        a1[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0[cur], w1[1][kk], a1[1], 0, 0, 0);
        GNNPN_FENCE;
        if (kk == 3) mid();
    }
#undef GNNPN_FENCE
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[n][r] = __fmul_rn(fmaf(fmaf(a2[n][r], SPLIT_INV, a1[n][r]), SPLIT_INV, __fadd_rn(a0[n][r], a0b[n][r])), inv[n]);
}

// LSTM cell update for the [i | f] / [g | o] tile pair: lanes c < 8 hold (i, g), lanes c >= 8 hold
// (f, o) of the same hidden unit; one DPP swap per value, then both halves update (c, h) alike.
// Same arithmetic as lstm_cell_update (recurrent.h): cy = f*c + i*g ; hy = o*tanh(cy), products and
// sum rounded separately.
// two problem rows at once (packed arithmetic, see cell_act2)
__device__ __forceinline__ void cell_update_pair2(f32x2 g0, f32x2 g1, bool lo_half, f32x2& cst, f32x2& h) {
    const f32x2 a0 = cell_act2(g0, false);
    const f32x2 a1 = cell_act2(g1, lo_half);
    const f32x2 p0 = {swap8(a0.x), swap8(a0.y)}, p1 = {swap8(a1.x), swap8(a1.y)};
    const f32x2 ig = lo_half ? a0 : p0, gg = lo_half ? a1 : p1;
    const f32x2 fg = lo_half ? p0 : a0, og = lo_half ? p1 : a1;
    cst = fg * cst + ig * gg;          // -ffp-contract=off: two rounded products, one rounded sum
    h = og * cell_act2(cst, true);
}
// The same cell update without the duplicated half: of a lane's four rows the lower half-row (lanes c < 8, holding i and g)
// finishes rows 0,1 and the upper one (c >= 8, holding f and o) rows 2,3.  g0 / g1: the lane's gate pre-activations of
// rows 0..3 as two packed pairs each.  One DPP exchange per value: a lane sends the gate pair of the rows its PARTNER
// finishes and receives the partner's gates of its own rows.  Element for element the arithmetic of cell_update_pair2 —
// five packed activations per step instead of six, two state registers per lane instead of four.
__device__ __forceinline__ void cell_update_split(const f32x2 (&g0)[2], const f32x2 (&g1)[2], bool lo_half, f32x2& cst, f32x2& h) {
    f32x2 a0[2], a1[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        a0[q] = cell_act2(g0[q], false);          // sigmoid(i) | sigmoid(f)   rows 2q, 2q+1
        a1[q] = cell_act2(g1[q], lo_half);        // tanh(g)    | sigmoid(o)
    }
    const f32x2 s0 = lo_half ? a0[1] : a0[0], s1 = lo_half ? a1[1] : a1[0];     // what the partner needs
    const f32x2 t0 = {swap8(s0.x), swap8(s0.y)}, t1 = {swap8(s1.x), swap8(s1.y)};
    const f32x2 ig = lo_half ? a0[0] : t0, gg = lo_half ? a1[0] : t1;           // own rows: 0,1 (lower) / 2,3 (upper)
    const f32x2 fg = lo_half ? t0 : a0[1], og = lo_half ? t1 : a1[1];
    cst = fg * cst + ig * gg;
    h = og * cell_act2(cst, true);
}
__device__ __forceinline__ void cell_update_pair(float g0, float g1, bool lo_half, float& cst, float& h) {
    const float a0 = cell_act(g0, false);            // sigmoid(i) | sigmoid(f)
    const float a1 = cell_act(g1, lo_half);          // tanh(g)    | sigmoid(o)
    const float p0 = swap8(a0), p1 = swap8(a1);
    const float ig = lo_half ? a0 : p0, gg = lo_half ? a1 : p1;
    const float fg = lo_half ? p0 : a0, og = lo_half ? p1 : a1;
    cst = __fadd_rn(__fmul_rn(fg, cst), __fmul_rn(ig, gg));
    h = __fmul_rn(og, cell_act(cst, true));
}
