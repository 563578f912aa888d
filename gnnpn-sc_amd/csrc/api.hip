#include "common.h"
#include <string.h>
#include <mutex>
#include "lstm_shared.h"
#include "decode_shared.h"
#include "coop_common.h"

thread_local char g_gnnpn_err[256] = "";

// proof of work, host side: the workgroup-tiles the LAST recurrent entry point called on this thread booked as expected in the
// caller's status block (0: it took a streaming form, or had no status block) — so that a caller's own count of the work it asked
// for follows the launcher's choice of form instead of guessing it
thread_local int64_t g_gnnpn_last_units = 0;
extern "C" int64_t gnnpn_last_launch_units(void) { return g_gnnpn_last_units; }

// ---- LDS footprint of the ordinary kernels in front of a cooperative launch (gnnpn_lds_footprint_kb, include/gnnpn_hip.h) -----
thread_local int g_gnnpn_lds_footprint_kb = 0;
extern "C" int gnnpn_lds_footprint_kb(int kb) {
    const int prev = g_gnnpn_lds_footprint_kb;
    if (kb >= 0 && kb <= 160) g_gnnpn_lds_footprint_kb = kb;
    return prev;
}
// dynamic LDS bytes that bring `func`'s footprint to the calling thread's setting (0: none asked for, or the kernel is larger)
unsigned gnnpn_front_lds_pad(const void* func) {
    const int kb = g_gnnpn_lds_footprint_kb;
    if (kb <= 0) return 0;
    hipFuncAttributes a;
    if (hipFuncGetAttributes(&a, func) != hipSuccess) return 0;
    const long dyn = (long)kb * 1024 - (long)a.sharedSizeBytes;
    if (dyn <= 0) return 0;
    if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) return 0;
    return (unsigned)dyn;
}

extern "C" int gnnpn_abi_version(void) { return GNNPN_ABI_VERSION; }
extern "C" const char* gnnpn_last_error(void) { return g_gnnpn_err; }

// ---- diagnostics switch (tools/ only).  Implementation choice, placement control and the hand-off form are
// per-call arguments (gnnpn_launch_opts_t): nothing that selects a kernel build is process-wide state.
static int g_lstm_ablate = 0;
int gnnpn_option_lstm_ablate() { return g_lstm_ablate; }

// ---- canonical seats of the cooperative kernels (coop_common.h::coop_place): one table per device, [8 XCDs][256 CU keys]
// seat + 1 of every CU that has ever hosted a cooperative workgroup, then the 8 per-XCD counters that hand the seats out.
// Filled by the kernels themselves (first come, first seated), never reset: every later launch of the process puts the
// SAME CU in the SAME seat, so the groups of two launches that share the chip share their CUs group by group.
// Created on the first call for a device — which the workspace-size queries make, i.e. before any launch and outside any
// stream capture (allocation and synchronisation are not capturable).
unsigned* gnnpn_cu_seat_table() {
    static unsigned* table[64] = {nullptr};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!table[dev]) {
        unsigned* p = nullptr;
        if (hipMalloc(&p, COOP_SEAT_TABLE_WORDS * sizeof(unsigned)) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, COOP_SEAT_TABLE_WORDS * sizeof(unsigned)) != hipSuccess) return nullptr;
        if (hipDeviceSynchronize() != hipSuccess) return nullptr;      // before any stream's first cooperative kernel reads it
        table[dev] = p;
    }
    return table[dev];
}

// after a launch that ended in a hand-off time-out without ever being staffed: it never left the count of launches that
// are staffing (coop_common.h::coop_place), which would make every later launch decline badly placed seats for good
extern "C" int gnnpn_coop_reset_staffing(void) {
    unsigned* p = gnnpn_cu_seat_table();
    if (!p) GNNPN_FAIL(GNNPN_E_LAUNCH, "coop_reset_staffing: no seat table");
    if (hipDeviceSynchronize() != hipSuccess || hipMemset(p + COOP_STAFFING_WORD, 0, sizeof(unsigned)) != hipSuccess)
        GNNPN_FAIL(GNNPN_E_LAUNCH, "coop_reset_staffing: memset failed");
    return GNNPN_OK;
}

// diagnostics: the count itself (0 whenever no cooperative launch is between its first arrival and its last seat); -1: error
extern "C" int gnnpn_coop_staffing_count(void) {
    unsigned* p = gnnpn_cu_seat_table();
    unsigned v = 0;
    if (!p || hipDeviceSynchronize() != hipSuccess || hipMemcpy(&v, p + COOP_STAFFING_WORD, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)v;
}

extern "C" int gnnpn_set_option(const char* name, int value) {
    if (!name) GNNPN_FAIL(GNNPN_E_ARG, "set_option: null name");
    if (!strcmp(name, "lstm_ablate")) {   // bit0 no MFMA, bit1 no transcendentals, bit2 no tag wait, bit3 no sweep, bit4 no publish, bit5 stamps
        GNNPN_REQUIRE(value >= 0 && (value & 128) == 0, "set_option: lstm_ablate bit 7 is gnnpn_launch_opts_t.write_through now");
        g_lstm_ablate = value;
        return GNNPN_OK;
    }
    GNNPN_FAIL(GNNPN_E_ARG, "set_option: unknown option '%s' (impl / lds_kb / write_through are gnnpn_launch_opts_t fields)", name);
}

// ---- test hook: the cell activations on an array (tests/test_gpu_ops.py::test_cell_activations)
__global__ void cell_activations_kernel(const float* __restrict__ x, float* __restrict__ sig, float* __restrict__ th,
                                        int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) {
        sig[i] = cell_sigmoid(x[i]);
        th[i] = cell_tanh(x[i]);
    }
}

extern "C" int gnnpn_debug_cell_activations(const float* x, float* sig, float* th, int64_t n, void* stream) {
    GNNPN_REQUIRE(x && sig && th && n >= 0, "debug_cell_activations: bad argument");
    if (n == 0) return GNNPN_OK;
    hipLaunchKernelGGL(cell_activations_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, sig, th, n);
    GNNPN_CHECK_LAUNCH("debug_cell_activations");
    return GNNPN_OK;
}

// ---- a gate a stream waits behind until the HOST opens it ---------------------------------------------------------------
// One wavefront polls a 32-bit word in pinned host memory (system-scope loads, a sleep between them) until it holds `expect`
// or `timeout_us` have passed, and ends; what is enqueued behind it on the stream starts then.  PipelinedRunner holds the
// first replays of a burst behind one such gate and opens it when both are in their queues (pipeline.py: the two slots then
// start at the same moment, however long the host took to enqueue them — bounded by the time-out).
namespace {
__global__ void gate_wait_kernel(const unsigned* __restrict__ flag, unsigned expect, unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != expect &&
           __builtin_amdgcn_s_memrealtime() - t0 < ticks)
        __builtin_amdgcn_s_sleep(20);
}
}  // namespace

extern "C" int gnnpn_gate_wait(const void* flag, uint32_t expect, int32_t timeout_us, void* stream) {
    GNNPN_REQUIRE(flag && timeout_us >= 0 && timeout_us <= 100000, "gate_wait: a flag in pinned host (or device) memory, a time-out of at most 0.1 s");
    GNNPN_REQUIRE(gnnpn_aligned(flag, 4), "gate_wait: the flag must be 4-byte aligned");
    hipLaunchKernelGGL(gate_wait_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<const unsigned*>(flag), expect,
                       (unsigned long long)timeout_us * 100ull);
    GNNPN_CHECK_LAUNCH("gate_wait");
    return GNNPN_OK;
}
