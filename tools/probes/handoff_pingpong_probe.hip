// What a memory-model-valid hand-off costs per exchange, measured as what it IS in the recurrent kernels: a dependency chain
// (VERDICT r5 item 7).  Two workgroups on two CUs of ONE XCD (blockIdx b and b + 8: workgroup ids go round-robin over the XCDs;
// the XCC id is checked) play ping-pong with one 8-byte granule each: A stores tag k, B polls its sc1 load until it sees k and
// stores k into its own granule, A polls until it sees it, k + 1 ...  Half a round trip = store issued -> visible to a polling
// peer -> the peer's next instruction.  Forms of the STORE (the loads are agent-scope relaxed = sc1 in all of them):
//   plain   workgroup-scope relaxed atomic store = global_store_dwordx2, the line stays in the XCD's L2 (the default hand-off)
//   sc1     agent-scope relaxed atomic store = global_store_dwordx2 sc0 sc1, written through (gnnpn_launch_opts_t.write_through)
// and, for both, with 1 / 32 / 128 pairs playing at once.
//     hipcc --offload-arch=gfx950 -O2 -o handoff_pingpong_probe handoff_pingpong_probe.hip && ./handoff_pingpong_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

typedef unsigned long long u64;
constexpr int ITERS = 2000;

template <bool SC1>
__global__ __launch_bounds__(64) void pingpong(u64* __restrict__ slots, u64* __restrict__ cycles, unsigned* __restrict__ xcc_of, int pairs) {
    // blocks [0, 8 * pairs) are the A sides, [8 * pairs, 16 * pairs) the B sides: pair p = blocks p and p + 8 * pairs (same b % 8)
    const int n = 8 * pairs;
    const bool is_a = (int)blockIdx.x < n;
    const int p = is_a ? blockIdx.x : blockIdx.x - n;
    u64* mine = slots + (size_t)(2 * p + (is_a ? 0 : 1)) * 16;     // 128 bytes apart: one line per granule
    u64* theirs = slots + (size_t)(2 * p + (is_a ? 1 : 0)) * 16;
    if (threadIdx.x == 0) xcc_of[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xfu;
    if (threadIdx.x != 0) return;
    u64 t0 = 0, t1 = 0;
    unsigned spins = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int k = 1; k <= ITERS; ++k) {
        if (is_a) {
            if (SC1) __hip_atomic_store(mine, (u64)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(mine, (u64)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (u64)k && ++spins < 400000000u) {}
        if (!is_a) {
            if (SC1) __hip_atomic_store(mine, (u64)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(mine, (u64)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (is_a) cycles[p] = t1 - t0;
}

int main() {
    u64 *slots, *cyc;
    unsigned* xcc;
    CK(hipMalloc(&slots, (size_t)2 * 1024 * 128));
    CK(hipMalloc(&cyc, 1024 * 8));
    CK(hipMalloc(&xcc, 2048 * 4));
    static u64 h[1024];
    static unsigned hx[2048];
    auto run = [&](const char* form, auto kern, int pairs) -> int {
        CK(hipMemset(slots, 0, (size_t)2 * 1024 * 128));
        hipLaunchKernelGGL(kern, dim3(16 * pairs), dim3(64), 0, 0, slots, cyc, xcc, pairs);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, cyc, 8 * pairs * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hx, xcc, 16 * pairs * 4, hipMemcpyDeviceToHost));
        double s = 0;
        int same = 0;
        for (int i = 0; i < 8 * pairs; ++i) {
            s += (double)h[i];
            same += hx[i] == hx[i + 8 * pairs];
        }
        printf("{\"stores\": \"%s\", \"pairs_playing\": %d, \"pairs_on_one_xcd\": %d, \"cycles_per_half_round_trip\": %.0f}\n", form, 8 * pairs, same,
               s / (8 * pairs) / ITERS / 2);
        return 0;
    };
    for (int pairs : {1, 4, 16}) {           // x 8: one pair per XCD, ..., 16 pairs per XCD (256 workgroups, as many as a cooperative launch seats)
        if (run("plain (stay in the XCD's L2)", pingpong<false>, pairs)) return 2;
        if (run("sc1 (agent scope, write-through)", pingpong<true>, pairs)) return 2;
    }
    return 0;
}
