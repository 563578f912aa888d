// Where the 16 % of the write-through hand-off go (profiles/r04_handoff_form.txt: agent-scope granule stores
// cost 16.5 % at QWS against stores that stay in the XCD's L2).
// A member publishes two 8-byte granules per lane and step and then sweeps its peers' granules with eight 16-byte sc1 loads
// followed by s_waitcnt vmcnt(0).  gfx9 has ONE counter for vector loads and stores, so that wait also covers the member's own
// stores of a moment ago: a store counts until it is ACKNOWLEDGED — by the L2 for a plain store, by memory behind the L2 for a
// write-through (sc1) store.  This program measures exactly that, per form and with 1 / 32 / 256 workgroups doing it at once:
//   ack      : store, s_waitcnt vmcnt(0)                          -> cycles until the store is acknowledged
//   ack+load : store, sc1 load of an L2-resident line, vmcnt(0)   -> what the sweep's first pass waits for
//   load     : the load alone
//     hipcc --offload-arch=gfx950 -O2 -o store_ack_probe store_ack_probe.hip && ./store_ack_probe
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

template <int MODE, bool SC1>   // MODE 0 ack, 1 ack+load, 2 load, 3 (round 6) load FIRST, then the stores, s_waitcnt vmcnt(2): the wait covers the load only
__global__ __launch_bounds__(256) void probe(u64* __restrict__ slots, const u64* __restrict__ hot, u64* __restrict__ cycles) {
    u64* mine = slots + ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;   // two granules per lane, as the encoder publishes
    const u64* peer = hot + threadIdx.x * 2;                             // a line that stays in this XCD's L2 (read-only here)
    u64 sink = 0;
    // warm
    for (int i = 0; i < 50; ++i) sink += __hip_atomic_load(peer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    u64 t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < ITERS; ++i) {
        const u64 v = ((u64)(i + 1) << 32) | threadIdx.x;
        if (MODE == 3) sink += __hip_atomic_load(peer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE != 2) {
            if (SC1) {
                __hip_atomic_store(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(mine + 1, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(mine + 1, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (MODE == 1 || MODE == 2) sink += __hip_atomic_load(peer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 3) {
            // vector memory operations retire in issue order: all but the two youngest (the stores) = the load.  The NEXT iteration's
            // wait then also covers this iteration's stores, so a spacer of ~2 k clocks (a recurrent step is 6.5 k) keeps them out of it
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            for (int z = 0; z < 32; ++z) __builtin_amdgcn_s_sleep(1);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    u64 t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (sink == 0x1234567887654321ull) cycles[blockIdx.x] = sink;
}

int main() {
    u64 *slots, *hot, *cyc;
    CK(hipMalloc(&slots, (size_t)256 * 256 * 16));
    CK(hipMalloc(&hot, 256 * 16));
    CK(hipMalloc(&cyc, 256 * 8));
    CK(hipMemset(slots, 0, (size_t)256 * 256 * 16));
    CK(hipMemset(hot, 0, 256 * 16));
    u64 h[256];
    auto run = [&](const char* what, const char* form, auto kern, int wgs) {
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, slots, hot, cyc);
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, slots, hot, cyc);
        if (hipDeviceSynchronize() != hipSuccess) return;
        hipMemcpy(h, cyc, wgs * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < wgs; ++i) s += (double)h[i];
        printf("{\"what\": \"%s\", \"stores\": \"%s\", \"workgroups\": %d, \"cycles_per_iteration\": %.0f}\n", what, form, wgs, s / wgs / ITERS);
    };
    for (int wgs : {1, 32, 256}) {
        run("ack", "plain (stay in the XCD's L2)", probe<0, false>, wgs);
        run("ack", "sc1 (agent scope, write-through)", probe<0, true>, wgs);
        run("ack+load", "plain (stay in the XCD's L2)", probe<1, false>, wgs);
        run("ack+load", "sc1 (agent scope, write-through)", probe<1, true>, wgs);
        run("load", "-", probe<2, false>, wgs);
        run("load first, stores behind it, vmcnt(2) [+ 2048 clocks of s_sleep per iteration]", "plain (stay in the XCD's L2)", probe<3, false>, wgs);
        run("load first, stores behind it, vmcnt(2) [+ 2048 clocks of s_sleep per iteration]", "sc1 (agent scope, write-through)", probe<3, true>, wgs);
    }
    CK(hipGetLastError());
    return 0;
}
