// Reproducer attempt for the silent failure of round 4 (profiles/r04_handoff_timeouts_on_some_boxes.jsonl): a cooperative
// launch that began on the PREVIOUS launch's seat counters although a zeroing kernel ran between the two on the same stream.
// Hypothesis of round 4: the zeroing kernel wrote plain stores from a workgroup on XCD x (left dirty in that XCD's L2 until the
// end-of-kernel write-back), and workgroups of the next kernel on XCD y != x worked on the counters with agent-scope atomics
// and sc1 loads before / without seeing those zeros.  This program builds exactly that sequence and counts what the next
// kernel sees:
//
//   dirty : 256 workgroups (every XCD) atomicAdd 1 to each of N words and read them back with sc1 loads  -> words = 256
//   zero  : ONE workgroup, on XCD `x` (8 are launched, the one that finds itself on x writes), plain or sc1 stores of 0
//   look  : 256 workgroups; each, with or without an agent-scope acquire first, reads every word with an agent-scope atomic
//           load AND with atomicAdd(word, 0); a non-zero value is a stale observation, counted per observing XCD
//
// eager (three launches per iteration on one stream) and as a captured graph replayed; all 8 zeroing XCDs; both store forms;
// with and without the acquire.  Any count > 0 reproduces the cause; all zero = "cannot be provoked on this box".
//     hipcc --offload-arch=gfx950 -O2 -o stale_zero_probe stale_zero_probe.hip && ./stale_zero_probe [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            return 2;                                                              \
        }                                                                          \
    } while (0)

constexpr int N = 64;   // words: the status area's counters span a few cache lines

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xfu; }

__global__ void dirty(unsigned* w) {
    if (threadIdx.x < N) {
        atomicAdd(w + threadIdx.x, 1u);
        (void)__hip_atomic_load(w + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void zero(unsigned* w, unsigned want_xcc, int sc1, unsigned* ran) {
    if (xcc_id() != want_xcc) return;
    __shared__ int go;
    if (threadIdx.x == 0) go = atomicAdd(ran, 1u) == 0u;       // one workgroup of that XCD
    __syncthreads();
    if (!go) return;
    if (threadIdx.x < N) {
        if (sc1) __hip_atomic_store(w + threadIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else w[threadIdx.x] = 0u;
    }
}
__global__ void look(unsigned* w, int acquire, unsigned* stale_load, unsigned* stale_rmw) {
    if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (threadIdx.x < N) {
        const unsigned a = __hip_atomic_load(w + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned b = atomicAdd(w + threadIdx.x, 0u);
        if (a) atomicAdd(stale_load + xcc_id(), 1u);
        if (b) atomicAdd(stale_rmw + xcc_id(), 1u);
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 1000;
    unsigned *w, *ran, *sl, *sr;
    CK(hipMalloc(&w, N * 4));
    CK(hipMalloc(&ran, 4));
    CK(hipMalloc(&sl, 64));
    CK(hipMalloc(&sr, 64));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    long long total = 0;
    for (int graph = 0; graph < 2; ++graph)
        for (int sc1 = 0; sc1 < 2; ++sc1)
            for (int acq = 0; acq < 2; ++acq)
                for (unsigned x = 0; x < 8; ++x) {
                    CK(hipMemsetAsync(w, 0, N * 4, s));
                    CK(hipMemsetAsync(sl, 0, 64, s));
                    CK(hipMemsetAsync(sr, 0, 64, s));
                    CK(hipStreamSynchronize(s));
                    hipGraphExec_t exec = nullptr;
                    hipGraph_t g = nullptr;
                    auto enqueue = [&]() {
                        hipLaunchKernelGGL(dirty, dim3(256), dim3(64), 0, s, w);
                        hipMemsetAsync(ran, 0, 4, s);
                        hipLaunchKernelGGL(zero, dim3(64), dim3(64), 0, s, w, x, sc1, ran);
                        hipLaunchKernelGGL(look, dim3(256), dim3(64), 0, s, w, acq, sl, sr);
                    };
                    if (graph) {
                        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
                        enqueue();
                        CK(hipStreamEndCapture(s, &g));
                        CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
                    }
                    for (int i = 0; i < iters; ++i) {
                        if (graph) CK(hipGraphLaunch(exec, s));
                        else enqueue();
                    }
                    CK(hipStreamSynchronize(s));
                    CK(hipGetLastError());
                    unsigned hl[16], hr[16];
                    CK(hipMemcpy(hl, sl, 64, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(hr, sr, 64, hipMemcpyDeviceToHost));
                    long long nl = 0, nr = 0;
                    for (int i = 0; i < 8; ++i) nl += hl[i], nr += hr[i];
                    total += nl + nr;
                    printf("{\"mode\": \"%s\", \"zero_stores\": \"%s\", \"entry_acquire\": %d, \"zeroing_xcd\": %u, \"iterations\": %d, "
                           "\"stale_loads\": %lld, \"stale_rmw\": %lld, \"stale_loads_by_observing_xcd\": [%u,%u,%u,%u,%u,%u,%u,%u]}\n",
                           graph ? "graph" : "eager", sc1 ? "sc1" : "plain", acq, x, iters, nl, nr, hl[0], hl[1], hl[2], hl[3], hl[4], hl[5],
                           hl[6], hl[7]);
                    if (exec) hipGraphExecDestroy(exec);
                    if (g) hipGraphDestroy(g);
                }
    printf("{\"total_stale_observations\": %lld, \"verdict\": \"%s\"}\n", total,
           total ? "REPRODUCED: zeros of a zeroing kernel not seen by the next kernel's agent-scope accesses" : "not provoked on this box");
    return 0;
}
