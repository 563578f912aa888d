// What does the GRID of the tiled aggregate cost before a byte of payload moves?  (the timing-only build with
// fill, stores, stream, LDS reads and arithmetic all removed still takes 0.292 ms of 0.775 at 5000 rows x 128 copies.)
// The kernel launches 4096 workgroups of 1024 threads that each need a CU's whole LDS (160 KB): one workgroup per CU, 16 rounds,
// and a CU cannot start the next workgroup before the previous one has drained.  This program times that grid with bodies of
// increasing content:
//   empty            : nothing
//   lds              : the same with 160 KB of dynamic LDS (one workgroup per CU)
//   lds+args+header  : + a dependent chain scalar argument -> one global load (L2) -> one more dependent global load
//   lds+hbm+store    : + every lane loads 16 B from a 1.3 GB array (HBM), barrier, stores 16 B (the shape of fill -> barrier -> store)
// and a PERSISTENT form of the last body: 256 workgroups that take the 4096 items from one atomic ticket per XCD.
//     hipcc --offload-arch=gfx950 -O2 -o agg_skeleton_probe agg_skeleton_probe.hip && ./agg_skeleton_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

__global__ __launch_bounds__(1024) void k_empty(int* out) {
    if (out == nullptr && threadIdx.x == 12345) out[0] = 1;
}
__global__ __launch_bounds__(1024) void k_lds(int* out) {
    extern __shared__ int sm[];
    if (threadIdx.x == 0) sm[0] = blockIdx.x;
    __syncthreads();
    if (sm[0] == -1) out[0] = 1;
}
__global__ __launch_bounds__(1024) void k_chain(const int* __restrict__ a, int* out) {
    extern __shared__ int sm[];
    const int i = a[blockIdx.x & 1023];                 // header: L2
    const int j = a[1024 + ((i + threadIdx.x) & 1023)]; // dependent
    if (threadIdx.x == 0) sm[0] = j;
    __syncthreads();
    if (sm[0] == -1) out[0] = 1;
}
__device__ __forceinline__ void item_body(const float4* __restrict__ x, float4* __restrict__ y, int item, float4* sm) {
    const size_t base = (size_t)item * 10240;           // 160 KB per item in, 160 KB out
    float4 v[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) v[k] = x[base + k * 1024 + threadIdx.x];
#pragma unroll
    for (int k = 0; k < 10; ++k) sm[k * 1024 + threadIdx.x] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        float4 t = sm[k * 1024 + (threadIdx.x ^ 1)];
        t.x += 1.0f;
        y[base + k * 1024 + threadIdx.x] = t;
    }
}
__global__ __launch_bounds__(1024) void k_payload(const float4* __restrict__ x, float4* __restrict__ y) {
    extern __shared__ float4 smf[];
    item_body(x, y, blockIdx.x, smf);
}
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xfu; }
__global__ __launch_bounds__(1024) void k_persistent(const float4* __restrict__ x, float4* __restrict__ y, int* ticket, int per_xcd) {
    extern __shared__ float4 smf[];
    __shared__ int cur;
    const unsigned xcd = xcc_id();
    for (;;) {
        __syncthreads();                                 // previous item's LDS reads are done
        if (threadIdx.x == 0) {
            int it = -1;
            for (unsigned k = 0; k < 8 && it < 0; ++k) { // own XCD's list first, then the others' leftovers
                const unsigned xx = (xcd + k) & 7u;
                const int j = atomicAdd(ticket + xx, 1);
                if (j < per_xcd) it = (int)xx + 8 * j;
            }
            cur = it;
        }
        __syncthreads();
        const int item = cur;
        if (item < 0) break;
        item_body(x, y, item, smf);
    }
}

int main() {
    const int items = 4096, lds = 160 * 1024;
    int *a, *out, *ticket;
    float4 *x, *y;
    CK(hipMalloc(&a, 2048 * 4));
    CK(hipMemset(a, 0, 2048 * 4));
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&ticket, 64));
    CK(hipMalloc(&x, (size_t)items * 10240 * 16));
    CK(hipMalloc(&y, (size_t)items * 10240 * 16));
    CK(hipMemset(x, 0, (size_t)items * 10240 * 16));
    CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)k_payload, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)k_persistent, hipFuncAttributeMaxDynamicSharedMemorySize, lds - 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch, double bytes) {
        float best = 1e9f;
        for (int rnd = 0; rnd < 4; ++rnd) {
            hipEventRecord(e0, 0);
            for (int i = 0; i < 20; ++i) launch();
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rnd && ms / 20 < best) best = ms / 20;
        }
        printf("{\"body\": \"%s\", \"workgroups\": %d, \"ms\": %.4f, \"us_per_workgroup_round\": %.2f%s", name, items, best, best * 1e3 / (items / 256.0),
               bytes > 0 ? "" : "}\n");
        if (bytes > 0) printf(", \"GB_per_s\": %.0f}\n", bytes / best / 1e6);
        return 0;
    };
    timeit("empty (1024 threads, no LDS)", [&] { hipLaunchKernelGGL(k_empty, dim3(items), dim3(1024), 0, 0, out); }, 0);
    timeit("160 KB LDS (one workgroup per CU)", [&] { hipLaunchKernelGGL(k_lds, dim3(items), dim3(1024), lds, 0, out); }, 0);
    timeit("160 KB LDS + dependent header loads", [&] { hipLaunchKernelGGL(k_chain, dim3(items), dim3(1024), lds, 0, a, out); }, 0);
    const double bytes = 2.0 * items * 10240 * 16;
    timeit("160 KB LDS + 160 KB HBM in -> LDS -> 160 KB out", [&] { hipLaunchKernelGGL(k_payload, dim3(items), dim3(1024), lds, 0, x, y); }, bytes);
    timeit("the same payload, 256 persistent workgroups, one ticket per XCD",
           [&] {
               hipMemsetAsync(ticket, 0, 64, 0);
               hipLaunchKernelGGL(k_persistent, dim3(256), dim3(1024), lds - 64, 0, x, y, ticket, items / 8);
           },
           bytes);
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    return 0;
}
