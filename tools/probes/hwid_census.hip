// Which bits of HW_REG_HW_ID identify a CU on gfx950?  256 workgroups that each need a whole CU's LDS (one per CU), each
// records XCC_ID and HW_ID; the host counts distinct (xcc, field) keys for candidate fields.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void census(unsigned* out) {
    extern __shared__ char big[];
    big[threadIdx.x] = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, 32 bits
        unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xf;
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
        long long t0 = clock64();
        while (clock64() - t0 < 2000000) {}                                    // ~1 ms: all 256 resident together
    }
    __syncthreads();
}
int main() {
    const int n = 256;
    unsigned* d;
    hipMalloc(&d, n * 8);
    hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
    hipLaunchKernelGGL(census, dim3(n), dim3(256), 160 * 1024 - 64, 0, d);
    std::vector<unsigned> h(2 * n);
    hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    for (int lo = 0; lo < 24; lo += 4) {
        for (int w = 4; w <= 12; w += 4) {
            std::set<unsigned> keys;
            for (int i = 0; i < n; ++i) keys.insert((h[2 * i + 1] << 16) | ((h[2 * i] >> lo) & ((1u << w) - 1)));
            printf("bits [%d,%d): %zu distinct (xcc, field) keys\n", lo, lo + w, keys.size());
        }
    }
    std::set<unsigned> x;
    for (int i = 0; i < n; ++i) x.insert(h[2 * i + 1]);
    printf("distinct xcc ids: %zu\n", x.size());
    for (int i = 0; i < 12; ++i) printf("block %d: hw_id %08x xcc %u\n", i, h[2 * i], h[2 * i + 1]);
    unsigned orall = 0, andall = ~0u;
    for (int i = 0; i < n; ++i) { orall |= h[2 * i]; andall &= h[2 * i]; }
    printf("OR %08x AND %08x\n", orall, andall);
    return 0;
}
