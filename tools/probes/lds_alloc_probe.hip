// What does HW_REG_LDS_ALLOC (s_getreg id 6) say about a workgroup's LDS allocation on gfx950 (160 KB per CU)?
// Launches workgroups with several dynamic LDS sizes (two can share a CU) and prints the raw register next to the size.
// build: hipcc --offload-arch=gfx950 -O2 -o lds_alloc_probe lds_alloc_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out, int spin) {
    extern __shared__ unsigned char lds[];
    const unsigned la = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (31 << 11));     // HW_REG_LDS_ALLOC, all 32 bits
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lds[threadIdx.x] = (unsigned char)threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);   // stay resident so that others share the CU
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = la;
        out[blockIdx.x * 4 + 1] = ((xcc & 0xf) << 8) | ((hw >> 8) & 0xff);
        out[blockIdx.x * 4 + 2] = lds[1];
    }
}
int main() {
    unsigned* out;
    hipMalloc(&out, 4096 * 16);
    for (int kb : {1, 8, 40, 64, 78, 80}) {
        const int n = 512;
        hipMemset(out, 0, 4096 * 16);
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
        hipLaunchKernelGGL(probe, dim3(n), dim3(256), kb * 1024, 0, out, 20000 /* 200 us */);
        hipDeviceSynchronize();
        std::vector<unsigned> h(n * 4);
        hipMemcpy(h.data(), out, n * 16, hipMemcpyDeviceToHost);
        printf("dynamic LDS %d KB: raw register of the first workgroups (hex), [7:0] | [20:12] | cu key\n", kb);
        for (int i = 0; i < 12; ++i) printf("   wg %3d: %08x  base %3u size %3u  cu %03x\n", i, h[i * 4], h[i * 4] & 0xff, (h[i * 4] >> 12) & 0x1ff, h[i * 4 + 1]);
        // distinct (base) values
        int seen[1024] = {0};
        for (int i = 0; i < n; ++i) seen[h[i * 4] & 0x3ff]++;
        printf("   distinct low-10-bit values:");
        for (int v = 0; v < 1024; ++v) if (seen[v]) printf(" %d(x%d)", v, seen[v]);
        printf("\n");
    }
    return 0;
}
