// Does an LDS-DMA load (global_load_lds_dwordx4: wave-uniform LDS base in M0 + lane * 16) reach LDS addresses above 64 KB on gfx950?
// The double-buffered tiled aggregate (csrc/graph_tiled.hip, round 6) keeps two 80 KB source tiles in a CU's 160 KB and fills
// the idle one by LDS-DMA while the other is gathered from: its second buffer lies at 80 KB .. 160 KB, its first crosses 64 KB.
// One workgroup of 1024 threads, 160 KB of dynamic LDS; every wavefront DMAs 1 KB pieces of a known pattern to a list of LDS
// byte addresses (below 64 KB, across it, up to the last KB), waits, barrier, reads them back with ds_read_b128 and compares.
//     hipcc --offload-arch=gfx950 -O2 -o lds_dma_high_probe lds_dma_high_probe.hip && ./lds_dma_high_probe
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

constexpr int LDS_BYTES = 160 * 1024;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// piece p (1 KB) of the LDS image <- src words [p * 256, p * 256 + 256); then out[] = the LDS image read back by ds_read_b128
__global__ __launch_bounds__(1024) void probe(const uint4* __restrict__ src, uint4* __restrict__ out, int n_pieces) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 1024) reinterpret_cast<uint4*>(lds)[i] = make_uint4(0xDEADu, 0xDEADu, 0xDEADu, 0xDEADu);
    __syncthreads();
    for (int p = wave; p < n_pieces; p += 16) glds16(src + (size_t)p * 64 + lane, (unsigned)(p * 1024));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < LDS_BYTES / 16; i += 1024) out[i] = reinterpret_cast<const uint4*>(lds)[i];
}

int main() {
    const int n_pieces = LDS_BYTES / 1024;
    std::vector<unsigned> h((size_t)LDS_BYTES / 4), back((size_t)LDS_BYTES / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x10000000u + (unsigned)i;
    uint4 *src, *out;
    CK(hipMalloc(&src, LDS_BYTES));
    CK(hipMalloc(&out, LDS_BYTES));
    CK(hipMemcpy(src, h.data(), LDS_BYTES, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipLaunchKernelGGL(probe, dim3(1), dim3(1024), LDS_BYTES, 0, src, out, n_pieces);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(back.data(), out, LDS_BYTES, hipMemcpyDeviceToHost));
    int bad_kb = 0, first_bad = -1;
    for (int p = 0; p < n_pieces; ++p) {
        bool ok = true;
        for (int i = 0; i < 256; ++i) ok = ok && back[(size_t)p * 256 + i] == h[(size_t)p * 256 + i];
        if (!ok) {
            ++bad_kb;
            if (first_bad < 0) first_bad = p;
        }
    }
    printf("{\"probe\": \"lds_dma_high\", \"pieces\": %d, \"bad_pieces\": %d, \"first_bad_kb\": %d, \"word_at_70KB\": \"0x%x\", \"word_at_159KB\": \"0x%x\"}\n",
           n_pieces, bad_kb, first_bad, back[70 * 256], back[159 * 256]);
    return bad_kb ? 1 : 0;
}
