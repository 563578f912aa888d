// micro-benchmark 2: the encoder's MFMA phase in isolation: 128 v_mfma_f32_16x16x4_f32 per wave in two
// chains, B-fragments = 128 distinct VGPRs, A-fragments read from an LDS tile (chunked prefetch), 4 waves/CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LDH = 258;
template <int MODE>   // 0: A from LDS (as the kernel), 1: A from registers (no LDS traffic)
__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* cyc, const float* w) {
    __shared__ float hs[16 * LDH];
    for (int i = threadIdx.x; i < 16 * LDH; i += 256) hs[i] = 0.001f * i;
    const int lane = threadIdx.x & 63, kq = lane >> 4, c = lane & 15;
    float w0[64], w1[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) { w0[i] = w[(i * 256 + threadIdx.x)]; w1[i] = w[(64 + i) * 256 + threadIdx.x]; }
    __syncthreads();
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    unsigned long long t0, t1, total = 0;
    for (int rep = 0; rep < 16; ++rep) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        const float* base = hs + c * LDH + kq;
        float a[2][16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[0][i] = MODE == 0 ? base[4 * i] : 0.5f + i;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            if (ch < 3) {
#pragma unroll
                for (int i = 0; i < 16; ++i) a[(ch + 1) & 1][i] = MODE == 0 ? base[4 * (16 * (ch + 1) + i)] : 0.25f + i + ch;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ch & 1][i], w0[16 * ch + i], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ch & 1][i], w1[16 * ch + i], acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(acc0[0]), "v"(acc1[0]) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        total += t1 - t0;
        __syncthreads();
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1];
    if (threadIdx.x == 0) cyc[blockIdx.x] = total / 16;
}
template <int MODE> void run(int blocks, const float* w) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, w);
    hipDeviceSynchronize();
    unsigned long long h[1024]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
    printf("mode %d (A from %s) blocks=%d : %.0f cycles per 128-MFMA phase = %.1f per MFMA\n", MODE, MODE == 0 ? "LDS" : "regs", blocks, s / blocks, s / blocks / 128.0);
    hipFree(out); hipFree(cyc);
}
int main() {
    float* w; hipMalloc(&w, 128 * 256 * 4); hipMemset(w, 0, 128 * 256 * 4);
    run<0>(1, w); run<1>(1, w); run<0>(256, w); run<1>(256, w);
    return 0;
}
