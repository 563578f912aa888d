// micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 with NC independent accumulator chains
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NC>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, float a0, float b0) {
    f32x4 acc[NC];
    for (int i = 0; i < NC; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    float bb[NC];
    for (int i = 0; i < NC; ++i) bb[i] = b + 0.125f * i;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
    for (int it = 0; it < 1024 / NC; ++it) {
#pragma unroll
        for (int i = 0; i < NC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bb[i], acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(s) : "memory");
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NC> void run(int blocks) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<NC>, dim3(blocks), dim3(256), 0, 0, out, cyc, 0.5f, 0.25f);
    hipDeviceSynchronize();
    unsigned long long h[1024]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
    printf("chains=%d blocks=%d : %.1f cycles per MFMA (1024 MFMAs per wave)\n", NC, blocks, s / blocks / 1024.0);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1>(1); run<2>(1); run<4>(1); run<8>(1);
    run<1>(256); run<2>(256); run<4>(256); run<8>(256);
    return 0;
}
