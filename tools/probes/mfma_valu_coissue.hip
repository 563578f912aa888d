// probe: how many independent vector instructions fit into the shadow of a v_mfma_f32_16x16x4_f32 issued by the SAME wave?
// 128 MFMAs (two chains) per repetition, N x v_fma_f32 / v_pk_fma_f32 / v_exp_f32 between consecutive MFMAs, 1 or 2 waves
// per SIMD.  Prints shader cycles per MFMA.   hipcc --offload-arch=gfx950 -O3 mfma_valu_coissue.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int N, int KIND>   // KIND 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_exp_f32, 3: v_mov_b32 dpp
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int reps) {
    const int lane = threadIdx.x & 63;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    float a = 0.5f + lane, b = 0.25f;
    float f[8];
    f32x2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { f[i] = 1.0f + i + lane; p[i] = f32x2{1.0f + i, 2.0f + lane}; }
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[j % 8]) : "v"(b));
                else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j % 8]) : "v"(p[(j + 1) % 8 == j % 8 ? 0 : 7 - (j % 8)]));
                else if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(f[j % 8]));
                else asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(f[j % 8]));
            }
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc1) : "v"(b), "v"(a));
#pragma unroll
            for (int j = 0; j < N; ++j) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[(j + 4) % 8]) : "v"(b));
                else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[(j + 4) % 8]) : "v"(p[(j + 4) % 8 == 0 ? 1 : 0]));
                else if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(f[(j + 4) % 8]));
                else asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(f[(j + 4) % 8]));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = acc0[0] + acc1[1];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int N, int KIND>
void run(const char* name, int wg_per_cu, float* out, unsigned long long* cyc) {
    const int reps = 50, blocks = 256 * wg_per_cu;
    hipLaunchKernelGGL((k<N, KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<N, KIND>), dim3(blocks), dim3(256), 0, 0, out, cyc, reps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-12s N=%d  %d wave(s)/SIMD: %6.1f memtime ticks per MFMA, %7.2f ns per MFMA per wave (wall)\n", name, N, wg_per_cu,
           (double)c / (reps * 128.0), ms * 1e6 / (reps * 128.0));
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4 * 2); hipMalloc(&cyc, 64);
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>("none", w, out, cyc);
        run<1, 0>("v_fma_f32", w, out, cyc); run<2, 0>("v_fma_f32", w, out, cyc); run<4, 0>("v_fma_f32", w, out, cyc);
        run<6, 0>("v_fma_f32", w, out, cyc); run<8, 0>("v_fma_f32", w, out, cyc);
        run<2, 1>("v_pk_fma_f32", w, out, cyc); run<4, 1>("v_pk_fma_f32", w, out, cyc); run<6, 1>("v_pk_fma_f32", w, out, cyc);
        run<1, 2>("v_exp_f32", w, out, cyc); run<2, 2>("v_exp_f32", w, out, cyc); run<4, 2>("v_exp_f32", w, out, cyc);
        run<2, 3>("mov_dpp", w, out, cyc); run<4, 3>("mov_dpp", w, out, cyc); run<6, 3>("mov_dpp", w, out, cyc);
    }
    return 0;
}
