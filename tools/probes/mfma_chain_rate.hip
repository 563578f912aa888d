// What paces a chain of 96 v_mfma_f32_16x16x32_f16 per wave (the exact-split recurrent product of coop_common.h)?
// One workgroup of 256 threads per CU (one wave per SIMD), `reps` chains back to back, s_memtime around them.
//   variant 0: 96 MFMAs on 8 accumulators, operands in registers (no LDS, no VALU)
//   variant 1: + the 32 ds_read_b128 of the A-fragments / third-piece bytes (values used)
//   variant 2: + the 64 v_perm_b32 that expand the bytes
//   variant 3: variant 0 with ONE accumulator (dependent chain)
//   variant 4: variant 0 with the A/B/C register numbers forced to different banks is not expressible in HIP: skipped
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_chain_rate mfma_chain_rate.hip ; run: ./mfma_chain_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256) void chain(const float* seed, float* out, unsigned long long* cycles, int reps) {
    __shared__ __attribute__((aligned(16))) _Float16 hs[3 * 16 * 264];
    __shared__ __attribute__((aligned(16))) unsigned wt[4 * 8 * 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, kq = lane >> 4;
    for (int i = tid; i < 3 * 16 * 264; i += 256) hs[i] = (_Float16)(seed[i & 255]);
    for (int i = tid; i < 4 * 8 * 64 * 4; i += 256) wt[i] = 0x3c3c3c3cu;
    f16x8 w0[2][8], w1[2][8];
    for (int n = 0; n < 2; ++n)
        for (int kk = 0; kk < 8; ++kk)
            for (int j = 0; j < 8; ++j) {
                w0[n][kk][j] = (_Float16)seed[(n * 64 + kk * 8 + j + lane) & 255];
                w1[n][kk][j] = (_Float16)seed[(n * 64 + kk * 8 + j + lane + 7) & 255];
            }
    __syncthreads();
    const _Float16* base = hs + c * 264 + 8 * kq;
    const unsigned* wl = wt + (wave * 8 * 64 + lane) * 4;
    f32x4 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 h0 = *reinterpret_cast<const f16x8*>(base), h1 = *reinterpret_cast<const f16x8*>(base + 16 * 264),
          h2 = *reinterpret_cast<const f16x8*>(base + 2 * 16 * 264);
    u32x4 t = *reinterpret_cast<const u32x4*>(wl);
    f16x8 e0 = h0, e1 = h1;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (V >= 1) {
                h0 = *reinterpret_cast<const f16x8*>(base + 32 * kk);
                h1 = *reinterpret_cast<const f16x8*>(base + 16 * 264 + 32 * kk);
                h2 = *reinterpret_cast<const f16x8*>(base + 2 * 16 * 264 + 32 * kk);
                t = *reinterpret_cast<const u32x4*>(wl + 4 * 64 * kk);
            }
            if (V >= 2) {
                const u32x4 x0 = {__builtin_amdgcn_perm(0u, t.x, 0x010c000cu), __builtin_amdgcn_perm(0u, t.x, 0x030c020cu),
                                  __builtin_amdgcn_perm(0u, t.y, 0x010c000cu), __builtin_amdgcn_perm(0u, t.y, 0x030c020cu)};
                const u32x4 x1 = {__builtin_amdgcn_perm(0u, t.z, 0x010c000cu), __builtin_amdgcn_perm(0u, t.z, 0x030c020cu),
                                  __builtin_amdgcn_perm(0u, t.w, 0x010c000cu), __builtin_amdgcn_perm(0u, t.w, 0x030c020cu)};
                e0 = __builtin_bit_cast(f16x8, x0);
                e1 = __builtin_bit_cast(f16x8, x1);
            } else if (V == 1) {
                e0 = __builtin_bit_cast(f16x8, t);
                e1 = __builtin_bit_cast(f16x8, t);
            }
            constexpr int ONE = V == 3;
            a[ONE ? 0 : 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, w1[0][kk], a[ONE ? 0 : 0], 0, 0, 0);
            a[ONE ? 0 : 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, w1[1][kk], a[ONE ? 0 : 1], 0, 0, 0);
            a[ONE ? 0 : 2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, w0[0][kk], a[ONE ? 0 : 2], 0, 0, 0);
            a[ONE ? 0 : 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, w0[1][kk], a[ONE ? 0 : 3], 0, 0, 0);
            a[ONE ? 0 : 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, e0, a[ONE ? 0 : 0], 0, 0, 0);
            a[ONE ? 0 : 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, e1, a[ONE ? 0 : 1], 0, 0, 0);
            a[ONE ? 0 : 4 + (kk >> 2)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, w0[0][kk], a[ONE ? 0 : 4 + (kk >> 2)], 0, 0, 0);
            a[ONE ? 0 : 6 + (kk >> 2)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, w0[1][kk], a[ONE ? 0 : 6 + (kk >> 2)], 0, 0, 0);
            a[ONE ? 0 : 0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h2, w0[0][kk], a[ONE ? 0 : 0], 0, 0, 0);
            a[ONE ? 0 : 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h2, w0[1][kk], a[ONE ? 0 : 1], 0, 0, 0);
            a[ONE ? 0 : 2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, w1[0][kk], a[ONE ? 0 : 2], 0, 0, 0);
            a[ONE ? 0 : 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, w1[1][kk], a[ONE ? 0 : 3], 0, 0, 0);
        }
    }
    f32x4 s = a[0];
    for (int i = 1; i < 8; ++i) s += a[i];
    asm volatile("" ::"v"(s));
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + tid] = s[0] + s[1] + s[2] + s[3];
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int V>
static void run(const char* what, const float* seed, float* out, unsigned long long* cyc, int blocks) {
    const int reps = 64;
    hipLaunchKernelGGL(chain<V>, dim3(blocks), dim3(256), 0, 0, seed, out, cyc, reps);
    hipLaunchKernelGGL(chain<V>, dim3(blocks), dim3(256), 0, 0, seed, out, cyc, reps);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0;
    for (auto v : h) m += (double)v;
    m /= blocks;
    printf("%-72s %8.1f ticks per chain of 96 MFMAs = %5.2f ticks per MFMA\n", what, m / reps, m / reps / 96);
}

int main() {
    float *seed, *out;
    unsigned long long* cyc;
    std::vector<float> hsd(256);
    for (int i = 0; i < 256; ++i) hsd[i] = (float)((i * 37) % 17 - 8) / 16.0f;
    hipMalloc(&seed, 1024);
    hipMemcpy(seed, hsd.data(), 1024, hipMemcpyHostToDevice);
    hipMalloc(&out, 1024 * 256 * 4);
    hipMalloc(&cyc, 1024 * 8);
    for (int blocks : {256, 512}) {
        printf("-- %d workgroups of 256 threads (%d per CU)\n", blocks, blocks / 256);
        run<0>("registers only, 8 accumulators", seed, out, cyc, blocks);
        run<3>("registers only, ONE accumulator", seed, out, cyc, blocks);
        run<1>("+ 32 ds_read_b128", seed, out, cyc, blocks);
        run<2>("+ 32 ds_read_b128 + 64 v_perm_b32", seed, out, cyc, blocks);
    }
    printf("(ticks: s_memtime-class counter via __builtin_readcyclecounter; the stamps of tools/stamp_encode.py use the same)\n");
    return 0;
}
