// Premise check for a two-tiles-per-workgroup recurrent kernel (round 5): can ONE wave per SIMD run the exact-split product of tile A
// (96 v_mfma_f32_16x16x32_f16 + LDS reads + 64 v_perm) with the cell update of tile B (5 packed activations, ~200 vector
// instructions, a dependent chain) folded into the matrix instructions' shadows — instead of one after the other?
// Uses the kernels' own device functions (csrc/coop_common.h).  Per variant: cycles per iteration, one workgroup per CU.
//   0  chain only (split_chain as the kernels call it: a scheduling fence behind every product)
//   1  chain, then cell update (what a lone workgroup does per step today, hand-off and barriers aside)
//   2  the same chain WITHOUT fences + the cell update of independent data in the same scheduling region, ordered by
//      sched_group_barrier: 1 MFMA, then up to 3 VALU, repeated
//   3  cell update only
//     hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I gnnpn-sc_amd/csrc -I include -o pair_interleave_probe tools/probes/pair_interleave_probe.hip
#include "coop_common.h"
#include <cstdio>
#include <vector>

__device__ __forceinline__ void chain_nofence(const _Float16* base, const f16x8 (&w0)[2][8], const f16x8 (&w1)[2][8], const unsigned* wt,
                                              const float (&inv)[2], f32x4 (&acc)[2]) {
    f32x4 a0[2], a0b[2], a1[2], a2[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) a0[n] = a0b[n] = a1[n] = a2[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const f16x8 h0 = *reinterpret_cast<const f16x8*>(base + 32 * kk);
        const f16x8 h1 = *reinterpret_cast<const f16x8*>(base + SPLIT_TILE + 32 * kk);
        const f16x8 h2 = *reinterpret_cast<const f16x8*>(base + 2 * SPLIT_TILE + 32 * kk);
        const u32x4 t = *reinterpret_cast<const u32x4*>(wt + 4 * 64 * kk);
        const f16x8 e0 = split_expand(t.x, t.y), e1 = split_expand(t.z, t.w);
        a2[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, w1[0][kk], a2[0], 0, 0, 0);
        a2[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, w1[1][kk], a2[1], 0, 0, 0);
        a1[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, w0[0][kk], a1[0], 0, 0, 0);
        a1[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, w0[1][kk], a1[1], 0, 0, 0);
        a2[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, e0, a2[0], 0, 0, 0);
        a2[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, e1, a2[1], 0, 0, 0);
        if (kk < 4) {
            a0[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, w0[0][kk], a0[0], 0, 0, 0);
            a0[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, w0[1][kk], a0[1], 0, 0, 0);
        } else {
            a0b[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, w0[0][kk], a0b[0], 0, 0, 0);
            a0b[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, w0[1][kk], a0b[1], 0, 0, 0);
        }
        a2[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h2, w0[0][kk], a2[0], 0, 0, 0);
        a2[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h2, w0[1][kk], a2[1], 0, 0, 0);
        a1[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, w1[0][kk], a1[0], 0, 0, 0);
        a1[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, w1[1][kk], a1[1], 0, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[n][r] = __fmul_rn(fmaf(fmaf(a2[n][r], SPLIT_INV, a1[n][r]), SPLIT_INV, __fadd_rn(a0[n][r], a0b[n][r])), inv[n]);
}

template <int V, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void probe(const float* __restrict__ seed, float* __restrict__ out, unsigned long long* __restrict__ cycles, int reps) {
    __shared__ __attribute__((aligned(16))) _Float16 hs[3 * SPLIT_TILE];
    __shared__ __attribute__((aligned(16))) unsigned wts[SPLIT_WT_DWORDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, kq = lane >> 4;
    for (int i = tid; i < 3 * SPLIT_TILE; i += 256) hs[i] = (_Float16)(seed[i & 255] * 0.01f);
    for (int i = tid; i < SPLIT_WT_DWORDS; i += 256) wts[i] = 0x04040404u;
    f16x8 w0[2][8], w1[2][8];
    for (int n = 0; n < 2; ++n)
        for (int kk = 0; kk < 8; ++kk)
            for (int j = 0; j < 8; ++j) {
                w0[n][kk][j] = (_Float16)seed[(n * 64 + kk * 8 + j + lane) & 255];
                w1[n][kk][j] = (_Float16)seed[(n * 64 + kk * 8 + j + lane + 7) & 255];
            }
    __syncthreads();
    const _Float16* base = hs + c * LDH16 + 8 * kq;
    const unsigned* wl = wts + (wave * 8 * 64 + lane) * 4;
    const float inv[2] = {1.0f / 1024.0f, 1.0f / 2048.0f};
    f32x2 cst = {0.1f, 0.2f}, hl = {0.f, 0.f};
    f32x4 accA[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 accB[2] = {f32x4{seed[lane], seed[lane + 1], seed[lane + 2], seed[lane + 3]}, f32x4{seed[lane + 4], seed[lane + 5], seed[lane + 6], seed[lane + 7]}};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if constexpr (V == 0 || V == 1) split_chain(base, w0, w1, wl, inv, accA);
        if constexpr (V >= 1) {
            // cell update of the OTHER tile: its gate sums are last iteration's products (independent of this iteration's chain)
            f32x2 g0[2], g1[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                g0[q] = f32x2{accB[0][2 * q], accB[0][2 * q + 1]} + pk_set(0.01f);
                g1[q] = f32x2{accB[1][2 * q], accB[1][2 * q + 1]} + pk_set(0.02f);
            }
            cell_update_split(g0, g1, c < 8, cst, hl);
        }
        if constexpr (V == 2) chain_nofence(base, w0, w1, wl, inv, accA);    // (source order: cell update first, the chain behind it)
        if constexpr (V == 2) {
#pragma unroll
            for (int i = 0; i < 96; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                if (i % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // an LDS read every third gap (32 in all)
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // up to three VALU
            }
        }
        // rotate: this iteration's products are the next iteration's "other tile" (keeps both results live and used)
        accB[0] = accA[0] + f32x4{hl.x, hl.y, hl.x, hl.y};
        accB[1] = accA[1];
        if (r == reps - 1) hs[tid] = (_Float16)hl.x;                     // keep the LDS live
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + tid] = accB[0][0] + accB[1][1] + cst.x + hl.y;
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    const int blocks = 256, reps = 400;
    float *seed, *out;
    unsigned long long* cyc;
    hipMalloc(&seed, 1024 * 4);
    hipMalloc(&out, blocks * 256 * 4);
    hipMalloc(&cyc, blocks * 8);
    float* out2;
    unsigned long long* cyc2;
    hipMalloc(&out2, 2 * blocks * 256 * 4);
    hipMalloc(&cyc2, 2 * blocks * 8);
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0.001f * (float)((i * 37) % 211) - 0.1f;
    hipMemcpy(seed, h.data(), 4096, hipMemcpyHostToDevice);
    std::vector<unsigned long long> c(blocks);
    auto run = [&](const char* what, auto kern) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, seed, out, cyc, reps);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, seed, out, cyc, reps);
        if (hipDeviceSynchronize() != hipSuccess) { printf("failed: %s\n", what); return; }
        hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : c) s += (double)v;
        printf("{\"variant\": \"%s\", \"cycles_per_iteration\": %.0f}\n", what, s / blocks / reps);
    };
    run("0 chain only (fenced, as the kernels)", probe<0>);
    run("1 chain then cell update (sequential)", probe<1>);
    run("2 fence-free chain + cell update in one region, sched_group_barrier 1 MFMA : 3 VALU", probe<2>);
    run("3 cell update only", probe<3>);
    // two workgroups per CU (what two co-resident launches are): cycles per iteration of EACH workgroup.  Ideal sharing of a SIMD by
    // two waves = max(matrix pipe 2 x chain, issue); no overlap at all = 2 x variant 1
    auto run2 = [&](const char* what, auto kern) {
        hipLaunchKernelGGL(kern, dim3(2 * blocks), dim3(256), 0, 0, seed, out2, cyc2, reps);
        hipLaunchKernelGGL(kern, dim3(2 * blocks), dim3(256), 0, 0, seed, out2, cyc2, reps);
        if (hipDeviceSynchronize() != hipSuccess) { printf("failed: %s\n", what); return; }
        std::vector<unsigned long long> c2(2 * blocks);
        hipMemcpy(c2.data(), cyc2, 2 * blocks * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : c2) s += (double)v;
        printf("{\"variant\": \"%s\", \"cycles_per_iteration\": %.0f}\n", what, s / (2 * blocks) / reps);
    };
    run2("1 chain then cell update, TWO workgroups per CU (512 workgroups)", probe<1, 2>);
    run2("0 chain only, TWO workgroups per CU", probe<0, 2>);
    run2("3 cell update only, TWO workgroups per CU", probe<3, 2>);
    return 0;
}
