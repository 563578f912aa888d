// Candidate reduction (per problem x category top-K of the feasible services) and the full
// per-row ranking.  Integer/compare work, HBM-bound on the score rows: one wavefront per
// (problem, category) segment, each lane scanning a strided share of the segment; the K picks are
// K wave-wide max-reductions over 64-bit keys (score order bits : inverted id), so ties resolve to
// the lowest service id and the result does not depend on lane count or launch geometry.
#include "common.h"

__device__ __forceinline__ unsigned long long rank_key(float score, uint32_t id) {
    return ((unsigned long long)float_order_key(score) << 32) | (0xffffffffu - id);
}

__global__ __launch_bounds__(256) void select_candidates_kernel(
    const float* __restrict__ scores, int64_t ld_scores, const int32_t* __restrict__ cat_ptr,
    const double* __restrict__ qos, const double* __restrict__ local_bounds, const uint8_t* __restrict__ present,
    const double* __restrict__ global_bounds, float* __restrict__ out_rows, int32_t* __restrict__ out_ids,
    int32_t B, int32_t T, int32_t n_per) {
    const int lane = threadIdx.x & 63;
    const int64_t seg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= (int64_t)B * T) return;
    const int b = (int)(seg / T), c = (int)(seg - (int64_t)b * T);
    const int s_begin = cat_ptr[c], s_end = cat_ptr[c + 1];
    const bool pres = present[seg] != 0;
    const double* lb = local_bounds + seg * 4;
    const double lo_c = lb[0], hi_c = lb[1], lo_q = lb[2], hi_q = lb[3];
    const float* srow = scores + (int64_t)b * ld_scores;

    unsigned long long last = ~0ull;
    int my_pick = -1;   // lane r keeps the r-th pick
    int n_found = 0;
    if (pres) {
        for (int r = 0; r < n_per; ++r) {
            unsigned long long best = 0ull;
            for (int s = s_begin + lane; s < s_end; s += 64) {
                const double cost = qos[(int64_t)s * 4 + 2], qual = qos[(int64_t)s * 4 + 3];
                const bool feas = lo_c <= cost && cost <= hi_c && lo_q <= qual && qual <= hi_q;
                const unsigned long long key = rank_key(srow[s], (uint32_t)s);
                if (feas && key < last && key > best) best = key;
            }
            best = wave_max_u64(best);
            if (best == 0ull) break;
            if (lane == r) my_pick = (int)(0xffffffffu - (uint32_t)(best & 0xffffffffu));
            last = best;
            ++n_found;
        }
    }
    // emit n_per rows: picks repeated cyclically (loadData.py:137-141), dummy rows otherwise (:148)
    const int src_lane = n_found > 0 ? lane % n_found : 0;
    const int id = __shfl(my_pick, src_lane, 64);
    if (lane < n_per) {
        const int64_t pos = (int64_t)b * T * n_per + (int64_t)c * n_per + lane;
        float4 q, tail = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n_found > 0) {
            const double* qs = qos + (int64_t)id * 4;
            q = make_float4((float)qs[0], (float)qs[1], (float)qs[2], (float)qs[3]);
        } else {
            q = make_float4(0.f, 1.f, 1.f, 1.f);
        }
        if (c == 0) {
            const double* g = global_bounds + (int64_t)b * 4;
            tail = make_float4((float)g[0], (float)g[1], (float)g[2], (float)g[3]);
        }
        float4* dst = reinterpret_cast<float4*>(out_rows + pos * 8);
        dst[0] = q;
        dst[1] = tail;
        out_ids[pos] = n_found > 0 ? id : -1;
    }
}

// The same reduction with 16 lanes per (problem, category) segment, four segments per wave, for n_per <= 16: the row-wide
// maximum is four DPP rotations instead of six cross-wave shuffles on each key half, and a category of 5 services (the
// 1000-task shapes) no longer occupies a whole wave.  Same keys, same order of picks, same emitted rows.
__device__ __forceinline__ unsigned long long row16_max_u64(unsigned long long v) {
#define GNNPN_MAX_STEP(CTRL)                                                                                       \
    {                                                                                                              \
        const unsigned lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xF, 0xF, true);                \
        const unsigned hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, true);        \
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;                                          \
        v = o > v ? o : v;                                                                                         \
    }
    GNNPN_MAX_STEP(0x121) GNNPN_MAX_STEP(0x122) GNNPN_MAX_STEP(0x124) GNNPN_MAX_STEP(0x128)   // row_ror:1, 2, 4, 8
    return v;
}
// ... and over groups of 8 lanes: the neighbour, the other pair (quad_perm), the other quad (row_half_mirror); a maximum does
// not depend on the order it is formed in
__device__ __forceinline__ unsigned long long row8_max_u64(unsigned long long v) {
    GNNPN_MAX_STEP(0xB1) GNNPN_MAX_STEP(0x4E) GNNPN_MAX_STEP(0x141)
    return v;
#undef GNNPN_MAX_STEP
}

// W = 16 or 8 lanes per segment; 8 (n_per <= 8, categories of about 8 services or fewer: the 1000-task shapes have 5) puts eight
// segments into a wave — the kernel is bound by its vector instructions, which are per wave.
template <int W>
__global__ __launch_bounds__(256) void select_candidates16_kernel(
    const float* __restrict__ scores, int64_t ld_scores, const int32_t* __restrict__ cat_ptr,
    const double* __restrict__ qos, const double* __restrict__ local_bounds, const uint8_t* __restrict__ present,
    const double* __restrict__ global_bounds, float* __restrict__ out_rows, int32_t* __restrict__ out_ids,
    int32_t B, int32_t T, int32_t n_per) {
    const int lane = threadIdx.x & 63, sub = lane & (W - 1);
    const int64_t seg = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (64 / W) + lane / W;
    if (seg >= (int64_t)B * T) return;                                  // whole W-lane groups leave together
    const int b = (int64_t)B * T <= 0x7fffffffll ? (int)((unsigned)seg / (unsigned)T) : (int)(seg / T);   // (a 64-bit division per lane is ~100 instructions)
    const int c = (int)(seg - (int64_t)b * T);
    const int s_begin = cat_ptr[c], s_end = cat_ptr[c + 1];
    const bool pres = present[seg] != 0;
    const double* lb = local_bounds + seg * 4;
    const double lo_c = lb[0], hi_c = lb[1], lo_q = lb[2], hi_q = lb[3];
    const float* srow = scores + (int64_t)b * ld_scores;

    auto group_max = [](unsigned long long v) { return W == 16 ? row16_max_u64(v) : row8_max_u64(v); };
    unsigned long long last = ~0ull;
    int my_pick = -1;   // lane r of the row keeps the r-th pick
    int n_found = 0;
    const bool small = s_end - s_begin <= W;
    float4 own = make_float4(0.f, 1.f, 1.f, 1.f);                      // this lane's service's QoS row (small categories)
    if (pres && small) {
        // a category of at most W services (the 1000-task shapes: 5): every lane forms its ONE key once — 0 if the service is
        // infeasible or absent — and the rounds are max-reductions over registers (no re-read of the scores and bounds per pick);
        // the lane keeps its service's whole QoS row, so the emitted rows come from a lane exchange, not from a second,
        // dependent round of loads
        unsigned long long key = 0ull;
        const int s = s_begin + sub;
        if (s < s_end) {
            const double* qs = qos + (int64_t)s * 4;
            const double q0 = qs[0], q1 = qs[1], cost = qs[2], qual = qs[3];
            own = make_float4((float)q0, (float)q1, (float)cost, (float)qual);
            if (lo_c <= cost && cost <= hi_c && lo_q <= qual && qual <= hi_q) key = rank_key(srow[s], (uint32_t)s);
        }
        for (int r = 0; r < n_per; ++r) {
            const unsigned long long best = group_max(key < last ? key : 0ull);
            if (best == 0ull) break;
            if (sub == r) my_pick = (int)(0xffffffffu - (uint32_t)(best & 0xffffffffu));
            last = best;
            ++n_found;
        }
    } else if (pres) {
        for (int r = 0; r < n_per; ++r) {
            unsigned long long best = 0ull;
            for (int s = s_begin + sub; s < s_end; s += W) {
                const double cost = qos[(int64_t)s * 4 + 2], qual = qos[(int64_t)s * 4 + 3];
                const bool feas = lo_c <= cost && cost <= hi_c && lo_q <= qual && qual <= hi_q;
                const unsigned long long key = rank_key(srow[s], (uint32_t)s);
                if (feas && key < last && key > best) best = key;
            }
            best = group_max(best);
            if (best == 0ull) break;
            if (sub == r) my_pick = (int)(0xffffffffu - (uint32_t)(best & 0xffffffffu));
            last = best;
            ++n_found;
        }
    }
    // emit n_per rows: picks repeated cyclically (loadData.py:137-141), dummy rows otherwise (:148)
    const int src_lane = (lane & ~(W - 1)) + (n_found > 0 ? sub % n_found : 0);
    const int id = __shfl(my_pick, src_lane, 64);
    float4 q = make_float4(0.f, 1.f, 1.f, 1.f);
    if (small) {                                                       // (the whole group of W lanes takes this branch together)
        const int owner = n_found > 0 ? (lane & ~(W - 1)) + (id - s_begin) : lane;
        const float4 g = make_float4(__shfl(own.x, owner, 64), __shfl(own.y, owner, 64), __shfl(own.z, owner, 64), __shfl(own.w, owner, 64));
        if (n_found > 0) q = g;
    } else if (n_found > 0 && sub < n_per) {
        const double* qs = qos + (int64_t)id * 4;
        q = make_float4((float)qs[0], (float)qs[1], (float)qs[2], (float)qs[3]);
    }
    if (sub < n_per) {
        const int64_t pos = (int64_t)b * T * n_per + (int64_t)c * n_per + sub;
        float4 tail = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c == 0) {
            const double* g = global_bounds + (int64_t)b * 4;
            tail = make_float4((float)g[0], (float)g[1], (float)g[2], (float)g[3]);
        }
        float4* dst = reinterpret_cast<float4*>(out_rows + pos * 8);
        dst[0] = q;
        dst[1] = tail;
        out_ids[pos] = n_found > 0 ? id : -1;
    }
}

extern "C" int gnnpn_select_candidates(const float* scores, int64_t ld_scores, const int32_t* cat_ptr,
                                       const double* qos, const double* local_bounds, const uint8_t* present,
                                       const double* global_bounds, float* out_rows, int32_t* out_ids, int32_t B,
                                       int32_t T, int32_t n_per, void* stream) {
    GNNPN_REQUIRE(B >= 0 && T > 0, "select_candidates: bad shape");
    if (B == 0) return GNNPN_OK;
    GNNPN_REQUIRE(scores && cat_ptr && qos && local_bounds && present && global_bounds && out_rows && out_ids,
                  "select_candidates: null operand");
    GNNPN_REQUIRE(n_per >= 1 && n_per <= 64, "select_candidates: n_per must be in [1,64], got %d", n_per);
    GNNPN_REQUIRE(gnnpn_aligned(out_rows, 16), "select_candidates: out_rows must be 16-byte aligned");
    if (B == 0) return GNNPN_OK;
    const int64_t n_seg = (int64_t)B * T;
    if (n_per <= 8 && ld_scores <= 8ll * T)                            // (ld_scores >= the number of services: mean category <= 8)
        hipLaunchKernelGGL(select_candidates16_kernel<8>, dim3((unsigned)((n_seg + 31) / 32)), dim3(256), 0,
                           (hipStream_t)stream, scores, ld_scores, cat_ptr, qos, local_bounds, present, global_bounds,
                           out_rows, out_ids, B, T, n_per);
    else if (n_per <= 16)
        hipLaunchKernelGGL(select_candidates16_kernel<16>, dim3((unsigned)((n_seg + 15) / 16)), dim3(256), 0,
                           (hipStream_t)stream, scores, ld_scores, cat_ptr, qos, local_bounds, present, global_bounds,
                           out_rows, out_ids, B, T, n_per);
    else
        hipLaunchKernelGGL(select_candidates_kernel, dim3((unsigned)((n_seg + 3) / 4)), dim3(256), 0,
                           (hipStream_t)stream, scores, ld_scores, cat_ptr, qos, local_bounds, present, global_bounds,
                           out_rows, out_ids, B, T, n_per);
    GNNPN_CHECK_LAUNCH("select_candidates");
    return GNNPN_OK;
}

// ---------------------------------------------------------------------------------------------
// Full ranking of one score row per workgroup: bitonic sort of 64-bit keys in LDS (descending).
__global__ __launch_bounds__(1024) void rank_rows_kernel(const float* __restrict__ scores, int64_t ld_scores,
                                                         int32_t* __restrict__ ranking, int32_t S, int32_t P) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const float* srow = scores + (int64_t)blockIdx.x * ld_scores;
    for (int i = threadIdx.x; i < P; i += blockDim.x) keys[i] = i < S ? rank_key(srow[i], (uint32_t)i) : 0ull;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
                const int i = 2 * t - (t & (j - 1));   // lower index of the pair (bit j clear)
                const int p = i + j;
                const bool desc = (i & k) == 0;        // descending blocks first -> overall descending
                const unsigned long long a = keys[i], b = keys[p];
                if ((a < b) == desc) {
                    keys[i] = b;
                    keys[p] = a;
                }
            }
            __syncthreads();
        }
    }
    int32_t* out = ranking + (int64_t)blockIdx.x * S;
    for (int i = threadIdx.x; i < S; i += blockDim.x) out[i] = (int32_t)(0xffffffffu - (uint32_t)(keys[i] & 0xffffffffu));
}

// Rows of 16385..32768 services (BASELINE configs[4]: S = 20000): the 64-bit keys no longer fit the LDS, so the
// same bitonic network runs on 32-bit service ids in LDS and forms each key from the (L2-resident) score row
// when it compares.  Identical order by construction; only used on the artefact path.
__global__ __launch_bounds__(1024) void rank_rows_indirect_kernel(const float* __restrict__ scores, int64_t ld_scores,
                                                                  int32_t* __restrict__ ranking, int32_t S, int32_t P) {
    extern __shared__ __attribute__((aligned(16))) unsigned int ids[];
    const float* srow = scores + (int64_t)blockIdx.x * ld_scores;
    auto key_of = [&](unsigned int i) { return i < (unsigned)S ? rank_key(srow[i], i) : 0ull; };
    for (int i = threadIdx.x; i < P; i += blockDim.x) ids[i] = (unsigned)i;
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (P >> 1); t += blockDim.x) {
                const int i = 2 * t - (t & (j - 1));
                const int p = i + j;
                const bool desc = (i & k) == 0;
                const unsigned int ia = ids[i], ib = ids[p];
                if ((key_of(ia) < key_of(ib)) == desc) {
                    ids[i] = ib;
                    ids[p] = ia;
                }
            }
            __syncthreads();
        }
    }
    int32_t* out = ranking + (int64_t)blockIdx.x * S;
    for (int i = threadIdx.x; i < S; i += blockDim.x) out[i] = (int32_t)ids[i];
}

extern "C" int gnnpn_rank_rows(const float* scores, int64_t ld_scores, int32_t* ranking, int32_t B, int32_t S,
                               void* stream) {
    GNNPN_REQUIRE(B >= 0 && S > 0 && ld_scores >= S, "rank_rows: bad shape");
    if (B == 0) return GNNPN_OK;
    GNNPN_REQUIRE(scores && ranking, "rank_rows: null operand");
    if (S > 32768) GNNPN_FAIL(GNNPN_E_UNSUP, "rank_rows: S=%d exceeds the single-workgroup LDS sort (32768)", S);
    if (B == 0) return GNNPN_OK;
    int P = 2;
    while (P < S) P <<= 1;
    if (S > 16384) {
        const size_t lds_i = (size_t)P * sizeof(unsigned int);
        hipError_t ei = hipFuncSetAttribute((const void*)rank_rows_indirect_kernel,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_i);
        if (ei != hipSuccess) GNNPN_FAIL(GNNPN_E_LAUNCH, "rank_rows: cannot reserve %zu B of LDS: %s", lds_i, hipGetErrorString(ei));
        hipLaunchKernelGGL(rank_rows_indirect_kernel, dim3(B), dim3(1024), lds_i, (hipStream_t)stream, scores, ld_scores,
                           ranking, S, P);
        GNNPN_CHECK_LAUNCH("rank_rows");
        return GNNPN_OK;
    }
    const size_t lds = (size_t)P * sizeof(unsigned long long);
    hipError_t e = hipFuncSetAttribute((const void*)rank_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) GNNPN_FAIL(GNNPN_E_LAUNCH, "rank_rows: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
    hipLaunchKernelGGL(rank_rows_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, scores, ld_scores, ranking, S, P);
    GNNPN_CHECK_LAUNCH("rank_rows");
    return GNNPN_OK;
}

// ---------------------------------------------------------------------------------------------
// P@k of TrainML.test (trainML.py:63-70): fraction of the top-k ranked services whose label is 1.
__global__ void precision_at_k_kernel(const int32_t* __restrict__ ranking, int64_t ld_rank,
                                      const float* __restrict__ labels, int64_t ld_lab, int32_t B, int32_t S,
                                      const int32_t* __restrict__ ks, int32_t n_k, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * n_k) return;
    const int b = i / n_k, k = ks[i - b * n_k];
    int hits = 0;
    for (int j = 0; j < k && j < S; ++j) {
        const int s = ranking[(int64_t)b * ld_rank + j];
        hits += (s >= 0 && s < S && labels[(int64_t)b * ld_lab + s] == 1.0f);
    }
    out[i] = (float)hits / (float)k;
}

extern "C" int gnnpn_precision_at_k(const int32_t* ranking, int64_t ld_rank, const float* labels, int64_t ld_lab,
                                    int32_t B, int32_t S, const int32_t* ks, int32_t n_k, float* out, void* stream) {
    GNNPN_REQUIRE(B >= 0 && S > 0 && n_k > 0 && ld_rank >= 1 && ld_lab >= S, "precision_at_k: bad shape");
    if (B == 0) return GNNPN_OK;
    GNNPN_REQUIRE(ranking && labels && ks && out, "precision_at_k: null operand");
    hipLaunchKernelGGL(precision_at_k_kernel, dim3((B * n_k + 127) / 128), dim3(128), 0, (hipStream_t)stream, ranking,
                       ld_rank, labels, ld_lab, B, S, ks, n_k, out);
    GNNPN_CHECK_LAUNCH("precision_at_k");
    return GNNPN_OK;
}
