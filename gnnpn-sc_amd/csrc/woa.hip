// ES-WOA fine-tuner (reference src/baselines/WOA.py:8-162), one wavefront per problem.
//
// The search is sequential in the individuals (every accepted individual moves the recorded best that the next one is
// attracted to) and its random draws are consumed in a data-dependent order, so the parallelism is across problems
// (embarrassingly: 1000 test problems = 1000 waves) and, inside a problem, across the T service categories: lane j owns
// category j's position of every individual.  All state of a problem lives in LDS (positions pop x T int32, the
// problem's candidate table n x 4 float64).
//
// Every draw is draw k of the counter-based stream of oracle/woa.py (splitmix64 of seed + k*golden), so a run is a pure
// function of (inputs, seed) and can be compared draw for draw with the restatement that is pinned against the real
// reference class.  The float64 figure of merit reproduces numpy's evaluation orders: np.cumprod = one sequential
// chain per QoS column, np.sum = the 8-accumulator pairwise block (n <= 128), np.min exact.  Python semantics restated:
// round-half-even (rint), modulo with the divisor's sign, negative positions index from the end, and the reference's
// list aliasing between the recorded best and the individual it was taken from (`alias`).
#include "common.h"

namespace {
constexpr double PE = 0.2;
constexpr unsigned long long GOLDEN = 0x9E3779B97F4A7C15ull;

__device__ __forceinline__ double draw_uniform(unsigned long long seed, unsigned long long k) {
    unsigned long long z = seed + GOLDEN * k;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ int draw_below(unsigned long long seed, unsigned long long k, int n) {
    return (int)__dmul_rn(draw_uniform(seed, k), (double)n);
}
__device__ __forceinline__ double wave_bcast(double v, int lane) {
    const int lo = __shfl(__double2loint(v), lane), hi = __shfl(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// violate + objFunc of the composition whose category-j row is (q[0..3]) in lane j (lanes >= T idle).
// `col` = 4 x 64 doubles of LDS scratch.
__device__ double figure_of_merit(const double (&q)[4], int T, int lane, double* col, const double* bounds) {
    if (lane < T) {
#pragma unroll
        for (int c = 0; c < 4; ++c) col[c * 64 + lane] = q[c];
    }
    __syncthreads();
    // np.cumprod of columns 2 and 3: lanes 0 and 1 run the two sequential chains
    double prod = 1.0;
    if (lane < 2) {
        const double* a = col + (2 + lane) * 64;
        prod = a[0];
        for (int i = 1; i < T; ++i) prod = __dmul_rn(prod, a[i]);
    }
    // np.sum of column 0 (pairwise_sum, n <= 128): n < 8 sequential; else 8 accumulators, tree, tail
    double sum = 0.0;
    if (T < 8) {
        if (lane == 0) {
            sum = col[0];
            for (int i = 1; i < T; ++i) sum = __dadd_rn(sum, col[i]);
        }
    } else {
        double r = 0.0;
        const int body = T - (T % 8);
        if (lane < 8) {
            r = col[lane];
            for (int i = 8; i < body; i += 8) r = __dadd_rn(r, col[i + lane]);
        }
        const double r0 = wave_bcast(r, 0), r1 = wave_bcast(r, 1), r2 = wave_bcast(r, 2), r3 = wave_bcast(r, 3);
        const double r4 = wave_bcast(r, 4), r5 = wave_bcast(r, 5), r6 = wave_bcast(r, 6), r7 = wave_bcast(r, 7);
        sum = __dadd_rn(__dadd_rn(__dadd_rn(r0, r1), __dadd_rn(r2, r3)), __dadd_rn(__dadd_rn(r4, r5), __dadd_rn(r6, r7)));
        for (int i = body; i < T; ++i) sum = __dadd_rn(sum, col[i]);
    }
    sum = wave_bcast(sum, 0);
    const double prod2 = wave_bcast(prod, 0), prod3 = wave_bcast(prod, 1);
    // serviceNum and np.min(column 1): exact reductions
    const unsigned long long real = __ballot(lane < T && q[0] > 0.0);
    const int n_real = __popcll(real);
    double mn = lane < T ? q[1] : INFINITY;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mn = fmin(mn, wave_bcast(mn, (lane + o) & 63));
    mn = wave_bcast(mn, 0);
    int violate = 0;
    if (prod2 < bounds[0] || prod2 > bounds[1]) ++violate;
    if (prod3 < bounds[2] || prod3 > bounds[3]) ++violate;
    double obj = sum / (double)n_real;          // (np.sum / serviceNum + 1 - np.min) / 2, one rounding per operation
    obj = __dadd_rn(obj, 1.0);
    obj = __dsub_rn(obj, mn);
    obj = obj / 2.0;
    __syncthreads();
    return __dadd_rn((double)violate, obj);
}

__device__ __forceinline__ void gather_row(const double* cand, int base, int len, int pos, double (&q)[4]) {
    const int idx = base + (pos < 0 ? pos + len : pos);      // Python list indexing
#pragma unroll
    for (int c = 0; c < 4; ++c) q[c] = cand[(size_t)idx * 4 + c];
}
}  // namespace

__global__ __launch_bounds__(64) void eswoa_kernel(int32_t T, const int32_t* __restrict__ cand_ptr,
                                                   const int32_t* __restrict__ len_init, const double* __restrict__ cand_g,
                                                   const double* __restrict__ bounds_g, const int32_t* __restrict__ start_pos,
                                                   int32_t pop, int32_t max_iter, const unsigned long long* __restrict__ seeds,
                                                   double* __restrict__ best_fitness, int32_t* __restrict__ best_pos_out,
                                                   double* __restrict__ history, long long* __restrict__ draws_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int p = blockIdx.x, lane = threadIdx.x;
    const int c0 = cand_ptr[(size_t)p * T], n_cand = cand_ptr[(size_t)p * T + T] - c0;
    double* col = reinterpret_cast<double*>(lds_raw);                 // [4][64]
    double* bounds = col + 256;                                       // [4]
    double* cand = bounds + 4;                                        // [n_cand][4]
    int* pos = reinterpret_cast<int*>(cand + (size_t)n_cand * 4);     // [pop][T]
    for (int i = lane; i < n_cand * 4; i += 64) cand[i] = cand_g[(size_t)c0 * 4 + i];
    if (lane < 4) bounds[lane] = bounds_g[(size_t)p * 4 + lane];
    const bool live = lane < T;
    const int base = live ? cand_ptr[(size_t)p * T + lane] - c0 : 0;
    const int len = live ? cand_ptr[(size_t)p * T + lane + 1] - cand_ptr[(size_t)p * T + lane] : 1;
    const int len0 = live ? len_init[(size_t)p * T + lane] : 1;
    const unsigned long long seed = seeds[p];
    unsigned long long k = 0;                                         // draws consumed so far (wave-uniform)
    __syncthreads();

    // initial population (WOA.py:51-52): individual i, category j <- draw k + i*T + j + 1, lengths BEFORE the append
    for (int i = 0; i < pop; ++i)
        if (live) pos[i * T + lane] = draw_below(seed, k + (unsigned long long)i * T + lane + 1, len0);
    k += (unsigned long long)pop * T;
    __syncthreads();

    double q[4] = {0.0, 0.0, 0.0, 0.0};
    double best_fit = 3.0;                                            // :71
    int best = 0;                                                     // this lane's category of the recorded best
    int alias = -1;                                                   // individual whose list the record shares
    if (start_pos[(size_t)p * T] >= 0) {                              // :55-69
        best = live ? start_pos[(size_t)p * T + lane] : 0;
        if (live) gather_row(cand, base, len, best, q);
        best_fit = figure_of_merit(q, T, lane, col, bounds);
    }
    for (int i = 0; i < pop; ++i) {                                   // :77-85
        const int x = live ? pos[i * T + lane] : 0;
        if (live) gather_row(cand, base, len, x, q);
        const double f = figure_of_merit(q, T, lane, col, bounds);
        if (best_fit > f) {
            best_fit = f;
            best = x;
            alias = i;
        }
    }

    for (int t = 0; t < max_iter; ++t) {                              // :107-161
        const double prob = __dmul_rn(0.2, __dsub_rn(1.0, (double)t / (double)max_iter));
        for (int i = 0; i < pop; ++i) {                               // global phase
            if (draw_uniform(seed, ++k) < prob) {
                const int j = draw_below(seed, ++k, T);
                const int kk = draw_below(seed, ++k, __shfl(len, j));
                if (lane == j) {
                    pos[i * T + lane] = kk;
                    if (alias == i) best = kk;                        // same list object in the reference
                }
                __syncthreads();
                const int x = live ? pos[i * T + lane] : 0;
                if (live) gather_row(cand, base, len, x, q);
                const double f = figure_of_merit(q, T, lane, col, bounds);
                if (best_fit > f) {
                    best_fit = f;
                    best = x;
                    alias = i;
                }
            }
        }
        if (PE > draw_uniform(seed, ++k)) {                           // :125-129
            if (lane == 0) history[(size_t)p * max_iter + t] = best_fit;
            continue;
        }
        const double a = __dsub_rn(2.0, __dmul_rn(2.0, (double)t) / (double)max_iter);
        for (int i = 0; i < pop; ++i) {                               // local phase
            const double r = draw_uniform(seed, ++k);
            const double A = __dsub_rn(__dmul_rn(__dmul_rn(2.0, a), r), a);
            const double C = __dmul_rn(2.0, r);
            const double l = draw_uniform(seed, ++k);
            const double pp = draw_uniform(seed, ++k);
            const int x = live ? pos[i * T + lane] : 0;
            bool moved = false;
            double nv = 0.0;
            if (pp < 0.5) {
                if (fabs(A) < 1.0) {                                  // round(b - A * (C*b - x))
                    moved = true;
                    nv = __dsub_rn((double)best, __dmul_rn(A, __dsub_rn(__dmul_rn(C, (double)best), (double)x)));
                }
            } else {                                                  // round((x - b) * e^l * cos(2 pi l) + b)
                moved = true;
                const double e = exp(l), c = cos(__dmul_rn(__dmul_rn(2.0, 3.141592653589793), l));
                nv = __dadd_rn(__dmul_rn(__dmul_rn((double)(x - best), e), c), (double)best);
            }
            if (moved) {
                long long nx = (long long)rint(nv);                   // Python round: half to even
                if (llabs(nx) >= len) {                               // Python %: sign of the divisor
                    nx %= len;
                    if (nx < 0) nx += len;
                }
                if (live) pos[i * T + lane] = (int)nx;
                if (alias == i) alias = -1;                           // rebinding: the record keeps the old list
                __syncthreads();
                if (live) gather_row(cand, base, len, (int)nx, q);
                const double f = figure_of_merit(q, T, lane, col, bounds);
                if (best_fit > f) {
                    best_fit = f;
                    best = (int)nx;
                    alias = i;
                }
            }
        }
        if (lane == 0) history[(size_t)p * max_iter + t] = best_fit;
    }
    if (live) best_pos_out[(size_t)p * T + lane] = best;
    if (lane == 0) {
        best_fitness[p] = best_fit;
        draws_out[p] = (long long)k;
    }
}

// ---- T > 64: one WORKGROUP (4 waves) per problem, thread tid owns categories tid, tid + 256, ... ---------------------------
// Same search, same draws, same float64 evaluation orders; what changes is where things live.  The positions of the
// population (pop x T int32: 800 KB at T = 2000, pop = 100) and the candidate table stay in global memory (L2), the
// recorded best IS the output row; LDS holds the three QoS columns of the composition under evaluation (np.sum needs
// column 0 as a whole — numpy's pairwise recursion splits at n/2 rounded down to a multiple of 8 until a block has at most
// 128 elements — and np.cumprod columns 2 and 3 in order) and the per-category base / length tables.  One evaluation = a
// parallel gather, then the two sequential product chains (lanes 0 and 1 of wave 0: T dependent multiplies each, the
// floor of this kernel — bit parity with np.cumprod leaves no other order) while wave 1 forms the pairwise sum.
namespace {
constexpr int WNT = 256;

// numpy's pairwise block (n <= 128) by one wave: result in every lane
__device__ double pw_leaf(const double* a, int n, int lane) {
    double sum = 0.0;
    if (n < 8) {
        if (lane == 0) {
            sum = a[0];
            for (int i = 1; i < n; ++i) sum = __dadd_rn(sum, a[i]);
        }
        return wave_bcast(sum, 0);
    }
    double r = 0.0;
    const int body = n - (n % 8);
    if (lane < 8) {
        r = a[lane];
        for (int i = 8; i < body; i += 8) r = __dadd_rn(r, a[i + lane]);
    }
    const double r0 = wave_bcast(r, 0), r1 = wave_bcast(r, 1), r2 = wave_bcast(r, 2), r3 = wave_bcast(r, 3);
    const double r4 = wave_bcast(r, 4), r5 = wave_bcast(r, 5), r6 = wave_bcast(r, 6), r7 = wave_bcast(r, 7);
    sum = __dadd_rn(__dadd_rn(__dadd_rn(r0, r1), __dadd_rn(r2, r3)), __dadd_rn(__dadd_rn(r4, r5), __dadd_rn(r6, r7)));
    for (int i = body; i < n; ++i) sum = __dadd_rn(sum, a[i]);
    return sum;
}
// np.sum of n doubles (pairwise_sum): post-order walk of the recursion with an explicit stack (depth <= log2(n / 128) + 1)
__device__ double pw_sum(const double* a, int n, int lane) {
    int off[24], len[24], st[24];
    double left[24];
    int sp = 0;
    off[0] = 0; len[0] = n; st[0] = 0; left[0] = 0.0;
    double ret = 0.0;
    sp = 1;
    while (sp > 0) {
        const int f = sp - 1;
        if (len[f] <= 128) {
            ret = pw_leaf(a + off[f], len[f], lane);
            --sp;
        } else {
            int n2 = len[f] / 2;
            n2 -= n2 % 8;
            if (st[f] == 0) {
                st[f] = 1;
                off[sp] = off[f]; len[sp] = n2; st[sp] = 0;
                ++sp;
            } else if (st[f] == 1) {
                left[f] = ret;
                st[f] = 2;
                off[sp] = off[f] + n2; len[sp] = len[f] - n2; st[sp] = 0;
                ++sp;
            } else {
                ret = __dadd_rn(left[f], ret);
                --sp;
            }
        }
    }
    return ret;
}

struct WideLds {
    double* col0;      // [T] column 0 (np.sum)
    double* col2;      // [T] column 2 (np.cumprod)
    double* col3;      // [T]
    double* red;       // [8]: min per wave (4), results: sum, prod2, prod3
    int* cnt;          // [4] real services per wave
    int* base;         // [T]
    int* len;          // [T]
    double* bounds;    // [4]
};

// figure of merit of the composition pos[j] (j < T), every thread returns it
__device__ double wide_merit(const WideLds& L, const int* pos, const double* cand, int T, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    double mn = INFINITY;
    int real = 0;
    for (int j = tid; j < T; j += WNT) {
        const int x = pos[j], ln = L.len[j];
        const double* q = cand + (size_t)(L.base[j] + (x < 0 ? x + ln : x)) * 4;
        const double q0 = q[0], q1 = q[1];
        L.col0[j] = q0;
        L.col2[j] = q[2];
        L.col3[j] = q[3];
        mn = fmin(mn, q1);
        real += q0 > 0.0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fmin(mn, wave_bcast(mn, (lane + o) & 63));
        real += __shfl(real, (lane + o) & 63);
    }
    if (lane == 0) {
        L.red[wave] = mn;
        L.cnt[wave] = real;
    }
    __syncthreads();
    if (wave == 0) {
        if (lane < 2) {                                   // np.cumprod: one sequential chain per column
            const double* a = lane == 0 ? L.col2 : L.col3;
            double prod = a[0];
            for (int i = 1; i < T; ++i) prod = __dmul_rn(prod, a[i]);
            L.red[5 + lane] = prod;
        }
    } else if (wave == 1) {
        const double sum = pw_sum(L.col0, T, lane);
        if (lane == 0) L.red[4] = sum;
    }
    __syncthreads();
    const double sum = L.red[4], prod2 = L.red[5], prod3 = L.red[6];
    mn = fmin(fmin(L.red[0], L.red[1]), fmin(L.red[2], L.red[3]));
    const int n_real = L.cnt[0] + L.cnt[1] + L.cnt[2] + L.cnt[3];
    int violate = 0;
    if (prod2 < L.bounds[0] || prod2 > L.bounds[1]) ++violate;
    if (prod3 < L.bounds[2] || prod3 > L.bounds[3]) ++violate;
    double obj = sum / (double)n_real;
    obj = __dadd_rn(obj, 1.0);
    obj = __dsub_rn(obj, mn);
    obj = obj / 2.0;
    __syncthreads();                                      // the columns and red[] are free again
    return __dadd_rn((double)violate, obj);
}
}  // namespace

__global__ __launch_bounds__(WNT) void eswoa_wide_kernel(int32_t T, const int32_t* __restrict__ cand_ptr,
                                                         const int32_t* __restrict__ len_init, const double* __restrict__ cand_g,
                                                         const double* __restrict__ bounds_g, const int32_t* __restrict__ start_pos,
                                                         int32_t pop, int32_t max_iter, const unsigned long long* __restrict__ seeds,
                                                         int32_t* __restrict__ pos_ws, double* __restrict__ best_fitness,
                                                         int32_t* __restrict__ best_pos_out, double* __restrict__ history,
                                                         long long* __restrict__ draws_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int p = blockIdx.x, tid = threadIdx.x;
    WideLds L;
    L.col0 = reinterpret_cast<double*>(lds_raw);
    L.col2 = L.col0 + T;
    L.col3 = L.col2 + T;
    L.red = L.col3 + T;
    L.bounds = L.red + 8;
    L.cnt = reinterpret_cast<int*>(L.bounds + 4);
    L.base = L.cnt + 4;
    L.len = L.base + T;
    const int c0 = cand_ptr[(size_t)p * T];
    const double* cand = cand_g + (size_t)c0 * 4;
    int* pos = pos_ws + (size_t)p * pop * T;              // [pop][T]
    int* best = best_pos_out + (size_t)p * T;             // the recorded best composition (a COPY: see `alias`)
    for (int j = tid; j < T; j += WNT) {
        L.base[j] = cand_ptr[(size_t)p * T + j] - c0;
        L.len[j] = cand_ptr[(size_t)p * T + j + 1] - cand_ptr[(size_t)p * T + j];
    }
    if (tid < 4) L.bounds[tid] = bounds_g[(size_t)p * 4 + tid];
    const unsigned long long seed = seeds[p];
    unsigned long long k = 0;

    // initial population (WOA.py:51-52): individual i, category j <- draw k + i*T + j + 1, lengths BEFORE the append
    for (int i = 0; i < pop; ++i)
        for (int j = tid; j < T; j += WNT)
            pos[(size_t)i * T + j] = draw_below(seed, k + (unsigned long long)i * T + j + 1, len_init[(size_t)p * T + j]);
    k += (unsigned long long)pop * T;
    __syncthreads();

    double best_fit = 3.0;                                            // :71
    int alias = -1;                                                   // individual whose list the record shares
    auto record = [&](int i) {                                        // best <- a copy of individual i
        for (int j = tid; j < T; j += WNT) best[j] = pos[(size_t)i * T + j];
        alias = i;
    };
    if (start_pos[(size_t)p * T] >= 0) {                              // :55-69
        for (int j = tid; j < T; j += WNT) best[j] = start_pos[(size_t)p * T + j];
        __syncthreads();
        best_fit = wide_merit(L, best, cand, T, tid);
    } else {
        for (int j = tid; j < T; j += WNT) best[j] = 0;
    }
    for (int i = 0; i < pop; ++i) {                                   // :77-85
        const double f = wide_merit(L, pos + (size_t)i * T, cand, T, tid);
        if (best_fit > f) {
            best_fit = f;
            record(i);
        }
    }
    __syncthreads();

    for (int t = 0; t < max_iter; ++t) {                              // :107-161
        const double prob = __dmul_rn(0.2, __dsub_rn(1.0, (double)t / (double)max_iter));
        for (int i = 0; i < pop; ++i) {                               // global phase
            if (draw_uniform(seed, ++k) < prob) {
                const int j = draw_below(seed, ++k, T);
                const int kk = draw_below(seed, ++k, L.len[j]);
                if (tid == 0) {
                    pos[(size_t)i * T + j] = kk;
                    if (alias == i) best[j] = kk;                     // same list object in the reference
                }
                __syncthreads();
                const double f = wide_merit(L, pos + (size_t)i * T, cand, T, tid);
                if (best_fit > f) {
                    best_fit = f;
                    record(i);
                    __syncthreads();
                }
            }
        }
        if (PE > draw_uniform(seed, ++k)) {                           // :125-129
            if (tid == 0) history[(size_t)p * max_iter + t] = best_fit;
            continue;
        }
        const double a = __dsub_rn(2.0, __dmul_rn(2.0, (double)t) / (double)max_iter);
        for (int i = 0; i < pop; ++i) {                               // local phase
            const double r = draw_uniform(seed, ++k);
            const double A = __dsub_rn(__dmul_rn(__dmul_rn(2.0, a), r), a);
            const double C = __dmul_rn(2.0, r);
            const double l = draw_uniform(seed, ++k);
            const double pp = draw_uniform(seed, ++k);
            const bool spiral = !(pp < 0.5);
            if (!spiral && !(fabs(A) < 1.0)) continue;
            const double e = exp(l), cs = cos(__dmul_rn(__dmul_rn(2.0, 3.141592653589793), l));
            for (int j = tid; j < T; j += WNT) {
                const int x = pos[(size_t)i * T + j], b = best[j], ln = L.len[j];
                double nv;
                if (!spiral) nv = __dsub_rn((double)b, __dmul_rn(A, __dsub_rn(__dmul_rn(C, (double)b), (double)x)));   // round(b - A (C b - x))
                else nv = __dadd_rn(__dmul_rn(__dmul_rn((double)(x - b), e), cs), (double)b);                             // round((x - b) e^l cos(2 pi l) + b)
                long long nx = (long long)rint(nv);                   // Python round: half to even
                if (llabs(nx) >= ln) {                                // Python %: sign of the divisor
                    nx %= ln;
                    if (nx < 0) nx += ln;
                }
                pos[(size_t)i * T + j] = (int)nx;
            }
            if (alias == i) alias = -1;                               // rebinding: the record keeps the old list
            __syncthreads();
            const double f = wide_merit(L, pos + (size_t)i * T, cand, T, tid);
            if (best_fit > f) {
                best_fit = f;
                record(i);
                __syncthreads();
            }
        }
        if (tid == 0) history[(size_t)p * max_iter + t] = best_fit;
    }
    if (tid == 0) {
        best_fitness[p] = best_fit;
        draws_out[p] = (long long)k;
    }
}

static size_t eswoa_wide_lds_bytes(int T) { return ((size_t)3 * T + 12) * sizeof(double) + ((size_t)2 * T + 4) * sizeof(int); }

extern "C" int64_t gnnpn_eswoa_wide_workspace_bytes(int32_t P, int32_t T, int32_t pop) {
    return (int64_t)(P > 0 ? P : 0) * (int64_t)(pop > 0 ? pop : 0) * (int64_t)(T > 0 ? T : 0) * (int64_t)sizeof(int32_t);
}

extern "C" int gnnpn_eswoa_wide_f64(int32_t P, int32_t T, const int32_t* cand_ptr, const int32_t* len_init, const double* cand,
                                    const double* bounds, const int32_t* start_pos, int32_t pop, int32_t max_iter,
                                    const uint64_t* seeds, void* workspace, int64_t workspace_bytes, double* best_fitness,
                                    int32_t* best_pos, double* history, int64_t* draws, void* stream) {
    GNNPN_REQUIRE(cand_ptr && len_init && cand && bounds && start_pos && seeds && best_fitness && best_pos && history && draws,
                  "eswoa_wide: null operand");
    GNNPN_REQUIRE(P >= 0 && T >= 1 && pop > 0 && max_iter >= 0, "eswoa_wide: bad argument");
    if (P == 0) return GNNPN_OK;
    GNNPN_REQUIRE(workspace && workspace_bytes >= gnnpn_eswoa_wide_workspace_bytes(P, T, pop), "eswoa_wide: workspace too small");
    const size_t lds = eswoa_wide_lds_bytes(T);
    if (lds > 160 * 1024 - 1024)
        GNNPN_FAIL(GNNPN_E_UNSUP, "eswoa_wide: T=%d categories need %zu B of LDS for the three QoS columns (a CU has 160 KB)", T, lds);
    if (hipFuncSetAttribute((const void*)eswoa_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        GNNPN_FAIL(GNNPN_E_LAUNCH, "eswoa_wide: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL(eswoa_wide_kernel, dim3(P), dim3(WNT), lds, (hipStream_t)stream, T, cand_ptr, len_init, cand, bounds,
                       start_pos, pop, max_iter, reinterpret_cast<const unsigned long long*>(seeds),
                       reinterpret_cast<int32_t*>(workspace), best_fitness, best_pos, history, reinterpret_cast<long long*>(draws));
    GNNPN_CHECK_LAUNCH("eswoa_wide_f64");
    return GNNPN_OK;
}

// LDS bytes one problem needs (host side): scratch + bounds + its candidate table + the population's positions
static size_t eswoa_lds_bytes(int n_cand, int pop, int T) {
    return (256 + 4) * sizeof(double) + (size_t)n_cand * 4 * sizeof(double) + (size_t)pop * T * sizeof(int);
}

extern "C" int gnnpn_eswoa_f64(int32_t P, int32_t T, const int32_t* cand_ptr, const int32_t* len_init, const double* cand,
                               const double* bounds, const int32_t* start_pos, int32_t pop, int32_t max_iter,
                               const uint64_t* seeds, int32_t max_cand, double* best_fitness, int32_t* best_pos,
                               double* history, int64_t* draws, void* stream) {
    GNNPN_REQUIRE(cand_ptr && len_init && cand && bounds && start_pos && seeds && best_fitness && best_pos && history && draws,
                  "eswoa: null operand");
    GNNPN_REQUIRE(P >= 0 && pop > 0 && max_iter >= 0 && max_cand > 0, "eswoa: bad argument");
    if (T < 1 || T > 64) GNNPN_FAIL(GNNPN_E_UNSUP, "eswoa: T=%d categories (this form maps one category to one lane: 1..64; gnnpn_eswoa_wide_f64 takes any T)", T);
    if (P == 0) return GNNPN_OK;
    const size_t lds = eswoa_lds_bytes(max_cand, pop, T);
    if (lds > 160 * 1024 - 1024)
        GNNPN_FAIL(GNNPN_E_UNSUP, "eswoa: %zu B of LDS per problem (population %d x %d, %d candidates) exceed a CU", lds, pop, T, max_cand);
    if (hipFuncSetAttribute((const void*)eswoa_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        GNNPN_FAIL(GNNPN_E_LAUNCH, "eswoa: cannot reserve %zu B of LDS", lds);
    hipLaunchKernelGGL(eswoa_kernel, dim3(P), dim3(64), lds, (hipStream_t)stream, T, cand_ptr, len_init, cand, bounds, start_pos,
                       pop, max_iter, reinterpret_cast<const unsigned long long*>(seeds), best_fitness, best_pos, history,
                       reinterpret_cast<long long*>(draws));
    GNNPN_CHECK_LAUNCH("eswoa_f64");
    return GNNPN_OK;
}
