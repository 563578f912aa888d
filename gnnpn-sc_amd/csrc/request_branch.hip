// The whole workflow (GIN) branch of Net.forward for small workflow graphs in ONE launch:
//   embedding lookup + concat -> n_layers x { GIN aggregate ; Linear+BN+ReLU ; Linear+BN+ReLU } -> nodeLin -> mean over the
//   graph's nodes                      (/root/reference/src/models/modelML.py:133-143,165-166)
// One 256-thread workgroup per graph (<= 16 nodes: the MFMA M dimension; QWS / Normal requests have <= 11).  The node
// features never leave LDS between the stages; the weights stream from L2 as ready-made v_mfma_f32_16x16x4_f32
// B-fragments (packed once on the host: a layout change, no arithmetic), 1 KiB per wave-load.
// Numerics: stage for stage the arithmetic of the separate kernels it replaces (embed_concat_kernel, csr_aggregate_kernel,
// linear_f32_kernel, segment_mean_kernel): sequential CSR-order sums, k-ascending fma chains from 0 (the fp32 MFMA is
// exactly that chain), bias / BN-affine / ReLU epilogues with separately rounded multiply and add, sum / count mean —
// so the result is bit-identical to the layered path (tests/test_gpu_ops.py::test_request_branch_equals_layered).
#include "common.h"

typedef float rb_f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int ROWS = 16;          // nodes per graph (MFMA M)
constexpr int HID = 128;          // hiddenChannels (environment.ini:4,13)
constexpr int LD = 258;           // LDS row stride in floats: (row*2 + kq) distinct banks for the A-fragment reads
constexpr int MAX_LAYERS = 4;
constexpr int MAX_LDS_EDGES = 256;

struct GinLayer {
    const float* w0p;             // packed [2H/16][K16][64][4]
    const float* b0;
    const float* a1;
    const float* s1;
    const float* w3p;             // packed [H/16][2H/16][64][4]
    const float* b3;
    const float* a2;
    const float* s2;
    const float* eps;
};
struct BranchArgs {
    GinLayer layer[MAX_LAYERS];
    const float* x;               // [N, 1 + nfeat]
    const float* table;           // [vocab, emb]
    const int32_t* rowptr;
    const int32_t* col;
    const int32_t* seg;
    const float* linp;            // packed nodeLin [H/16][H/16][64][4]
    const float* linb;
    float* out;                   // [B, H]
    int32_t n_graphs, n_layers, nfeat, vocab, emb;
};

// The B-fragments of one GEMM for this wave's NT column tiles: ALL of them are requested at once (K16 x NT 1-KiB
// wave-loads in flight) and, because they do not depend on data, one whole stage BEFORE the MFMAs that use them — under the
// previous stage's arithmetic, barrier and aggregation — so the weight stream's L2 latency is off the critical path.
template <int NT, int K16>
struct Frags {
    float4 b[K16][NT];
};
template <int NT, int K16>
__device__ __forceinline__ void load_frags(const float* __restrict__ wp, int tile0, int lane, Frags<NT, K16>& f) {
    const float4* wv = reinterpret_cast<const float4*>(wp);
#pragma unroll
    for (int kb = 0; kb < K16; ++kb)
#pragma unroll
        for (int t = 0; t < NT; ++t) f.b[kb][t] = wv[((size_t)(tile0 + t) * K16 + kb) * 64 + lane];
}
// acc[t] = in[16][16*K16] . W^T for this wave's tiles: k-ascending fma chains from 0 (v_mfma_f32_16x16x4_f32)
template <int NT, int K16>
__device__ __forceinline__ void mma_frags(const float* in, int lane, const Frags<NT, K16>& f, rb_f32x4 (&acc)[NT]) {
    const int row = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = rb_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < K16; ++kb) {
        const float* a = in + row * LD + 16 * kb + kq;
        const float a0 = a[0], a1 = a[4], a2 = a[8], a3 = a[12];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, f.b[kb][t].x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, f.b[kb][t].y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, f.b[kb][t].z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, f.b[kb][t].w, acc[t], 0, 0, 0);
        }
    }
}

// epilogue operands of this lane's NT columns, requested together with the stage's weight fragments (off the critical path)
template <int NT>
struct Epi {
    float b[NT], sc[NT], sh[NT];
};
template <int NT>
__device__ __forceinline__ void load_epi(const float* __restrict__ bias, const float* __restrict__ scale,
                                         const float* __restrict__ shift, int tile0, int lane, Epi<NT>& e) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int colx = (tile0 + t) * 16 + (lane & 15);
        e.b[t] = bias[colx];
        e.sc[t] = scale ? scale[colx] : 1.0f;
        e.sh[t] = scale ? shift[colx] : 0.0f;
    }
}
// epilogue of linear_f32_kernel: (+bias) (*scale + shift) (relu) -> LDS tile; accumulator register r = row 4*kq + r, col c
template <int NT>
__device__ __forceinline__ void store_tiles(float* dst, int tile0, int lane, const rb_f32x4 (&acc)[NT], const Epi<NT>& e,
                                            bool affine, bool relu) {
    const int c = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int colx = (tile0 + t) * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = __fadd_rn(acc[t][r], e.b[t]);
            if (affine) v = __fadd_rn(__fmul_rn(v, e.sc[t]), e.sh[t]);
            if (relu) v = v < 0.0f ? 0.0f : v;
            dst[(4 * kq + r) * LD + colx] = v;
        }
    }
}
}  // namespace

// GIN aggregate (csr_aggregate_kernel, w == NULL, self_coef): sum of the in-neighbours in CSR order, then + (1 + eps) * own
// row; h -> t, zero outside [cnt) x [cin).  The graph's CSR slice sits in LDS (s_rp / s_col) when it fits, else in global.
__device__ __forceinline__ void gin_aggregate(const float* h, float* t, int cnt, int cin, int cpad, float ope, const int* s_rp,
                                              const int* s_col, const int32_t* __restrict__ g_col, int n0, int e_base,
                                              bool csr_in_lds) {
    for (int i = threadIdx.x; i < ROWS * cpad; i += 256) {
        const int r = i / cpad, c = i - r * cpad;
        float acc = 0.0f;
        if (r < cnt && c < cin) {
            const int e0 = s_rp[r], e1 = s_rp[r + 1];
            if (csr_in_lds) {
                for (int e = e0; e < e1; ++e) acc = __fadd_rn(acc, h[s_col[e - e_base] * LD + c]);
            } else {
                for (int e = e0; e < e1; ++e) acc = __fadd_rn(acc, h[(g_col[e] - n0) * LD + c]);
            }
            acc = __fadd_rn(acc, __fmul_rn(ope, h[r * LD + c]));
        }
        t[r * LD + c] = acc;
    }
}

template <int NL>   // number of GIN layers: compile-time, so that every stage's weight fragments have a plain straight-line lifetime
__global__ __launch_bounds__(256, 1) void gin_request_branch_kernel(BranchArgs a) {
    __shared__ float bufA[ROWS * LD];
    __shared__ float bufB[ROWS * LD];
    __shared__ int s_rp[ROWS + 1];
    __shared__ int s_col[MAX_LDS_EDGES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x;
    const int n0 = a.seg[g], cnt = a.seg[g + 1] - n0;   // cnt <= 16 (checked by the host)
    const int cin0 = a.emb + a.nfeat;                     // 26
    // the first GEMM's weights are on their way before anything else happens
    Frags<4, 2> f0;
    Epi<4> e0;
    load_frags<4, 2>(a.layer[0].w0p, wave * 4, lane, f0);
    load_epi<4>(a.layer[0].b0, a.layer[0].a1, a.layer[0].s1, wave * 4, lane, e0);
    if (tid <= ROWS) s_rp[tid] = a.rowptr[n0 + min(tid, cnt)];
    const int e_base = a.rowptr[n0], n_edges = a.rowptr[n0 + cnt] - e_base;
    const bool csr_in_lds = n_edges <= MAX_LDS_EDGES;
    if (csr_in_lds)
        for (int e = tid; e < n_edges; e += 256) s_col[e] = a.col[e_base + e] - n0;
    // ---- embedding lookup + concat (embed_concat_kernel) -> bufA[r][0..cin0), zero padded to 32 columns and to 16 rows
    // (zero operands add exactly nothing to an fma chain)
    for (int i = tid; i < ROWS * 32; i += 256) {
        const int r = i >> 5, c = i & 31;
        float v = 0.0f;
        if (r < cnt && c < cin0) {
            const float* xr = a.x + (int64_t)(n0 + r) * (1 + a.nfeat);
            if (c < a.emb) {
                const int id = (int)xr[0];
                v = (id >= 0 && id < a.vocab) ? a.table[(int64_t)id * a.emb + c] : __int_as_float(0x7fc00000);
            } else {
                v = xr[1 + c - a.emb];
            }
        }
        bufA[r * LD + c] = v;
    }
    __syncthreads();
    float* h = bufA;        // current node features
    float* t = bufB;        // scratch
    // ================= layer 0 (in = emb + nfeat <= 32 columns)
    Frags<2, 16> f3;
    Epi<2> e3;
    {
        const GinLayer& L = a.layer[0];
        load_frags<2, 16>(L.w3p, wave * 2, lane, f3);
        load_epi<2>(L.b3, L.a2, L.s2, wave * 2, lane, e3);
        gin_aggregate(h, t, cnt, cin0, 32, __fadd_rn(1.0f, *L.eps), s_rp, s_col, a.col, n0, e_base, csr_in_lds);
        __syncthreads();
        rb_f32x4 acc[4];
        mma_frags<4, 2>(t, lane, f0, acc);                                   // Linear(in -> 2H) + BN + ReLU : t -> h
        store_tiles<4>(h, wave * 4, lane, acc, e0, true, true);
        __syncthreads();
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const GinLayer& L = a.layer[l];
        const bool last = l + 1 == NL;
        rb_f32x4 acc2[2];
        mma_frags<2, 16>(h, lane, f3, acc2);                                 // Linear(2H -> H) + BN + ReLU : h -> t
        __builtin_amdgcn_sched_barrier(0);                                   // keep the next loads BEHIND these MFMAs (registers)
        // f3's registers are free again: the NEXT stage's weights (layer l+1's first Linear, or nodeLin; both K = H) go out
        // now and travel under the epilogue, the barrier and the aggregation
        Frags<4, 8> fn;
        Frags<2, 8> fl;
        Epi<4> en;
        Epi<2> el;
        if (!last) {
            load_frags<4, 8>(a.layer[l + 1].w0p, wave * 4, lane, fn);
            load_epi<4>(a.layer[l + 1].b0, a.layer[l + 1].a1, a.layer[l + 1].s1, wave * 4, lane, en);
        } else {
            load_frags<2, 8>(a.linp, wave * 2, lane, fl);
            load_epi<2>(a.linb, nullptr, nullptr, wave * 2, lane, el);
        }
        store_tiles<2>(t, wave * 2, lane, acc2, e3, true, true);
        __syncthreads();
        if (last) {                                                          // nodeLin (modelML.py:165): t -> h
            rb_f32x4 acc[2];
            mma_frags<2, 8>(t, lane, fl, acc);
            store_tiles<2>(h, wave * 2, lane, acc, el, false, false);
            __syncthreads();
            break;
        }
        const GinLayer& N = a.layer[l + 1];
        gin_aggregate(t, h, cnt, HID, HID, __fadd_rn(1.0f, *N.eps), s_rp, s_col, a.col, n0, e_base, csr_in_lds);   // t -> h
        __syncthreads();
        rb_f32x4 acc[4];
        mma_frags<4, 8>(h, lane, fn, acc);                                   // Linear(H -> 2H) + BN + ReLU : h -> (h)
        __builtin_amdgcn_sched_barrier(0);
        load_frags<2, 16>(N.w3p, wave * 2, lane, f3);                        // fn's registers are free: next Linear(2H -> H)
        load_epi<2>(N.b3, N.a2, N.s2, wave * 2, lane, e3);
        __syncthreads();                                                     // every wave has read h before it is rewritten
        store_tiles<4>(h, wave * 4, lane, acc, en, true, true);
        __syncthreads();
    }
    // ---- mean over the graph's nodes (segment_mean_kernel): sequential sum / count
    if (tid < HID) {
        float acc = 0.0f;
        for (int r = 0; r < cnt; ++r) acc = __fadd_rn(acc, h[r * LD + tid]);
        a.out[(int64_t)g * HID + tid] = acc / (float)max(cnt, 1);
    }
}

extern "C" int gnnpn_request_branch_f32(const float* x, int32_t nfeat, const float* table, int32_t vocab, int32_t emb,
                                        const int32_t* rowptr, const int32_t* col, const int32_t* seg_ptr,
                                        int32_t n_graphs, int32_t max_nodes, int32_t n_layers,
                                        const gnnpn_gin_layer_t* layers, int32_t hidden, const float* lin_w_packed,
                                        const float* lin_b, float* out, void* stream) {
    GNNPN_REQUIRE(n_graphs >= 0 && n_layers >= 1 && nfeat >= 0 && vocab > 0 && emb > 0, "request_branch: bad shape");
    if (n_graphs == 0) return GNNPN_OK;
    GNNPN_REQUIRE(x && table && rowptr && seg_ptr && layers && lin_w_packed && lin_b && out,
                  "request_branch: null operand");   // col may be NULL when no graph has an edge
    if (hidden != HID || n_layers > MAX_LAYERS || emb + nfeat > 32 || max_nodes > ROWS)
        GNNPN_FAIL(GNNPN_E_UNSUP, "request_branch: built for hidden = %d, <= %d layers, <= 32 input features and graphs of "
                   "<= %d nodes (got hidden %d, %d layers, %d features, %d nodes): use the layered kernels", HID,
                   MAX_LAYERS, ROWS, hidden, n_layers, emb + nfeat, max_nodes);
    if (n_graphs == 0) return GNNPN_OK;
    BranchArgs a{};
    for (int l = 0; l < n_layers; ++l) {
        const gnnpn_gin_layer_t& s = layers[l];
        GNNPN_REQUIRE(s.w0_packed && s.b0 && s.bn1_scale && s.bn1_shift && s.w3_packed && s.b3 && s.bn2_scale &&
                      s.bn2_shift && s.eps, "request_branch: null operand in layer %d", l);
        GNNPN_REQUIRE(gnnpn_aligned(s.w0_packed, 16) && gnnpn_aligned(s.w3_packed, 16), "request_branch: packed weights must be 16-byte aligned");
        a.layer[l] = GinLayer{s.w0_packed, s.b0, s.bn1_scale, s.bn1_shift, s.w3_packed, s.b3, s.bn2_scale, s.bn2_shift, s.eps};
    }
    GNNPN_REQUIRE(gnnpn_aligned(lin_w_packed, 16), "request_branch: packed weights must be 16-byte aligned");
    a.x = x;
    a.table = table;
    a.rowptr = rowptr;
    a.col = col;
    a.seg = seg_ptr;
    a.linp = lin_w_packed;
    a.linb = lin_b;
    a.out = out;
    a.n_graphs = n_graphs;
    a.n_layers = n_layers;
    a.nfeat = nfeat;
    a.vocab = vocab;
    a.emb = emb;
    hipStream_t st = (hipStream_t)stream;
    switch (n_layers) {
        case 1: hipLaunchKernelGGL(gin_request_branch_kernel<1>, dim3(n_graphs), dim3(256), gnnpn_front_lds_pad((const void*)gin_request_branch_kernel<1>), st, a); break;
        case 2: hipLaunchKernelGGL(gin_request_branch_kernel<2>, dim3(n_graphs), dim3(256), gnnpn_front_lds_pad((const void*)gin_request_branch_kernel<2>), st, a); break;
        case 3: hipLaunchKernelGGL(gin_request_branch_kernel<3>, dim3(n_graphs), dim3(256), gnnpn_front_lds_pad((const void*)gin_request_branch_kernel<3>), st, a); break;
        default: hipLaunchKernelGGL(gin_request_branch_kernel<4>, dim3(n_graphs), dim3(256), gnnpn_front_lds_pad((const void*)gin_request_branch_kernel<4>), st, a); break;
    }
    GNNPN_CHECK_LAUNCH("request_branch_f32");
    return GNNPN_OK;
}
