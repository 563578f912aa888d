// Pointer-network LSTM encoder recurrence (v1: one workgroup owns BT problems for all L steps;
// no inter-workgroup communication).
//
// Layout in HBM: pregates [B,L,4H] (gate-major i,f,g,o), enc_out [B,L,H], packed recurrent
// weights Wp[k/4][gate][j] as float4 = W_hh[gate*H + j][k..k+3] so that for a fixed (k/4, gate)
// the 64 lanes of a wave read 1 KiB contiguous.  Thread j owns hidden unit j: 4 gate dot
// products per problem as ONE k-ordered fmaf chain each (the arithmetic of an fp32 MFMA
// accumulation), h broadcast from LDS, c kept in registers, one barrier per step.
// W_hh (1 MiB at H=256) is re-streamed from L2 every step; the step is bound by that stream.
#include "common.h"
#include "recurrent.h"
#include "lstm_shared.h"

template <int H, int BT>
__global__ __launch_bounds__(H) void lstm_encode_kernel(LstmNets nets, int32_t B, int32_t L) {
    __shared__ __attribute__((aligned(16))) float hs[2][BT][H];
    const int net = blockIdx.y;
    const int b0 = blockIdx.x * BT;
    const int j = threadIdx.x;
    const float* __restrict__ pre = nets.pregates[net];
    const float4* __restrict__ Wp = reinterpret_cast<const float4*>(nets.whh[net]);
    float* __restrict__ enc = nets.enc_out[net];

    float bh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bh[g] = nets.bhh[net][g * H + j];
    float c[BT], h[BT];
#pragma unroll
    for (int p = 0; p < BT; ++p) {
        c[p] = 0.0f;
        h[p] = 0.0f;
        hs[0][p][j] = 0.0f;
    }
    __syncthreads();

    int cur = 0;
    for (int t = 0; t < L; ++t) {
        float pg[BT][4];
#pragma unroll
        for (int p = 0; p < BT; ++p)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                pg[p][g] = (b0 + p < B) ? pre[((int64_t)(b0 + p) * L + t) * (4 * H) + g * H + j] : 0.0f;
        float acc[BT][4];
#pragma unroll
        for (int p = 0; p < BT; ++p)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[p][g] = 0.0f;
        gemv_chain<H, BT>(Wp, hs[cur], j, acc);
#pragma unroll
        for (int p = 0; p < BT; ++p) {
            // gates = (h.W_hh^T + b_hh) + (x.W_ih^T + b_ih)
            lstm_cell_update(__fadd_rn(__fadd_rn(acc[p][0], bh[0]), pg[p][0]),
                             __fadd_rn(__fadd_rn(acc[p][1], bh[1]), pg[p][1]),
                             __fadd_rn(__fadd_rn(acc[p][2], bh[2]), pg[p][2]),
                             __fadd_rn(__fadd_rn(acc[p][3], bh[3]), pg[p][3]), c[p], h[p]);
            hs[cur ^ 1][p][j] = h[p];
            if (b0 + p < B) enc[((int64_t)(b0 + p) * L + t) * H + j] = h[p];
        }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int p = 0; p < BT; ++p) {
        if (b0 + p < B) {
            nets.h_n[net][(int64_t)(b0 + p) * H + j] = h[p];
            nets.c_n[net][(int64_t)(b0 + p) * H + j] = c[p];
        }
    }
}

template <int H>
static int launch_encode(const LstmNets& nets, int n_nets, int32_t B, int32_t L, hipStream_t s) {
    // problems per workgroup: enough workgroups to cover the 256 CUs, then amortise the W stream
    int bt = 1;
    while (bt < 8 && (int64_t)B * n_nets / (bt * 2) >= 256) bt *= 2;
    dim3 grid((B + bt - 1) / bt, n_nets), block(H);
    switch (bt) {
        case 1: hipLaunchKernelGGL((lstm_encode_kernel<H, 1>), grid, block, 0, s, nets, B, L); break;
        case 2: hipLaunchKernelGGL((lstm_encode_kernel<H, 2>), grid, block, 0, s, nets, B, L); break;
        case 4: hipLaunchKernelGGL((lstm_encode_kernel<H, 4>), grid, block, 0, s, nets, B, L); break;
        default: hipLaunchKernelGGL((lstm_encode_kernel<H, 8>), grid, block, 0, s, nets, B, L); break;
    }
    return 0;
}

extern "C" int gnnpn_lstm_encode_f32(int n_nets, const gnnpn_encode_net_t* in, int32_t B, int32_t L, int32_t H,
                                     int32_t F, int32_t precision, const gnnpn_launch_opts_t* opts_in, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
    GNNPN_REQUIRE(n_nets >= 1 && n_nets <= GNNPN_MAX_NETS, "lstm_encode: n_nets must be 1..%d", GNNPN_MAX_NETS);
    GNNPN_REQUIRE(in, "lstm_encode: null net array");
    GNNPN_REQUIRE(B >= 0 && L > 0, "lstm_encode: bad shape");
    g_gnnpn_last_units = 0;
    if (B == 0) return GNNPN_OK;                    // empty batch: its buffers may be NULL
    GNNPN_REQUIRE(precision >= GNNPN_PREC_F32 && precision <= GNNPN_PREC_SPLIT, "lstm_encode: unknown precision %d", precision);
    if (H != 256 && H != 32) GNNPN_FAIL(GNNPN_E_UNSUP, "lstm_encode: hidden size %d not built (256, 32)", H);
    LstmNets nets{};
    bool any_fold = false;
    for (int n = 0; n < n_nets; ++n) {
        const gnnpn_encode_net_t& e = in[n];
        GNNPN_REQUIRE(e.whh_packed && e.bhh && e.enc_out && e.h_n && e.c_n, "lstm_encode: null operand for net %d", n);
        GNNPN_REQUIRE(e.pregates || (e.inputs && e.w_in && e.b_in),
                      "lstm_encode: net %d needs pregates or (inputs, w_in, b_in)", n);
        GNNPN_REQUIRE(gnnpn_aligned(e.whh_packed, 16), "lstm_encode: packed weights must be 16-byte aligned");
        any_fold |= (e.pregates == nullptr);
        nets.pregates[n] = e.pregates;
        nets.inputs[n] = e.inputs;
        nets.w_in[n] = e.w_in;
        nets.b_in[n] = e.b_in;
        nets.whh[n] = e.whh_packed;
        GNNPN_REQUIRE(!e.whh_split || gnnpn_aligned(e.whh_split, 16), "lstm_encode: whh_split must be 16-byte aligned");
        nets.whh_split[n] = e.whh_split;
        nets.bhh[n] = e.bhh;
        nets.enc_out[n] = e.enc_out;
        nets.h_n[n] = e.h_n;
        nets.c_n[n] = e.c_n;
    }
    if (any_fold) GNNPN_REQUIRE(F == 8, "lstm_encode: in-kernel input projection is built for F = 8, got %d", F);
    if (B == 0) return GNNPN_OK;
    hipStream_t s = (hipStream_t)stream;
    const CoopOpts opts = coop_opts(opts_in);
    GNNPN_REQUIRE(opts.impl >= 0 && opts.impl <= 2, "lstm_encode: opts.impl must be 0 (auto), 1 (streaming) or 2 (cooperative, 8-member groups) — the 16-member form (3) was removed in ABI version 7");
    GNNPN_REQUIRE(opts.lds_kb >= 0 && opts.lds_kb <= 160, "lstm_encode: opts.lds_kb must be 0..160");
    const int impl = opts.impl;
    const bool coop = H == 256 && impl != 1 && (workspace != nullptr || impl >= 2);
    if (!coop && any_fold)
        GNNPN_FAIL(GNNPN_E_UNSUP, "lstm_encode: the streaming form needs precomputed pregates");
    if (precision != GNNPN_PREC_F32 && !coop)
        GNNPN_FAIL(GNNPN_E_UNSUP, "lstm_encode: fp16-operand precisions need the cooperative form (H = 256, workspace)");
    if (coop) {
        const int rc = gnnpn_launch_encode_coop(nets, n_nets, B, L, precision, opts, workspace, workspace_bytes, s);
        if (rc != GNNPN_OK) return rc;
    } else if (H == 256) {
        launch_encode<256>(nets, n_nets, B, L, s);
    } else {
        launch_encode<32>(nets, n_nets, B, L, s);
    }
    GNNPN_CHECK_LAUNCH("lstm_encode_f32");
    return GNNPN_OK;
}
