// Argument blocks shared by the two decoder implementations (decode.hip, decode_coop.hip).
#pragma once
#include "common.h"
#include "lstm_shared.h"   // CoopOpts

// device-side view of gnnpn_decode_net_t (same field order; see include/gnnpn_hip.h)
struct DecodeNet {
    const float* embedded;
    const float* enc_out;
    const float* h0;
    const float* c0;
    const float* start;
    const float* wih;
    const float* whh;
    const float* bih;
    const float* bhh;
    const float* latent_win;
    const float* emb_w;
    const float* emb_b;
    const float* xw_fold;
    const float* xb_fold;
    const float* start_fold;
    int32_t* idx;
    float* win_logits;
    float* pick_prob;
    float* actions;
    float* queries;
    int32_t latent_from;
    int32_t sample;
    uint64_t sample_seed;
    const void* whh_split;       // or nullptr (gnnpn_lstm_pack_split_weights_f32)
};
static_assert(sizeof(DecodeNet) == sizeof(gnnpn_decode_net_t), "DecodeNet must mirror gnnpn_decode_net_t");

#define GNNPN_MAX_DECODE_NETS 2

struct DecodeArgs {
    DecodeNet net[GNNPN_MAX_DECODE_NETS];
    const float* inputs;
    float tanh_c;
    int use_tanh;
    int32_t B, T, K;
};

bool gnnpn_decode_coop_supported(int32_t H, int32_t n_per);
int gnnpn_launch_decode_coop(const DecodeArgs& args, int n_nets, int precision, bool shared_cu, const CoopOpts& opts,
                             void* workspace, int64_t workspace_bytes, hipStream_t s);
// decode_lean.hip: the production build of the cooperative form (folded input side, greedy picks; fp32 and exact split)
int gnnpn_launch_decode_lean(const DecodeArgs& args, int n_nets, int precision, bool shared_cu, const CoopOpts& opts,
                             void* workspace, int64_t workspace_bytes, hipStream_t s);
int64_t gnnpn_decode_lean_workspace_bytes(int32_t B, int32_t T, int32_t n_per);
unsigned* gnnpn_decode_diag_buffer();   // decode_coop.hip: device address of the failure record (gnnpn_decode_diag), or nullptr
