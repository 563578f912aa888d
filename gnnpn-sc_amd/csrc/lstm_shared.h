// Argument block shared by the two encoder implementations (lstm.hip, lstm_coop.hip).
#pragma once
#include "common.h"

#define GNNPN_MAX_NETS 4

struct LstmNets {
    const float* pregates[GNNPN_MAX_NETS];   // or nullptr: input projection in-kernel from the three below
    const float* inputs[GNNPN_MAX_NETS];
    const float* w_in[GNNPN_MAX_NETS];
    const float* b_in[GNNPN_MAX_NETS];
    const float* whh[GNNPN_MAX_NETS];
    const void* whh_split[GNNPN_MAX_NETS];   // or nullptr: the exact-split build splits whh itself (coop_common.h: load_split_weights)
    const float* bhh[GNNPN_MAX_NETS];
    float* enc_out[GNNPN_MAX_NETS];
    float* h_n[GNNPN_MAX_NETS];
    float* c_n[GNNPN_MAX_NETS];
};


// per-call launch options, resolved from gnnpn_launch_opts_t (NULL = defaults)
struct CoopOpts {
    int impl = 0, lds_kb = 0;
    bool write_through = false, paired_start = false;
    unsigned* sticky = nullptr;
};
inline CoopOpts coop_opts(const gnnpn_launch_opts_t* o) {
    CoopOpts c;
    if (o) {
        c.impl = o->impl;
        c.lds_kb = o->lds_kb;
        c.write_through = o->write_through != 0;
        c.paired_start = o->paired_start != 0;
        c.sticky = o->sticky_status;
    }
    return c;
}

int gnnpn_launch_encode_coop(const LstmNets& nets, int n_nets, int32_t B, int32_t L, int precision, const CoopOpts& opts,
                             void* workspace, int64_t workspace_bytes, hipStream_t s);
constexpr int COOP_SEAT_TABLE_WORDS = 8 * 256 + 8 + 8;   // CU seats, per-XCD hand-out counters, [8*256+8] = launches staffing right now
unsigned* gnnpn_cu_seat_table();   // api.hip: the device's canonical CU -> seat table (nullptr: allocation failed)
extern thread_local int64_t g_gnnpn_last_units;   // api.hip: gnnpn_last_launch_units
int gnnpn_option_lstm_ablate();   // timing experiments only: results are wrong when non-zero
