#include "common.h"
#include <string.h>
#include "lstm_shared.h"
#include "decode_shared.h"

thread_local char g_gnnpn_err[256] = "";

extern "C" int gnnpn_abi_version(void) { return GNNPN_ABI_VERSION; }
extern "C" const char* gnnpn_last_error(void) { return g_gnnpn_err; }

// ---- run-time options (A/B switches for tests and benchmarks) ---------------------------------
static int g_lstm_impl = 0, g_decode_impl = 0, g_lstm_ablate = 0, g_coop_lds_kb = 0;
int gnnpn_option_coop_lds_kb() { return g_coop_lds_kb; }
int gnnpn_option_lstm_ablate() { return g_lstm_ablate; }
int gnnpn_option_lstm_impl() { return g_lstm_impl; }
int gnnpn_option_decode_impl() { return g_decode_impl; }

extern "C" int gnnpn_set_option(const char* name, int value) {
    if (!name) GNNPN_FAIL(GNNPN_E_ARG, "set_option: null name");
    if (!strcmp(name, "lstm_impl")) {
        GNNPN_REQUIRE(value >= 0 && value <= 2, "set_option: lstm_impl must be 0 (auto), 1 (streaming) or 2 (cooperative)");
        g_lstm_impl = value;
        return GNNPN_OK;
    }
    if (!strcmp(name, "lstm_ablate")) {   // bit0 no MFMA, bit1 no transcendentals, bit2 no tag wait, bit3 no sweep, bit4 no publish
        g_lstm_ablate = value;
        return GNNPN_OK;
    }
    if (!strcmp(name, "coop_lds_kb")) {   // LDS footprint (KB per workgroup, padded with unused dynamic LDS) of the cooperative
        // kernels launched from now on; 0 = just what they use.  Placement control for co-resident launches:
        // with 100 KB on one stream and 56 KB on the other a CU takes one workgroup of each, never two of the first.
        GNNPN_REQUIRE(value >= 0 && value <= 160, "set_option: coop_lds_kb must be 0..160");
        g_coop_lds_kb = value;
        return GNNPN_OK;
    }
    if (!strcmp(name, "decode_impl")) {
        GNNPN_REQUIRE(value >= 0 && value <= 4, "set_option: decode_impl must be 0 (auto), 1 (streaming), 2 (cooperative, 8-CU groups), 3 (16-CU groups) or 4 (8-CU groups, 256-register build)");
        g_decode_impl = value;
        return GNNPN_OK;
    }
    GNNPN_FAIL(GNNPN_E_ARG, "set_option: unknown option '%s'", name);
}

// ---- test hook: the cell activations on an array (tests/test_gpu_ops.py::test_cell_activations)
__global__ void cell_activations_kernel(const float* __restrict__ x, float* __restrict__ sig, float* __restrict__ th,
                                        int64_t n) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) {
        sig[i] = cell_sigmoid(x[i]);
        th[i] = cell_tanh(x[i]);
    }
}

extern "C" int gnnpn_debug_cell_activations(const float* x, float* sig, float* th, int64_t n, void* stream) {
    GNNPN_REQUIRE(x && sig && th && n >= 0, "debug_cell_activations: bad argument");
    if (n == 0) return GNNPN_OK;
    hipLaunchKernelGGL(cell_activations_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, sig, th, n);
    GNNPN_CHECK_LAUNCH("debug_cell_activations");
    return GNNPN_OK;
}
