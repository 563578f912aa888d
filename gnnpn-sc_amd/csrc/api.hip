#include "common.h"
#include <string.h>
#include "lstm_shared.h"
#include "decode_shared.h"

thread_local char g_gnnpn_err[256] = "";

extern "C" int gnnpn_abi_version(void) { return GNNPN_ABI_VERSION; }
extern "C" const char* gnnpn_last_error(void) { return g_gnnpn_err; }

// ---- run-time options (A/B switches for tests and benchmarks) ---------------------------------
static int g_lstm_impl = 0, g_decode_impl = 0;
int gnnpn_option_lstm_impl() { return g_lstm_impl; }
int gnnpn_option_decode_impl() { return g_decode_impl; }

extern "C" int gnnpn_set_option(const char* name, int value) {
    if (!name) GNNPN_FAIL(GNNPN_E_ARG, "set_option: null name");
    if (!strcmp(name, "lstm_impl")) {
        GNNPN_REQUIRE(value >= 0 && value <= 2, "set_option: lstm_impl must be 0 (auto), 1 (streaming) or 2 (cooperative)");
        g_lstm_impl = value;
        return GNNPN_OK;
    }
    if (!strcmp(name, "decode_impl")) {
        GNNPN_REQUIRE(value >= 0 && value <= 2, "set_option: decode_impl must be 0 (auto), 1 (streaming) or 2 (cooperative)");
        g_decode_impl = value;
        return GNNPN_OK;
    }
    GNNPN_FAIL(GNNPN_E_ARG, "set_option: unknown option '%s'", name);
}
