#include "common.h"

thread_local char g_gnnpn_err[256] = "";

extern "C" int gnnpn_abi_version(void) { return GNNPN_ABI_VERSION; }
extern "C" const char* gnnpn_last_error(void) { return g_gnnpn_err; }
