// csrc/abi.hip -- error reporting and version of the C ABI (include/eap_hip.h).
#include "common.h"

#include <string.h>

namespace eap {
static thread_local char g_err[256] = "";
void set_error(const char *msg) {
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
}  // namespace eap

extern "C" const char *eap_last_error(void) { return eap::g_err; }
extern "C" int eap_abi_version(void) { return 1; }
