// csrc/abi.hip -- error reporting and version of the C ABI (include/eap_hip.h).
#include "common.h"

#include <string.h>

namespace eap {
static thread_local char g_err[256] = "";
void set_error(const char *msg) {
    strncpy(g_err, msg, sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
static thread_local char g_kernel[192] = "";
void set_kernel(const char *name) {
    strncpy(g_kernel, name, sizeof(g_kernel) - 1);
    g_kernel[sizeof(g_kernel) - 1] = 0;
}
}  // namespace eap

extern "C" const char *eap_last_error(void) { return eap::g_err; }
extern "C" const char *eap_last_kernel(void) {      // read-and-clear: a name is reported once, for the entry that set it
    static thread_local char out[192];
    memcpy(out, eap::g_kernel, sizeof(out));
    eap::g_kernel[0] = 0;
    return out;
}
extern "C" int eap_abi_version(void) { return 1; }
