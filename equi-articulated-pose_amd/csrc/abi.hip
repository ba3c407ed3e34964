// csrc/abi.hip -- error reporting and version of the C ABI (include/eap_hip.h).
#include "common.h"

#include <string.h>
#include <mutex>

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

// A side stream per device for work that may overlap the caller's stream inside ONE entry (the zpconv index check, an HBM
// stream, beside the matrix kernel it guards): side_fork makes the side stream wait for everything queued on `s` so far,
// side_join makes `s` wait for everything queued on the side stream.  Created on first use, never destroyed.
namespace {
struct Side { hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
Side g_side[64];
std::mutex g_side_mutex;       // creation only: the events are recorded / waited on by the calling thread's own launches
}  // namespace
static int side_of(Side **out) {
    int dev = 0;
    int e = hip_fail(hipGetDevice(&dev), "side stream (device)");
    if (e) return e;
    if (dev < 0 || dev >= 64) return bad_arg("side stream: device index out of range");
    Side &sd = g_side[dev];
    std::lock_guard<std::mutex> lock(g_side_mutex);
    if (!sd.stream) {
        if ((e = hip_fail(hipStreamCreateWithFlags(&sd.stream, hipStreamNonBlocking), "side stream (create)"))) return e;
        if ((e = hip_fail(hipEventCreateWithFlags(&sd.fork, hipEventDisableTiming), "side stream (event)"))) return e;
        if ((e = hip_fail(hipEventCreateWithFlags(&sd.join, hipEventDisableTiming), "side stream (event)"))) return e;
    }
    *out = &sd;
    return 0;
}
int side_fork(hipStream_t s, hipStream_t *side) {
    Side *sd;
    int e = side_of(&sd);
    if (e) return e;
    if ((e = hip_fail(hipEventRecord(sd->fork, s), "side stream (fork)"))) return e;
    if ((e = hip_fail(hipStreamWaitEvent(sd->stream, sd->fork, 0), "side stream (fork)"))) return e;
    *side = sd->stream;
    return 0;
}
int side_join(hipStream_t s) {
    Side *sd;
    int e = side_of(&sd);
    if (e) return e;
    if ((e = hip_fail(hipEventRecord(sd->join, sd->stream), "side stream (join)"))) return e;
    return hip_fail(hipStreamWaitEvent(s, sd->join, 0), "side stream (join)");
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
