// libmmgl_hip.so: error channel + version.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void mmgl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mmgl_last_error(void) { return g_err; }
// ABI version: bumped whenever an exported signature changes (mmgl_amd/_lib.py refuses a library that reports another one -- a
// stale build would otherwise load and run with misaligned arguments).  101: mmgl_xattn_fwd lost p_drop / seed / offset.
// 102: round 4 (tile counters bound to one stream; entry points added / removed with the kernel families).
// 103: mmgl_comm_* / mmgl_allreduce_sum / mmgl_allgather / mmgl_broadcast.
extern "C" int mmgl_version(void) { return 104; }
