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
extern "C" int mmgl_version(void) { return 100; }
