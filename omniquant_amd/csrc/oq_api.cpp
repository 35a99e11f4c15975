// Error reporting + version for the C ABI (include/oq_hip.h).  No exceptions cross the boundary.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/oq_hip.h"

static thread_local char g_err[512] = "";

void oq_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int oq_version(void) { return 100; }   // 0.1.0
extern "C" const char* oq_last_error(void) { return g_err; }
