// Error plumbing and library identity for libocc_hip.so.
#include "occ_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void occ_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
const char* occ_last_error(void) { return g_err; }
int occ_version(void) { return 100; }
const char* occ_arch(void) { return "gfx950"; }
}
