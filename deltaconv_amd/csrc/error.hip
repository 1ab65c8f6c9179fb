// Error string + version of the C-ABI library.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void dc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

DC_EXPORT const char* dc_last_error(void) { return g_err; }
DC_EXPORT int32_t dc_version(void) { return 100; }  // 0.1.0 -> major*10000 + minor*100 + patch
