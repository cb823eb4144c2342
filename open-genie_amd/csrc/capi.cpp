// Error reporting and ABI version for libgenie_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "genie_hip.h"

static thread_local char g_err[512] = "";

void genie_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* genie_last_error(void) { return g_err; }
extern "C" int genie_abi_version(void) { return GENIE_ABI_VERSION; }
