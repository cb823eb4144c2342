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

// which kernel the last genie_conv_igemm / genie_conv_wgrad call of this thread launched (profiling aid)
static thread_local int g_last_variant = -1;
void genie_note_variant(int v) { g_last_variant = v; }
extern "C" int genie_last_conv_variant(void) { return g_last_variant; }

// GroupNorm work the last genie_conv_igemm call of this thread fused into its epilogue (bit 0: output sums, bit 1: backward partials)
static thread_local int g_last_gn_fused = 0;
void genie_note_gn_fused(int mask) { g_last_gn_fused = mask; }
extern "C" int genie_last_conv_gn_fused(void) { return g_last_gn_fused; }
