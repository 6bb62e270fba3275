// Thread-local error string + ABI version for libufm_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/ufm_hip.h"

static thread_local char g_err[512] = "";

void ufm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int ufm_abi_version(void) { return UFM_ABI_VERSION; }
extern "C" const char* ufm_last_error(void) { return g_err; }
extern "C" const char* ufm_built_arch(void) { return "gfx950"; }
