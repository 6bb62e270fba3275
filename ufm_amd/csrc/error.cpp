// Thread-local error string + ABI version for libufm_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include <hip/hip_runtime.h>

#include <atomic>

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

// Compute units of the current device, cached per device id (a process may drive several devices; the count sizes the
// persistent attention grid and the GEMM / conv whole-rounds split, so it must follow the device a launch goes to).
int ufm_device_cu_count() {
    constexpr int MAXDEV = 64;
    static std::atomic<int> cache[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return 256;
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
