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

// Streams the caller has declared to run CONCURRENTLY with others of its own (ufm_hint_concurrent_stream): the tile-height choice of the
// GEMM / bf16x3 dispatch then minimises the CU time a launch takes from its neighbours instead of its own latency.  A small fixed table,
// written rarely (an engine flags its micro-batch streams when it creates them), read on every GEMM / convolution launch.
static std::atomic<void*> g_conc_streams[32];
extern "C" int ufm_hint_concurrent_stream(void* stream, int on) {
    if (stream == nullptr) {
        ufm_set_error("ufm_hint_concurrent_stream: the null (legacy default) stream cannot be flagged");
        return UFM_ERR_ARG;
    }
    for (auto& e : g_conc_streams) {  // already there?
        void* cur = e.load(std::memory_order_acquire);
        if (cur == stream) {
            if (!on) e.store(nullptr, std::memory_order_release);
            return UFM_OK;
        }
    }
    if (!on) return UFM_OK;
    for (auto& e : g_conc_streams) {
        void* expect = nullptr;
        if (e.compare_exchange_strong(expect, stream, std::memory_order_acq_rel)) return UFM_OK;
    }
    ufm_set_error("ufm_hint_concurrent_stream: more than %d streams flagged", (int)(sizeof(g_conc_streams) / sizeof(g_conc_streams[0])));
    return UFM_ERR_ARG;
}
bool ufm_stream_is_concurrent(void* stream) {
    if (stream == nullptr) return false;
    for (auto& e : g_conc_streams)
        if (e.load(std::memory_order_relaxed) == stream) return true;
    return false;
}
