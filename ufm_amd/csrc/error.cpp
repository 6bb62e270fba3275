// Thread-local error string + ABI version for libufm_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>

#include "../../include/ufm_hip.h"
#include "lab_flags.h"

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
// written rarely (an engine flags its micro-batch streams when it creates them and un-flags them when it dies), read lock-free on every
// GEMM / convolution launch.  Entries are REFERENCE-COUNTED per handle (round 6): stream handles come out of a small pool and two engines
// may hold the same one -- the flag goes when the last holder gives it back, and a handle given back more often than taken is an error.
// A stale flag (a holder that never un-flags; a destroyed stream whose address is reused) costs tile policy only, never bits.
static constexpr int CONC_SLOTS = 32;
static std::atomic<void*> g_conc_streams[CONC_SLOTS];
static int g_conc_refs[CONC_SLOTS];  // guarded by g_conc_mutex
static std::mutex g_conc_mutex;
extern "C" int ufm_hint_concurrent_stream(void* stream, int on) {
    if (stream == nullptr) {
        ufm_set_error("ufm_hint_concurrent_stream: the null (legacy default) stream cannot be flagged");
        return UFM_ERR_ARG;
    }
    std::lock_guard<std::mutex> lock(g_conc_mutex);
    for (int i = 0; i < CONC_SLOTS; ++i) {  // already there?
        if (g_conc_streams[i].load(std::memory_order_acquire) == stream) {
            if (on) {
                ++g_conc_refs[i];
            } else if (--g_conc_refs[i] <= 0) {
                g_conc_refs[i] = 0;
                g_conc_streams[i].store(nullptr, std::memory_order_release);
            }
            return UFM_OK;
        }
    }
    if (!on) return UFM_OK;  // un-flagging an unknown handle: a no-op (the hint may have been refused when it was asked for)
    for (int i = 0; i < CONC_SLOTS; ++i) {
        if (g_conc_streams[i].load(std::memory_order_relaxed) == nullptr) {
            g_conc_refs[i] = 1;
            g_conc_streams[i].store(stream, std::memory_order_release);
            return UFM_OK;
        }
    }
    ufm_set_error("ufm_hint_concurrent_stream: more than %d streams flagged", CONC_SLOTS);
    return UFM_ERR_ARG;
}
bool ufm_stream_is_concurrent(void* stream) {
    if (stream == nullptr) return false;
    for (auto& e : g_conc_streams)
        if (e.load(std::memory_order_relaxed) == stream) return true;
    return false;
}

// The lab flag tables (lab_flags.h), walkable from outside: word 0 = ufm_debug_set_gemm_flags, 1 = ufm_debug_set_conv_variant.
extern "C" int ufm_debug_lab_field(int word, int index, const char** name, int* shift, int* width) {
    const LabField* t = word == 0 ? gemm_lab::ALL : word == 1 ? conv_lab::ALL : nullptr;
    const int n = word == 0 ? (int)(sizeof(gemm_lab::ALL) / sizeof(LabField)) : word == 1 ? (int)(sizeof(conv_lab::ALL) / sizeof(LabField)) : 0;
    if (t == nullptr || index < 0 || index >= n || !name || !shift || !width) {
        ufm_set_error("ufm_debug_lab_field: word %d has no field %d", word, index);
        return UFM_ERR_ARG;
    }
    *name = t[index].name, *shift = t[index].shift, *width = t[index].width;
    return UFM_OK;
}
