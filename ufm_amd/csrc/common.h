// Shared device/host helpers for libufm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ufm_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;    // MFMA bf16 A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;    // tr-read result (2 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;     // 16x16 accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;   // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ---- host-side error plumbing (error.cpp) ----
void ufm_set_error(const char* fmt, ...);
#define UFM_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            ufm_set_error(__VA_ARGS__); \
            return UFM_ERR_ARG;         \
        }                               \
    } while (0)
#define UFM_CHECK_LAUNCH(name)                                                \
    do {                                                                      \
        hipError_t e_ = hipGetLastError();                                    \
        if (e_ != hipSuccess) {                                               \
            ufm_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return UFM_ERR_LAUNCH;                                            \
        }                                                                     \
    } while (0)

// ---- bf16 <-> f32 (RNE; plain casts so NaN stays NaN, MI355X_MICROARCH "Correctness boundaries") ----
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32 at -O3
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Branch-free GELU for bf16 outputs: erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, three
// orders below bf16 resolution); ~14 VALU ops vs the branchy libm erff.
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = 1.0f - poly * __builtin_amdgcn_exp2f(-z * z * 1.44269504088896340736f);
    return 0.5f * x * (1.0f + copysignf(e, x));
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == UFM_ACT_GELU) return gelu_erf(v);
    if (act == UFM_ACT_RELU) return fmaxf(v, 0.0f);
    return v;
}

// XCD-aware bijective block remap (cdna_hip_programming.md T1): blocks b and b+8 share an XCD's L2;
// give each XCD a contiguous chunk of the logical tile space so neighbouring tiles (which share
// operand panels) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int nx = 8;
    int q = nblocks / nx, r = nblocks % nx;
    int xcd = bid % nx, local = bid / nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}
