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

// Compute units of the CURRENT device (cached per device id): sizes persistent grids and the whole-rounds dispatch.
int ufm_device_cu_count();
bool ufm_stream_is_concurrent(void* stream);  // error.cpp: flagged by ufm_hint_concurrent_stream

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

// GELU for bf16 outputs, four values at a time, ONE transcendental per value:
//   gelu(x) = relu(x) - |x| * Phi(-|x|),   Phi(-a) = 2^-g(a),   g = degree-6 minimax fit of -log2 Phi(-a) on [0, 6]
// (|x| is clamped to 6 inside g only: Phi(-6) = 1e-9).  Max |error| 6.9e-6, max relative error 4.8e-5 of the exact
// erf form = 0.012 bf16 ulp, relative accuracy kept in the negative tail (tools/fit_gelu.py regenerates and checks
// the coefficients).  Written on float2 so that hipcc emits v_pk_fma_f32: ~10 issue slots per value against ~24 for
// gelu_erf_fast -- in the 256x256 GEMM epilogue all 8 waves of a CU do this at once with the matrix cores idle.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x4 gelu_bf16_x4(f32x4 v) {
    f32x2 x[2] = {{v[0], v[1]}, {v[2], v[3]}};
    f32x4 out;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // |x| and min(|x|, 6) on the bit patterns (non-negative floats order like unsigned integers): no IEEE
        // canonicalisation v_max in front of every fmin/fmax of an MFMA result
        const unsigned b0 = __float_as_uint(x[h][0]) & 0x7fffffffu, b1 = __float_as_uint(x[h][1]) & 0x7fffffffu;
        const f32x2 ax = {__uint_as_float(b0), __uint_as_float(b1)};
        const f32x2 a = {__uint_as_float(min(b0, 0x40C00000u)), __uint_as_float(min(b1, 0x40C00000u))};
        f32x2 g = {-2.29992501e-05f, -2.29992501e-05f};
        g = __builtin_elementwise_fma(g, a, (f32x2){0.000611490188f, 0.000611490188f});
        g = __builtin_elementwise_fma(g, a, (f32x2){-0.00720018874f, -0.00720018874f});
        g = __builtin_elementwise_fma(g, a, (f32x2){0.0512082147f, 0.0512082147f});
        g = __builtin_elementwise_fma(g, a, (f32x2){0.461222249f, 0.461222249f});
        g = __builtin_elementwise_fma(g, a, (f32x2){1.15021447f, 1.15021447f});
        g = __builtin_elementwise_fma(g, a, (f32x2){1.00005891f, 1.00005891f});
        const f32x2 e = {__builtin_amdgcn_exp2f(-g[0]), __builtin_amdgcn_exp2f(-g[1])};
        // relu(x) as a signed-integer max on the bit pattern (negative floats are negative integers)
        const f32x2 r = {__int_as_float(max(__float_as_int(x[h][0]), 0)), __int_as_float(max(__float_as_int(x[h][1]), 0))};
        const f32x2 y = __builtin_elementwise_fma(-ax, e, r);
        out[2 * h] = y[0];
        out[2 * h + 1] = y[1];
    }
    return out;
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

// In-kernel stamps of the 8-phase GEMM / convolution kernels (cdna_hip_programming.md section 7; MI355X_MICROARCH.md "DVFS give-back" item 6).  Diagnostic instantiations
// only: the shipped kernels execute no stamp.  Row of workgroup b (8 x uint64), written by its first lane into a buffer of its own:
//   0: blockIdx | HW_ID << 32      1: XCC_ID | LDS_ALLOC << 32      2: s_memtime at entry      3: s_memtime after the K loop
//   4: s_memtime after the epilogue's last store has completed      5: s_memrealtime (100 MHz) at entry      6: s_memrealtime at the end
//   7: s_memtime after the prologue (first fragments readable)
static __device__ __forceinline__ unsigned long long gemm_stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
struct GemmStamps {
    unsigned long long t_entry, t_prologue, t_loop, rt_entry;
    __device__ __forceinline__ void entry() {
        t_entry = gemm_stamp();
        rt_entry = __builtin_amdgcn_s_memrealtime();
    }
    __device__ __forceinline__ void finish(unsigned long long* buf, int rows) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = gemm_stamp(), rt_end = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0 && (int)blockIdx.x < rows) {
            unsigned hw, xcc, lds;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(lds));
            unsigned long long* d = buf + (size_t)blockIdx.x * 8;
            d[0] = (unsigned long long)blockIdx.x | ((unsigned long long)hw << 32);
            d[1] = (unsigned long long)xcc | ((unsigned long long)lds << 32);
            d[2] = t_entry, d[3] = t_loop, d[4] = t_end, d[5] = rt_entry, d[6] = rt_end, d[7] = t_prologue;
        }
    }
};

