// LayerNorm ablation lab (not part of the product): variants of the row kernel to find what costs 2.5x vs streaming.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ unsigned pk(float a, float b) { typedef __attribute__((ext_vector_type(2))) __bf16 bf2; bf2 v = {(__bf16)a, (__bf16)b}; return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// V: 0 = full LN (wave per row, 4 float4/lane, D=1024), 1 = no reductions (fixed stats), 2 = no store, 3 = loads only (sum to dummy)
// 4 = block per 4 rows but each THREAD-quad... (row per 64 lanes same) with 16-B bf16 stores (lane owns 8 consecutive)
template <int V>
__global__ __launch_bounds__(256) void ln(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, uint16_t* out, float* dummy, int rows) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * 1024;
    f32x4 v[4];
    if (V == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *(const f32x4*)(xr + ((i >> 1) * 128 + lane * 2 + (i & 1)) * 4);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = *(const f32x4*)(xr + (lane + i * 64) * 4);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    if (V == 3) { if (s == 123.456f) dummy[0] = s; return; }
    float mean, rstd;
    if (V == 1) { mean = 0.1f; rstd = 0.9f + s * 1e-30f; }
    else {
        mean = wsum(s) / 1024.f;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { float d = v[i][j] - mean; q += d * d; }
        rstd = rsqrtf(wsum(q) / 1024.f + 1e-6f);
    }
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i += (V == 4 ? 2 : 1)) {
        if (V == 4) {
            const int c = (i >> 1) * 128 + lane * 2;
            f32x4 w0 = *(const f32x4*)(w + c * 4), w1 = *(const f32x4*)(w + c * 4 + 4), b0 = *(const f32x4*)(b + c * 4), b1 = *(const f32x4*)(b + c * 4 + 4);
            f32x4 y0, y1;
#pragma unroll
            for (int j = 0; j < 4; ++j) { y0[j] = (v[i][j] - mean) * rstd * w0[j] + b0[j]; y1[j] = (v[i + 1][j] - mean) * rstd * w1[j] + b1[j]; }
            u32x4 p = {pk(y0[0], y0[1]), pk(y0[2], y0[3]), pk(y1[0], y1[1]), pk(y1[2], y1[3])};
            *(u32x4*)(out + (size_t)row * 1024 + c * 4) = p;
        } else {
            const int c = lane + i * 64;
            f32x4 wv = *(const f32x4*)(w + c * 4), bv = *(const f32x4*)(b + c * 4), y;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = (v[i][j] - mean) * rstd * wv[j] + bv[j];
            if (V == 2) keep += y[0] + y[1] + y[2] + y[3];
            else { u32x2 p = {pk(y[0], y[1]), pk(y[2], y[3])}; *(u32x2*)(out + (size_t)row * 1024 + c * 4) = p; }
        }
    }
    if (V == 2 && keep == 123.456f) dummy[0] = keep;
}
// V5: 8 rows per block of 512 threads? V6: thread-per-float4 two-kernel style is not LN. keep simple.
extern "C" void run(int v, const float* x, const float* w, const float* b, uint16_t* out, float* dummy, int rows, void* stream) {
    dim3 g((rows + 3) / 4), bl(256);
    hipStream_t s = (hipStream_t)stream;
    switch (v) {
        case 0: hipLaunchKernelGGL(ln<0>, g, bl, 0, s, x, w, b, out, dummy, rows); break;
        case 1: hipLaunchKernelGGL(ln<1>, g, bl, 0, s, x, w, b, out, dummy, rows); break;
        case 2: hipLaunchKernelGGL(ln<2>, g, bl, 0, s, x, w, b, out, dummy, rows); break;
        case 3: hipLaunchKernelGGL(ln<3>, g, bl, 0, s, x, w, b, out, dummy, rows); break;
        case 4: hipLaunchKernelGGL(ln<4>, g, bl, 0, s, x, w, b, out, dummy, rows); break;
    }
}
