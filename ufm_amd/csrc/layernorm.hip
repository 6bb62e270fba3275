// LayerNorm over channels: one 64-lane wave per row, float4 loads, two-pass statistics in
// registers (mean, then centred variance -- the same association torch uses), wave reduction by
// DPP/shuffle; HBM-bound (reads D*4 B, writes D*2 or D*4 B per row).  An optional row-index table
// turns the kernel into gather+LN so "drop cls / reorder views" costs no extra pass.
#include "common.h"

namespace {

constexpr int MAX_VPL = 8;  // float4 per lane -> D <= 2048

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// D == VPL*256 known at compile time: the row is held in VPL float4 registers across the three passes.  (With a
// run-time D the guarded loops below compile to 14 VGPRs: hipcc re-reads the row from memory in every pass --
// 54 us vs 24 us for 21920 x 1024, found with tools/lab/ln_lab.hip.)
template <int OUT_BF16, int VPL>
__global__ __launch_bounds__(256) void layernorm_kernel_fixed(const float* __restrict__ x, int ldx,
                                                              const int32_t* __restrict__ row_index, int rows_out,
                                                              const float* __restrict__ w, const float* __restrict__ b,
                                                              float eps, void* out, int ldo) {
    constexpr int D = VPL * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows_out) return;
    const int in_row = row_index ? row_index[row] : row;
    const float* xr = x + (size_t)in_row * ldx;
    f32x4 v[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[i] = *(const f32x4*)(xr + (lane + i * 64) * 4);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[i][j] - mean;
            q += d * d;
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + i * 64;
        const f32x4 wv = *(const f32x4*)(w + c * 4);
        const f32x4 bv = *(const f32x4*)(b + c * 4);
        f32x4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = (v[i][j] - mean) * rstd * wv[j] + bv[j];
        if (OUT_BF16 == 2) {
            float h[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h[j] = bf16_to_f32(f32_to_bf16(y[j]));
                l[j] = y[j] - h[j];
            }
            u32x2 ph = {pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3])};
            u32x2 pl = {pack_bf16x2(l[0], l[1]), pack_bf16x2(l[2], l[3])};
            uint16_t* o = (uint16_t*)out + (size_t)row * ldo + c * 4;
            *(u32x2*)o = ph;
            *(u32x2*)(o + (size_t)rows_out * ldo) = pl;
        } else if (OUT_BF16 == 1) {
            u32x2 pk = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
            *(u32x2*)((uint16_t*)out + (size_t)row * ldo + c * 4) = pk;
        } else {
            *(f32x4*)((float*)out + (size_t)row * ldo + c * 4) = y;
        }
    }
}

template <int OUT_BF16>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx,
                                                        const int32_t* __restrict__ row_index, int rows_out,
                                                        int D, const float* __restrict__ w,
                                                        const float* __restrict__ b, float eps, void* out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows_out) return;
    const int in_row = row_index ? row_index[row] : row;
    const float* xr = x + (size_t)in_row * ldx;
    const int nvec = D >> 2;
    f32x4 v[MAX_VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_VPL; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            v[i] = *(const f32x4*)(xr + c * 4);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_VPL; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = v[i][j] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < MAX_VPL; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            const f32x4 wv = *(const f32x4*)(w + c * 4);
            const f32x4 bv = *(const f32x4*)(b + c * 4);
            f32x4 y;
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = (v[i][j] - mean) * rstd * wv[j] + bv[j];
            if (OUT_BF16 == 2) {  // split (hi, lo) bf16 planes: hi = bf16(y), lo = bf16(y - hi)
                float h[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h[j] = bf16_to_f32(f32_to_bf16(y[j]));
                    l[j] = y[j] - h[j];
                }
                u32x2 ph = {pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3])};
                u32x2 pl = {pack_bf16x2(l[0], l[1]), pack_bf16x2(l[2], l[3])};
                uint16_t* o = (uint16_t*)out + (size_t)row * ldo + c * 4;
                *(u32x2*)o = ph;
                *(u32x2*)(o + (size_t)rows_out * ldo) = pl;
            } else if (OUT_BF16 == 1) {
                u32x2 pk = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
                *(u32x2*)((uint16_t*)out + (size_t)row * ldo + c * 4) = pk;
            } else {
                *(f32x4*)((float*)out + (size_t)row * ldo + c * 4) = y;
            }
        }
    }
}

__global__ __launch_bounds__(256) void fill_rows_kernel(float* out, int ldo, int n_groups, int group_stride_rows,
                                                        const float* __restrict__ src, int D) {
    const int nvec = D >> 2;
    const int total = n_groups * nvec;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int g = i / nvec, c = i - g * nvec;
        *(f32x4*)(out + (size_t)g * group_stride_rows * ldo + c * 4) = *(const f32x4*)(src + c * 4);
    }
}

}  // namespace

extern "C" int ufm_layernorm(const float* x, int ldx, const int32_t* row_index, int rows_out, int D,
                             const float* weight, const float* bias, float eps, void* out, int out_dtype,
                             int ldo, void* stream) {
    UFM_REQUIRE(x && weight && bias && out, "ufm_layernorm: null pointer");
    UFM_REQUIRE(rows_out > 0, "ufm_layernorm: rows_out=%d", rows_out);
    UFM_REQUIRE(D % 4 == 0 && D <= MAX_VPL * 256 && D > 0, "ufm_layernorm: D=%d must be a multiple of 4 and <= %d", D, MAX_VPL * 256);
    UFM_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && ldx >= D && ldo >= D, "ufm_layernorm: bad ldx/ldo %d/%d", ldx, ldo);
    UFM_REQUIRE(out_dtype == UFM_F32 || out_dtype == UFM_BF16 || out_dtype == UFM_BF16X2, "ufm_layernorm: bad out_dtype");
    dim3 grid((rows_out + 3) / 4), block(256);
    hipStream_t st = (hipStream_t)stream;
#define UFM_LN_FIXED(OUT, VPL) hipLaunchKernelGGL((layernorm_kernel_fixed<OUT, VPL>), grid, block, 0, st, x, ldx, row_index, rows_out, weight, bias, eps, out, ldo)
#define UFM_LN_BY_VPL(OUT)                  \
    switch (D / 256) {                      \
        case 1: UFM_LN_FIXED(OUT, 1); break; \
        case 2: UFM_LN_FIXED(OUT, 2); break; \
        case 3: UFM_LN_FIXED(OUT, 3); break; \
        case 4: UFM_LN_FIXED(OUT, 4); break; \
        case 5: UFM_LN_FIXED(OUT, 5); break; \
        case 6: UFM_LN_FIXED(OUT, 6); break; \
        case 7: UFM_LN_FIXED(OUT, 7); break; \
        default: UFM_LN_FIXED(OUT, 8); break; \
    }
    if (D % 256 == 0) {
        if (out_dtype == UFM_BF16X2) { UFM_LN_BY_VPL(2) } else if (out_dtype == UFM_BF16) { UFM_LN_BY_VPL(1) } else { UFM_LN_BY_VPL(0) }
    } else if (out_dtype == UFM_BF16X2)
        hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, st, x, ldx, row_index, rows_out, D, weight, bias, eps, out, ldo);
    else if (out_dtype == UFM_BF16)
        hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, st, x, ldx, row_index, rows_out, D, weight, bias, eps, out, ldo);
    else
        hipLaunchKernelGGL(layernorm_kernel<0>, grid, block, 0, st, x, ldx, row_index, rows_out, D, weight, bias, eps, out, ldo);
#undef UFM_LN_BY_VPL
#undef UFM_LN_FIXED
    UFM_CHECK_LAUNCH("ufm_layernorm");
    return UFM_OK;
}

extern "C" int ufm_fill_rows(float* out, int ldo, int n_groups, int group_stride_rows, const float* src, int D,
                             void* stream) {
    UFM_REQUIRE(out && src, "ufm_fill_rows: null pointer");
    UFM_REQUIRE(D % 4 == 0 && ldo % 4 == 0 && n_groups > 0, "ufm_fill_rows: bad shape");
    const int total = n_groups * (D / 4);
    dim3 grid(min((total + 255) / 256, 2048)), block(256);
    hipLaunchKernelGGL(fill_rows_kernel, grid, block, 0, (hipStream_t)stream, out, ldo, n_groups, group_stride_rows, src, D);
    UFM_CHECK_LAUNCH("ufm_fill_rows");
    return UFM_OK;
}
