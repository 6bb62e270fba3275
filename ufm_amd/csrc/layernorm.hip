// LayerNorm over channels: one 64-lane wave per row, float4 loads, two-pass statistics in
// registers (mean, then centred variance -- the same association torch uses), wave reduction by
// DPP/shuffle; HBM-bound (reads D*4 B, writes D*2 or D*4 B per row).  An optional row-index table
// turns the kernel into gather+LN so "drop cls / reorder views" costs no extra pass.
// ADD (ufm_add_layernorm): the residual update of the preceding branch is fused in front of the normalisation,
//   x[row] += gamma * branch[row]   (branch: the bf16 output of the proj / fc2 Linear, gamma: LayerScale, fp32 math),
// x is written back -- the fp32 read-modify-write of the residual stream leaves the GEMM epilogue (where all 256 CUs
// hit HBM at once with the matrix cores idle) for this HBM-bound kernel, and the GEMM gets the cheap bf16 store.
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
struct LnAdd {  // ADD: x[row] += gamma * branch[row] before the statistics (row_index must be null)
    const uint16_t* branch;  // bf16 [rows][ldb]
    const float* gamma;      // fp32 [D] or null (= 1)
    int ldb;
    long long out_plane;     // UFM_BF16X2 output: element offset of the lo plane (rows_out * ldo unless the output is a row slice of a larger buffer)
};

__device__ __forceinline__ f32x4 ln_add4(const f32x4& x, const uint16_t* br, const float* gamma, int c4) {
    const u32x2 a = *(const u32x2*)(br + c4);
    f32x4 g = {1.f, 1.f, 1.f, 1.f};
    if (gamma) g = *(const f32x4*)(gamma + c4);
    f32x4 r;
    r[0] = x[0] + g[0] * __uint_as_float(a[0] << 16);
    r[1] = x[1] + g[1] * __uint_as_float(a[0] & 0xffff0000u);
    r[2] = x[2] + g[2] * __uint_as_float(a[1] << 16);
    r[3] = x[3] + g[3] * __uint_as_float(a[1] & 0xffff0000u);
    return r;
}

template <int OUT_BF16, int VPL, bool ADD = false>
__global__ __launch_bounds__(256) void layernorm_kernel_fixed(const float* __restrict__ x, int ldx,
                                                              const int32_t* __restrict__ row_index, int rows_out,
                                                              const float* __restrict__ w, const float* __restrict__ b,
                                                              float eps, void* out, int ldo, LnAdd add) {
    constexpr int D = VPL * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows_out) return;
    const int in_row = (!ADD && row_index) ? row_index[row] : row;
    const float* xr = x + (size_t)in_row * ldx;
    f32x4 v[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[i] = *(const f32x4*)(xr + (lane + i * 64) * 4);
    if (ADD) {
        const uint16_t* br = add.branch + (size_t)row * add.ldb;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            v[i] = ln_add4(v[i], br, add.gamma, (lane + i * 64) * 4);
            *(f32x4*)(const_cast<float*>(xr) + (lane + i * 64) * 4) = v[i];
        }
    }
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
        if (OUT_BF16 == 2 || OUT_BF16 == 3) {
            float h[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h[j] = bf16_to_f32(f32_to_bf16(y[j]));
                l[j] = y[j] - h[j];
            }
            u32x2 ph = {pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3])};
            u32x2 pl = {pack_bf16x2(l[0], l[1]), pack_bf16x2(l[2], l[3])};
            if (OUT_BF16 == 3) {  // UFM_BF16X2_IL: [row][C / 32][hi 32 | lo 32] (a row is 2 ldo elements): the same values, interleaved per 32-channel chunk
                uint16_t* o = (uint16_t*)out + (size_t)row * (2 * ldo) + ((c * 4) >> 5) * 64 + ((c * 4) & 31);
                *(u32x2*)o = ph;
                *(u32x2*)(o + 32) = pl;
            } else {
                uint16_t* o = (uint16_t*)out + (size_t)row * ldo + c * 4;
                *(u32x2*)o = ph;
                *(u32x2*)(o + add.out_plane) = pl;
            }
        } else if (OUT_BF16 == 1) {
            u32x2 pk = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
            *(u32x2*)((uint16_t*)out + (size_t)row * ldo + c * 4) = pk;
        } else {
            *(f32x4*)((float*)out + (size_t)row * ldo + c * 4) = y;
        }
    }
}

template <int OUT_BF16, bool ADD = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx,
                                                        const int32_t* __restrict__ row_index, int rows_out,
                                                        int D, const float* __restrict__ w,
                                                        const float* __restrict__ b, float eps, void* out, int ldo, LnAdd add) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows_out) return;
    const int in_row = (!ADD && row_index) ? row_index[row] : row;
    const float* xr = x + (size_t)in_row * ldx;
    const int nvec = D >> 2;
    f32x4 v[MAX_VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_VPL; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            v[i] = *(const f32x4*)(xr + c * 4);
            if (ADD) {
                v[i] = ln_add4(v[i], add.branch + (size_t)row * add.ldb, add.gamma, c * 4);
                *(f32x4*)(const_cast<float*>(xr) + c * 4) = v[i];
            }
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
                *(u32x2*)(o + add.out_plane) = pl;
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

// out[r, :] = x[row_index[r], :] (fp32 rows, 16 B per lane)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ x, int ldx, const int32_t* __restrict__ row_index, int rows, int D,
                                                          float* __restrict__ out, int ldo) {
    const int nvec = D >> 2;
    const long long total = (long long)rows * nvec;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / nvec), c = (int)(i - (long long)r * nvec);
        *(f32x4*)(out + (size_t)r * ldo + c * 4) = *(const f32x4*)(x + (size_t)row_index[r] * ldx + c * 4);
    }
}

}  // namespace

template <bool ADD>
static void launch_layernorm(const float* x, int ldx, const int32_t* row_index, int rows_out, int D, const float* weight,
                             const float* bias, float eps, void* out, int out_dtype, int ldo, LnAdd add, hipStream_t st) {
    dim3 grid((rows_out + 3) / 4), block(256);
#define UFM_LN_FIXED(OUT, VPL) hipLaunchKernelGGL((layernorm_kernel_fixed<OUT, VPL, ADD>), grid, block, 0, st, x, ldx, row_index, rows_out, weight, bias, eps, out, ldo, add)
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
        if (out_dtype == UFM_BF16X2_IL) { UFM_LN_BY_VPL(3) } else if (out_dtype == UFM_BF16X2) { UFM_LN_BY_VPL(2) } else if (out_dtype == UFM_BF16) { UFM_LN_BY_VPL(1) } else { UFM_LN_BY_VPL(0) }
    } else if (out_dtype == UFM_BF16X2)
        hipLaunchKernelGGL((layernorm_kernel<2, ADD>), grid, block, 0, st, x, ldx, row_index, rows_out, D, weight, bias, eps, out, ldo, add);
    else if (out_dtype == UFM_BF16)
        hipLaunchKernelGGL((layernorm_kernel<1, ADD>), grid, block, 0, st, x, ldx, row_index, rows_out, D, weight, bias, eps, out, ldo, add);
    else
        hipLaunchKernelGGL((layernorm_kernel<0, ADD>), grid, block, 0, st, x, ldx, row_index, rows_out, D, weight, bias, eps, out, ldo, add);
#undef UFM_LN_BY_VPL
#undef UFM_LN_FIXED
}

extern "C" int ufm_layernorm(const float* x, int ldx, const int32_t* row_index, int rows_out, int D,
                             const float* weight, const float* bias, float eps, void* out, int out_dtype,
                             int ldo, void* stream) {
    UFM_REQUIRE(x && weight && bias && out, "ufm_layernorm: null pointer");
    UFM_REQUIRE(rows_out > 0, "ufm_layernorm: rows_out=%d", rows_out);
    UFM_REQUIRE(D % 4 == 0 && D <= MAX_VPL * 256 && D > 0, "ufm_layernorm: D=%d must be a multiple of 4 and <= %d", D, MAX_VPL * 256);
    UFM_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && ldx >= D && ldo >= D, "ufm_layernorm: bad ldx/ldo %d/%d", ldx, ldo);
    UFM_REQUIRE(out_dtype == UFM_F32 || out_dtype == UFM_BF16 || out_dtype == UFM_BF16X2 || out_dtype == UFM_BF16X2_IL, "ufm_layernorm: bad out_dtype");
    UFM_REQUIRE(out_dtype != UFM_BF16X2_IL || (D % 256 == 0 && ldo % 32 == 0), "ufm_layernorm: the interleaved split output needs D %% 256 == 0 and ldo %% 32 == 0 (D=%d, ldo=%d)", D, ldo);
    launch_layernorm<false>(x, ldx, row_index, rows_out, D, weight, bias, eps, out, out_dtype, ldo, LnAdd{nullptr, nullptr, 0, (long long)rows_out * ldo}, (hipStream_t)stream);
    UFM_CHECK_LAUNCH("ufm_layernorm");
    return UFM_OK;
}

// ufm_layernorm writing a ROW SLICE of a larger (2, rows_total, ldo) split buffer: the lo plane sits out_plane elements behind
// the hi plane (engine: the micro-batch streams' pyramid levels land directly in the full-batch buffer the heads read)
extern "C" int ufm_layernorm_slice(const float* x, int ldx, const int32_t* row_index, int rows_out, int D,
                                   const float* weight, const float* bias, float eps, void* out, int out_dtype,
                                   int ldo, long long out_plane, void* stream) {
    UFM_REQUIRE(x && weight && bias && out, "ufm_layernorm_slice: null pointer");
    UFM_REQUIRE(rows_out > 0, "ufm_layernorm_slice: rows_out=%d", rows_out);
    UFM_REQUIRE(D % 4 == 0 && D <= MAX_VPL * 256 && D > 0, "ufm_layernorm_slice: D=%d must be a multiple of 4 and <= %d", D, MAX_VPL * 256);
    UFM_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && ldx >= D && ldo >= D, "ufm_layernorm_slice: bad ldx/ldo %d/%d", ldx, ldo);
    UFM_REQUIRE(out_dtype == UFM_F32 || out_dtype == UFM_BF16 || out_dtype == UFM_BF16X2, "ufm_layernorm_slice: bad out_dtype");
    UFM_REQUIRE(out_dtype != UFM_BF16X2 || (out_plane >= (long long)rows_out * ldo && out_plane % 4 == 0), "ufm_layernorm_slice: out_plane=%lld must be >= rows_out * ldo and a multiple of 4", out_plane);
    launch_layernorm<false>(x, ldx, row_index, rows_out, D, weight, bias, eps, out, out_dtype, ldo, LnAdd{nullptr, nullptr, 0, out_plane}, (hipStream_t)stream);
    UFM_CHECK_LAUNCH("ufm_layernorm_slice");
    return UFM_OK;
}

extern "C" int ufm_add_layernorm(float* x, int ldx, const uint16_t* branch, int ldb, const float* gamma, int rows, int D,
                                 const float* weight, const float* bias, float eps, void* out, int out_dtype, int ldo,
                                 void* stream) {
    UFM_REQUIRE(x && branch && weight && bias && out, "ufm_add_layernorm: null pointer");
    UFM_REQUIRE(rows > 0, "ufm_add_layernorm: rows=%d", rows);
    UFM_REQUIRE(D % 4 == 0 && D <= MAX_VPL * 256 && D > 0, "ufm_add_layernorm: D=%d must be a multiple of 4 and <= %d", D, MAX_VPL * 256);
    UFM_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && ldb % 4 == 0 && ldx >= D && ldo >= D && ldb >= D, "ufm_add_layernorm: bad ldx/ldb/ldo %d/%d/%d", ldx, ldb, ldo);
    UFM_REQUIRE(((uintptr_t)branch % 8) == 0 && ((uintptr_t)x % 16) == 0, "ufm_add_layernorm: misaligned pointer");
    UFM_REQUIRE(out_dtype == UFM_F32 || out_dtype == UFM_BF16 || out_dtype == UFM_BF16X2, "ufm_add_layernorm: bad out_dtype");
    launch_layernorm<true>(x, ldx, nullptr, rows, D, weight, bias, eps, out, out_dtype, ldo, LnAdd{branch, gamma, ldb, (long long)rows * ldo}, (hipStream_t)stream);
    UFM_CHECK_LAUNCH("ufm_add_layernorm");
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

extern "C" int ufm_gather_rows_f32(const float* x, int ldx, const int32_t* row_index, int rows, int D, float* out, int ldo, void* stream) {
    UFM_REQUIRE(x && row_index && out, "ufm_gather_rows_f32: null pointer");
    UFM_REQUIRE(rows > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldx >= D && ldo >= D, "ufm_gather_rows_f32: bad shape rows=%d D=%d ldx=%d ldo=%d", rows, D, ldx, ldo);
    UFM_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 16) == 0, "ufm_gather_rows_f32: misaligned pointer");
    const long long total = (long long)rows * (D / 4);
    dim3 grid((unsigned)min((total + 255) / 256, (long long)4096)), block(256);
    hipLaunchKernelGGL(gather_rows_kernel, grid, block, 0, (hipStream_t)stream, x, ldx, row_index, rows, D, out, ldo);
    UFM_CHECK_LAUNCH("ufm_gather_rows_f32");
    return UFM_OK;
}
