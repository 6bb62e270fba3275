// RoPE-2D on the q / k columns of a projection output ([U] the `custom_positional_encoding` of uniception's cross-attention
// info-sharing variant = CroCo RoPE2D; call site models/ufm.py:193).  Every 64-wide head is two 32-wide halves (first: the
// token's y index, second: its x index); inside a half the pairs are (j, j ^ 16):
//     out[j] = v[j] * cos[t][j] + v[j ^ 16] * sin[t][j]        (sin carries the sign: -sin for j % 32 < 16, +sin otherwise)
// with t = row % mod the token's index in its image and cos / sin fp32 tables [mod][64] built by the host from the grid
// positions.  In numerics "fast" the rotation is FUSED into the QKV GEMM's epilogue (ufm_gemm_bf16_rope, gemm_common.h:
// it runs on the fp32 accumulator before the bf16 rounding); this file is the standalone, in-place form for the fp32 and
// split formats ("parity" / "precise") and the reference point of the fused one.  HBM-bound: one read + one write.
#include "common.h"

namespace {

template <int FMT>  // UFM_F32 / UFM_BF16 / UFM_BF16X2
__global__ __launch_bounds__(256) void rope2d_kernel(void* x, long long plane, int rows, int ld, int col0, int ncols,
                                                     const float* __restrict__ cosT, const float* __restrict__ sinT, int mod) {
    // one thread = 4 consecutive columns of one row and their 4 partners (j ^ 16): a pair of 16-byte (fp32) chunks
    const int per_row = ncols / 8;  // threads per row: each owns columns [c, c+4) and [c+16, c+20) of a 32-wide half
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)rows * per_row) return;
    const int row = (int)(gid / per_row), u = (int)(gid % per_row);
    const int half32 = u / 4, q4 = u % 4;              // which 32-wide half of the column range, which 4-column group of its first 16
    const int c = half32 * 32 + q4 * 4;                // first column (relative to col0); partner block at c + 16
    const int t = row % mod;
    const int tj = c & 63;                             // column inside the head
    const f32x4 ca = *(const f32x4*)(cosT + (size_t)t * 64 + tj), cb = *(const f32x4*)(cosT + (size_t)t * 64 + tj + 16);
    const f32x4 sa = *(const f32x4*)(sinT + (size_t)t * 64 + tj), sb = *(const f32x4*)(sinT + (size_t)t * 64 + tj + 16);
    f32x4 a, b;
    if (FMT == UFM_F32) {
        float* p = (float*)x + (size_t)row * ld + col0 + c;
        a = *(const f32x4*)p;
        b = *(const f32x4*)(p + 16);
        *(f32x4*)p = a * ca + b * sa;
        *(f32x4*)(p + 16) = b * cb + a * sb;
    } else if (FMT == UFM_BF16) {
        uint16_t* p = (uint16_t*)x + (size_t)row * ld + col0 + c;
        const u32x2 pa = *(const u32x2*)p, pb = *(const u32x2*)(p + 16);
        a = f32x4{__uint_as_float(pa[0] << 16), __uint_as_float(pa[0] & 0xffff0000u), __uint_as_float(pa[1] << 16), __uint_as_float(pa[1] & 0xffff0000u)};
        b = f32x4{__uint_as_float(pb[0] << 16), __uint_as_float(pb[0] & 0xffff0000u), __uint_as_float(pb[1] << 16), __uint_as_float(pb[1] & 0xffff0000u)};
        const f32x4 ra = a * ca + b * sa, rb = b * cb + a * sb;
        *(u32x2*)p = u32x2{pack_bf16x2(ra[0], ra[1]), pack_bf16x2(ra[2], ra[3])};
        *(u32x2*)(p + 16) = u32x2{pack_bf16x2(rb[0], rb[1]), pack_bf16x2(rb[2], rb[3])};
    } else {  // split planes: value = hi + lo, rotated in fp32, re-split
        uint16_t* p = (uint16_t*)x + (size_t)row * ld + col0 + c;
        auto ld4 = [&](const uint16_t* q) {
            const u32x2 h = *(const u32x2*)q, l = *(const u32x2*)(q + plane);
            return f32x4{__uint_as_float(h[0] << 16) + __uint_as_float(l[0] << 16), __uint_as_float(h[0] & 0xffff0000u) + __uint_as_float(l[0] & 0xffff0000u),
                         __uint_as_float(h[1] << 16) + __uint_as_float(l[1] << 16), __uint_as_float(h[1] & 0xffff0000u) + __uint_as_float(l[1] & 0xffff0000u)};
        };
        auto st4 = [&](uint16_t* q, const f32x4& v) {
            float h[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h[j] = bf16_to_f32(f32_to_bf16(v[j]));
                l[j] = v[j] - h[j];
            }
            *(u32x2*)q = u32x2{pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3])};
            *(u32x2*)(q + plane) = u32x2{pack_bf16x2(l[0], l[1]), pack_bf16x2(l[2], l[3])};
        };
        a = ld4(p);
        b = ld4(p + 16);
        st4(p, a * ca + b * sa);
        st4(p + 16, b * cb + a * sb);
    }
}

}  // namespace

extern "C" int ufm_rope2d(void* x, int dtype, int rows, int ld, int col0, int ncols, const float* cos_table,
                          const float* sin_table, int mod, void* stream) {
    UFM_REQUIRE(x && cos_table && sin_table, "ufm_rope2d: null pointer");
    UFM_REQUIRE(rows > 0 && mod > 0 && ncols > 0 && ncols % 64 == 0 && col0 % 64 == 0 && col0 >= 0 && col0 + ncols <= ld, "ufm_rope2d: bad shape rows=%d ld=%d col0=%d ncols=%d mod=%d", rows, ld, col0, ncols, mod);
    UFM_REQUIRE(dtype == UFM_F32 || dtype == UFM_BF16 || dtype == UFM_BF16X2, "ufm_rope2d: bad dtype");
    UFM_REQUIRE(ld % 4 == 0 && ((uintptr_t)x % 16) == 0, "ufm_rope2d: misaligned");
    const long long n = (long long)rows * (ncols / 8);
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    const long long plane = (long long)rows * ld;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == UFM_F32) hipLaunchKernelGGL(rope2d_kernel<UFM_F32>, grid, block, 0, st, x, plane, rows, ld, col0, ncols, cos_table, sin_table, mod);
    else if (dtype == UFM_BF16) hipLaunchKernelGGL(rope2d_kernel<UFM_BF16>, grid, block, 0, st, x, plane, rows, ld, col0, ncols, cos_table, sin_table, mod);
    else hipLaunchKernelGGL(rope2d_kernel<UFM_BF16X2>, grid, block, 0, st, x, plane, rows, ld, col0, ncols, cos_table, sin_table, mod);
    UFM_CHECK_LAUNCH("ufm_rope2d");
    return UFM_OK;
}
