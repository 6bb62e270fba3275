// Layout / normalisation kernels of the "moge_conv" prediction head ([U] uniception MoGeConvFeature = the convolutional
// head of MoGe; call site models/ufm.py:266-267).  Its convolutions run on ufm_conv2d_nhwc_f32 / ufm_conv2d_nhwc_bf16x3
// (1x1 projections summed through the residual input, ConvTranspose2d(k=s=2) in shuffle mode, 3x3 with padding_mode
// "replicate"); these are the HBM-bound pieces in between, all on NHWC fp32 or UFM_BF16X2 maps:
//   ufm_group_norm_nhwc      GroupNorm(G, C) (+ ReLU): deterministic two-stage statistics (fp32 partial sums per pixel
//                            chunk, combined in double in a fixed order), then normalise + affine (+ ReLU)
//   ufm_fill_uv_nhwc         the two view-plane coordinate channels MoGe concatenates in front of every stage
//                            (normalized_view_plane_uv), written into channels [c_off, c_off + 2) of a wider map, the
//                            channels up to the next multiple of 32 zero-filled (the conv kernels' K chunk)
//   ufm_resize_bilinear_nhwc F.interpolate(mode="bilinear", align_corners=False) into channels [c_off, c_off + C) of a
//                            wider map (the torch.cat slot)
#include "common.h"

namespace {

constexpr int GN_CHUNK = 128;  // pixels per partial-sum block

__device__ __forceinline__ f32x4 load4(const void* base, int fmt, long long plane, size_t idx) {
    if (fmt == UFM_F32) return *(const f32x4*)((const float*)base + idx);
    const uint16_t* p = (const uint16_t*)base + idx;
    const u32x2 h = *(const u32x2*)p, l = *(const u32x2*)(p + plane);
    return f32x4{__uint_as_float(h[0] << 16) + __uint_as_float(l[0] << 16), __uint_as_float(h[0] & 0xffff0000u) + __uint_as_float(l[0] & 0xffff0000u),
                 __uint_as_float(h[1] << 16) + __uint_as_float(l[1] << 16), __uint_as_float(h[1] & 0xffff0000u) + __uint_as_float(l[1] & 0xffff0000u)};
}

__device__ __forceinline__ void store4(void* base, int fmt, long long plane, size_t idx, const f32x4& v) {
    if (fmt == UFM_F32) {
        *(f32x4*)((float*)base + idx) = v;
        return;
    }
    float h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = bf16_to_f32(f32_to_bf16(v[j]));
        l[j] = v[j] - h[j];
    }
    uint16_t* p = (uint16_t*)base + idx;
    *(u32x2*)p = u32x2{pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3])};
    *(u32x2*)(p + plane) = u32x2{pack_bf16x2(l[0], l[1]), pack_bf16x2(l[2], l[3])};
}

// partial[b][chunk][g] = (sum, sum of squares) over GN_CHUNK pixels x (C / G) channels.  One block per (chunk, b);
// thread t owns channel quad t % (C/4) and pixels t / (C/4), t / (C/4) + stride, ...; per-group LDS tree in a fixed order.
__global__ __launch_bounds__(256) void gn_partial_kernel(const void* x, int fmt, long long plane, int HW, int C, int ldc, int G,
                                                         float* __restrict__ partial, int nchunk) {
    __shared__ float s_sum[256], s_sq[256];
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int cq = C / 4;                 // channel quads per pixel
    const int lanes_px = 256 / cq > 0 ? 256 / cq : 1;
    const int q = tid % cq, pl = tid / cq;
    float s = 0.f, ss = 0.f;
    if (tid < lanes_px * cq) {
        const int p0 = chunk * GN_CHUNK, p1 = min(p0 + GN_CHUNK, HW);
        for (int px = p0 + pl; px < p1; px += lanes_px) {
            const f32x4 v = load4(x, fmt, plane, ((size_t)b * HW + px) * ldc + q * 4);
            s += (v[0] + v[1]) + (v[2] + v[3]);
            ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
    }
    s_sum[tid] = s;
    s_sq[tid] = ss;
    __syncthreads();
    // group g = channel quads [g * cq / G, (g + 1) * cq / G): thread g sums its group's entries in index order
    if (tid < G) {
        const int qpg = cq / G;
        float a = 0.f, a2 = 0.f;
        for (int pl2 = 0; pl2 < lanes_px; ++pl2)
            for (int qq = tid * qpg; qq < (tid + 1) * qpg; ++qq) {
                a += s_sum[pl2 * cq + qq];
                a2 += s_sq[pl2 * cq + qq];
            }
        float* o = partial + (((size_t)b * nchunk + chunk) * G + tid) * 2;
        o[0] = a;
        o[1] = a2;
    }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const void* x, void* out, int fmt, long long plane, int HW, int C, int ldc, int G,
                                                       const float* __restrict__ partial, int nchunk, const float* __restrict__ w,
                                                       const float* __restrict__ bvec, float eps, int relu) {
    __shared__ float s_mean[64], s_rstd[64];
    const int b = blockIdx.y, tid = threadIdx.x;
    if (tid < G) {  // every block recombines the partials of its image in the same fixed order, in double
        double a = 0.0, a2 = 0.0;
        for (int c = 0; c < nchunk; ++c) {
            const float* pp = partial + (((size_t)b * nchunk + c) * G + tid) * 2;
            a += (double)pp[0];
            a2 += (double)pp[1];
        }
        const double n = (double)HW * (double)(C / G);
        const double mean = a / n;
        double var = a2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        s_mean[tid] = (float)mean;
        s_rstd[tid] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int cq = C / 4;
    const long long total = (long long)HW * cq;
    for (long long i = (long long)blockIdx.x * blockDim.x + tid; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int px = (int)(i / cq), q = (int)(i % cq);
        const int g = q / (cq / G);
        const size_t idx = ((size_t)b * HW + px) * ldc + q * 4;
        f32x4 v = load4(x, fmt, plane, idx);
        const f32x4 wv = *(const f32x4*)(w + q * 4), bv = *(const f32x4*)(bvec + q * 4);
        const float mean = s_mean[g], rstd = s_rstd[g];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = (v[j] - mean) * rstd * wv[j] + bv[j];
            if (relu) v[j] = fmaxf(v[j], 0.0f);
        }
        store4(out, fmt, plane, idx, v);
    }
}

// channels [c_off, c_off+2) = (u, v) of the pixel centre, [c_off+2, c_end) = 0; one thread per pixel.
// u = -span_x (W-1)/W + x * step_x  with torch.linspace's symmetric evaluation (start + i*step for i < W/2, end - (W-1-i)*step
// otherwise), the form the oracle's torch.linspace produces.
__global__ __launch_bounds__(256) void fill_uv_kernel(void* out, int fmt, long long plane, int B, int H, int W, int ldc, int c_off, int c_end,
                                                      float u0, float u1, float ustep, float v0, float v1, float vstep) {
    const long long total = (long long)B * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const float u = x < W / 2 ? u0 + ustep * (float)x : u1 - ustep * (float)(W - 1 - x);
    const float v = y < H / 2 ? v0 + vstep * (float)y : v1 - vstep * (float)(H - 1 - y);
    for (int c = c_off; c < c_end; ++c) {
        const float val = c == c_off ? u : (c == c_off + 1 ? v : 0.0f);
        const size_t idx = (size_t)i * ldc + c;
        if (fmt == UFM_F32) {
            ((float*)out)[idx] = val;
        } else {
            const uint16_t h = f32_to_bf16(val);
            ((uint16_t*)out)[idx] = h;
            ((uint16_t*)out)[idx + plane] = f32_to_bf16(val - bf16_to_f32(h));
        }
    }
}

// torch's bilinear, align_corners=False: src = max((dst + 0.5) * in/out - 0.5, 0), i1 = min(i0 + 1, in - 1)
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const void* in, void* out, int fmt, long long in_plane, long long out_plane, int B, int H,
                                                              int W, int C, int ldi, int Ho, int Wo, int ldc, int c_off, float sy, float sx) {
    const int cq = C / 4;
    const long long total = (long long)B * Ho * Wo * cq;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i % cq);
    const long long px = i / cq;
    const int ox = (int)(px % Wo), oy = (int)((px / Wo) % Ho), b = (int)(px / ((long long)Wo * Ho));
    const float fy = fmaxf(((float)oy + 0.5f) * sy - 0.5f, 0.0f), fx = fmaxf(((float)ox + 0.5f) * sx - 0.5f, 0.0f);
    const int y0 = min((int)fy, H - 1), x0 = min((int)fx, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const size_t base = (size_t)b * H * W;
    const f32x4 v00 = load4(in, fmt, in_plane, (base + (size_t)y0 * W + x0) * ldi + q * 4);
    const f32x4 v01 = load4(in, fmt, in_plane, (base + (size_t)y0 * W + x1) * ldi + q * 4);
    const f32x4 v10 = load4(in, fmt, in_plane, (base + (size_t)y1 * W + x0) * ldi + q * 4);
    const f32x4 v11 = load4(in, fmt, in_plane, (base + (size_t)y1 * W + x1) * ldi + q * 4);
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j)  // torch: (1-ly) * ((1-lx) v00 + lx v01) + ly * ((1-lx) v10 + lx v11)
        r[j] = (1.0f - ly) * ((1.0f - lx) * v00[j] + lx * v01[j]) + ly * ((1.0f - lx) * v10[j] + lx * v11[j]);
    store4(out, fmt, out_plane, (size_t)px * ldc + c_off + q * 4, r);
}

}  // namespace

extern "C" int ufm_group_norm_nhwc(const void* in, int dtype, int B, int HW, int C, int ldc, int groups, const float* weight,
                                   const float* bias, float eps, int relu, void* out, float* partial_ws, void* stream) {
    UFM_REQUIRE(in && out && weight && bias && partial_ws, "ufm_group_norm_nhwc: null pointer");
    UFM_REQUIRE(dtype == UFM_F32 || dtype == UFM_BF16X2, "ufm_group_norm_nhwc: dtype must be UFM_F32 or UFM_BF16X2");
    UFM_REQUIRE(B > 0 && HW > 0 && C > 0 && groups > 0 && groups <= 64 && C % (4 * groups) == 0 && C <= 1024 && ldc >= C && ldc % 4 == 0,
                "ufm_group_norm_nhwc: bad shape C=%d groups=%d ldc=%d", C, groups, ldc);
    const int nchunk = (HW + GN_CHUNK - 1) / GN_CHUNK;
    const long long plane = (long long)B * HW * ldc;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, B), dim3(256), 0, st, in, dtype, plane, HW, C, ldc, groups, partial_ws, nchunk);
    const long long per_img = (long long)HW * (C / 4);
    const int gx = (int)((per_img + 255) / 256 < 1024 ? (per_img + 255) / 256 : 1024);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(gx, B), dim3(256), 0, st, in, out, dtype, plane, HW, C, ldc, groups, partial_ws, nchunk, weight, bias, eps, relu);
    UFM_CHECK_LAUNCH("ufm_group_norm_nhwc");
    return UFM_OK;
}

extern "C" int ufm_group_norm_ws_floats(int B, int HW, int groups) { return B * ((HW + GN_CHUNK - 1) / GN_CHUNK) * groups * 2; }

extern "C" int ufm_fill_uv_nhwc(void* out, int dtype, int B, int H, int W, int ldc, int c_off, float aspect_ratio, void* stream) {
    UFM_REQUIRE(out, "ufm_fill_uv_nhwc: null pointer");
    UFM_REQUIRE(dtype == UFM_F32 || dtype == UFM_BF16X2, "ufm_fill_uv_nhwc: dtype must be UFM_F32 or UFM_BF16X2");
    UFM_REQUIRE(B > 0 && H > 0 && W > 0 && c_off >= 0 && c_off + 2 <= ldc, "ufm_fill_uv_nhwc: bad shape");
    // normalized_view_plane_uv: spans of the unit diagonal, pixel centres; torch.linspace(start, end, n) in fp32
    const double span_x = (double)aspect_ratio / sqrt(1.0 + (double)aspect_ratio * aspect_ratio), span_y = 1.0 / sqrt(1.0 + (double)aspect_ratio * aspect_ratio);
    const float u0 = (float)(-span_x * (W - 1) / W), u1 = (float)(span_x * (W - 1) / W);
    const float v0 = (float)(-span_y * (H - 1) / H), v1 = (float)(span_y * (H - 1) / H);
    const float ustep = W > 1 ? (u1 - u0) / (float)(W - 1) : 0.0f, vstep = H > 1 ? (v1 - v0) / (float)(H - 1) : 0.0f;
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(fill_uv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, dtype, total * ldc, B, H, W, ldc, c_off, ldc,
                       u0, u1, ustep, v0, v1, vstep);
    UFM_CHECK_LAUNCH("ufm_fill_uv_nhwc");
    return UFM_OK;
}

extern "C" int ufm_resize_bilinear_nhwc(const void* in, int dtype, int B, int H, int W, int C, int ldi, void* out, int Ho, int Wo, int ldc,
                                        int c_off, void* stream) {
    UFM_REQUIRE(in && out, "ufm_resize_bilinear_nhwc: null pointer");
    UFM_REQUIRE(dtype == UFM_F32 || dtype == UFM_BF16X2, "ufm_resize_bilinear_nhwc: dtype must be UFM_F32 or UFM_BF16X2");
    UFM_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 4 == 0 && ldi >= C && ldi % 4 == 0 && c_off % 4 == 0 && c_off + C <= ldc && ldc % 4 == 0,
                "ufm_resize_bilinear_nhwc: bad shape");
    const long long total = (long long)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, dtype,
                       (long long)B * H * W * ldi, (long long)B * Ho * Wo * ldc, B, H, W, C, ldi, Ho, Wo, ldc, c_off, (float)H / (float)Ho, (float)W / (float)Wo);
    UFM_CHECK_LAUNCH("ufm_resize_bilinear_nhwc");
    return UFM_OK;
}
