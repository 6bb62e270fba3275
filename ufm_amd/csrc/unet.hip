// HBM-bound helpers of the UNet fine-feature path of UFM-Refine (uniflowmatch/models/unet_encoder.py:26-71 and the
// combine step of models/ufm.py:967-983).  The 3x3 / 1x1 / transposed convolutions of the UNet run on the implicit-GEMM
// conv kernels (conv_bf16x3*.hip in numerics "fast", conv_f32.hip in "parity"); what is left are layout moves:
//   ufm_image_to_nhwc        normalise + BHWC/BCHW -> NHWC with the 3 channels zero-padded to the conv kernels' K-chunk
//   ufm_maxpool2x2_nhwc      nn.MaxPool2d(2, 2)                                           (unet_encoder.py:37, :57)
//   ufm_resize_nearest_nhwc  F.interpolate(x, size) (legacy "nearest") + torch.cat slot   (unet_encoder.py:66-68)
//   ufm_unet_combine         cat -> conv1 1x1 -> ReLU -> conv2 1x1  |  cls * tanh(unet) -> conv2 1x1  (ufm.py:967-983)
// Activations are NHWC fp32 or the UFM_BF16X2 split format (two bf16 planes, lo plane at +plane elements); all kernels
// move 16 B per lane, grid-strided, one pass over the data.
#include "common.h"

namespace {

struct Affine3 {
    float scale[3];
    float shift[3];
};

inline dim3 grid_for(size_t work_items) {
    size_t blocks = (work_items + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    return dim3((unsigned)blocks);
}

template <int SPLIT>
__global__ __launch_bounds__(256) void image_to_nhwc_kernel(const void* __restrict__ img, int in_dtype, int in_layout, int B, int H, int W,
                                                            Affine3 af, void* __restrict__ out, int Cpad) {
    const size_t npix = (size_t)B * H * W, plane = npix * Cpad;
    const int chunks = Cpad >> 2;  // 4 channels per thread-iteration
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < npix * chunks; t += (size_t)gridDim.x * blockDim.x) {
        const size_t pix = t / chunks;
        const int c4 = (int)(t - pix * chunks);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c4 == 0) {
            const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((size_t)W * H));
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const size_t idx = in_layout == 0 ? (((size_t)b * H + y) * W + x) * 3 + c : (((size_t)b * 3 + c) * H + y) * W + x;
                const float raw = in_dtype == 0 ? (float)((const uint8_t*)img)[idx] : ((const float*)img)[idx];
                v[c] = in_dtype == 0 ? (raw / 255.0f - af.shift[c]) / af.scale[c] : raw * af.scale[c] + af.shift[c];  // base.py:228-229
            }
        }
        if (SPLIT) {
            uint16_t* o = (uint16_t*)out + pix * Cpad + c4 * 4;
            float h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = bf16_to_f32(f32_to_bf16(v[e]));
            *(u32x2*)o = u32x2{pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3])};
            *(u32x2*)(o + plane) = u32x2{pack_bf16x2(v[0] - h[0], v[1] - h[1]), pack_bf16x2(v[2] - h[2], v[3] - h[3])};
        } else {
            *(f32x4*)((float*)out + pix * Cpad + c4 * 4) = v;
        }
    }
}

// split format: value = hi + lo; the maximum of the four values is copied with ITS (hi, lo) pair (the split of a value is a
// function of the value, so this equals splitting the fp32 maximum)
template <int SPLIT>
__global__ __launch_bounds__(256) void maxpool2_kernel(const void* __restrict__ in, int B, int H, int W, int C, void* __restrict__ out) {
    const int Ho = H >> 1, Wo = W >> 1;
    constexpr int V = SPLIT ? 8 : 4;  // channels per 16-B access
    const int chunks = C / V;
    const size_t total = (size_t)B * Ho * Wo * chunks;
    const size_t in_plane = (size_t)B * H * W * C, out_plane = (size_t)B * Ho * Wo * C;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int ck = (int)(t % chunks);
        const size_t opix = t / chunks;
        const int ox = (int)(opix % Wo), oy = (int)((opix / Wo) % Ho), b = (int)(opix / ((size_t)Wo * Ho));
        const size_t i00 = ((((size_t)b * H + 2 * oy) * W) + 2 * ox) * C + ck * V;
        const size_t offs[4] = {i00, i00 + C, i00 + (size_t)W * C, i00 + (size_t)W * C + C};
        if (SPLIT) {
            const uint16_t* p = (const uint16_t*)in;
            float best[8];
            unsigned short bh[8], bl[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 ph = *(const u32x4*)(p + offs[q]);
                const u32x4 pl = *(const u32x4*)(p + offs[q] + in_plane);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned short hh = (unsigned short)((e & 1) ? (ph[e >> 1] >> 16) : (ph[e >> 1] & 0xffffu));
                    const unsigned short ll = (unsigned short)((e & 1) ? (pl[e >> 1] >> 16) : (pl[e >> 1] & 0xffffu));
                    const float v = bf16_to_f32(hh) + bf16_to_f32(ll);
                    if (q == 0 || v > best[e]) best[e] = v, bh[e] = hh, bl[e] = ll;
                }
            }
            u32x4 oh, ol;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                oh[e] = (unsigned)bh[2 * e] | ((unsigned)bh[2 * e + 1] << 16);
                ol[e] = (unsigned)bl[2 * e] | ((unsigned)bl[2 * e + 1] << 16);
            }
            uint16_t* o = (uint16_t*)out + opix * C + ck * V;
            *(u32x4*)o = oh;
            *(u32x4*)(o + out_plane) = ol;
        } else {
            const float* p = (const float*)in;
            f32x4 m = *(const f32x4*)(p + offs[0]);
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const f32x4 v = *(const f32x4*)(p + offs[q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
            }
            *(f32x4*)((float*)out + opix * C + ck * V) = m;
        }
    }
}

// out[b][y][x][c_off + c] = in[b][sy(y)][sx(x)][c], legacy nearest: src = min(floor(dst * (float)in / out), in - 1)
template <int SPLIT>
__global__ __launch_bounds__(256) void resize_nearest_kernel(const void* __restrict__ in, int B, int H, int W, int C, void* __restrict__ out, int Ho, int Wo,
                                                             int ldc, int c_off, float sy, float sx) {
    constexpr int V = SPLIT ? 8 : 4;
    const int chunks = C / V;
    const size_t total = (size_t)B * Ho * Wo * chunks;
    const size_t in_plane = (size_t)B * H * W * C, out_plane = (size_t)B * Ho * Wo * ldc;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int ck = (int)(t % chunks);
        const size_t opix = t / chunks;
        const int ox = (int)(opix % Wo), oy = (int)((opix / Wo) % Ho), b = (int)(opix / ((size_t)Wo * Ho));
        const int iy = min((int)floorf(oy * sy), H - 1), ix = min((int)floorf(ox * sx), W - 1);
        const size_t io = (((size_t)b * H + iy) * W + ix) * C + ck * V;
        const size_t oo = opix * ldc + c_off + ck * V;
        if (SPLIT) {
            const uint16_t* p = (const uint16_t*)in;
            uint16_t* o = (uint16_t*)out;
            *(u32x4*)(o + oo) = *(const u32x4*)(p + io);
            *(u32x4*)(o + oo + out_plane) = *(const u32x4*)(p + io + in_plane);
        } else {
            *(f32x4*)((float*)out + oo) = *(const f32x4*)((const float*)in + io);
        }
    }
}

// one thread per pixel.  cls: planar [N][16][HW]; unet: NHWC [N][HW][ldu] (first 16 channels), fp32 or split.
// method 0 ("conv"):     x = [cls | unet] (32) -> y = relu(W1 x + b1) (32) -> z = W2 y + b2 (16)
// method 1 ("modulate"): x = cls * tanh(unet) (16) -> z = W2 x + b2 (16)
template <int SPLIT>
__global__ __launch_bounds__(256) void unet_combine_kernel(const float* __restrict__ cls, const void* __restrict__ unet, int N, int HW, int ldu,
                                                           const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                           const float* __restrict__ b2, int method, float* __restrict__ out) {
    const size_t total = (size_t)N * HW, plane = total * ldu;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
        const size_t n = p / HW, q = p - n * HW;
        float x[32];
#pragma unroll
        for (int c = 0; c < 16; ++c) x[c] = cls[(n * 16 + c) * HW + q];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            f32x4 u;
            if (SPLIT) {
                const uint16_t* up = (const uint16_t*)unet + p * ldu + c4 * 4;
                const u32x2 ph = *(const u32x2*)up, pl = *(const u32x2*)(up + plane);
                u[0] = __uint_as_float(ph[0] << 16) + __uint_as_float(pl[0] << 16);
                u[1] = __uint_as_float(ph[0] & 0xffff0000u) + __uint_as_float(pl[0] & 0xffff0000u);
                u[2] = __uint_as_float(ph[1] << 16) + __uint_as_float(pl[1] << 16);
                u[3] = __uint_as_float(ph[1] & 0xffff0000u) + __uint_as_float(pl[1] & 0xffff0000u);
            } else {
                u = *(const f32x4*)((const float*)unet + p * ldu + c4 * 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) x[16 + c4 * 4 + e] = u[e];
        }
        float z[16];
        if (method == 0) {
            float y[32];
#pragma unroll
            for (int o = 0; o < 32; ++o) {
                float a = b1[o];
#pragma unroll
                for (int c = 0; c < 32; ++c) a = __builtin_fmaf(w1[o * 32 + c], x[c], a);
                y[o] = fmaxf(a, 0.0f);
            }
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                float a = b2[o];
#pragma unroll
                for (int c = 0; c < 32; ++c) a = __builtin_fmaf(w2[o * 32 + c], y[c], a);
                z[o] = a;
            }
        } else {
            float m[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) m[c] = x[c] * tanhf(x[16 + c]);
#pragma unroll
            for (int o = 0; o < 16; ++o) {
                float a = b2[o];
#pragma unroll
                for (int c = 0; c < 16; ++c) a = __builtin_fmaf(w2[o * 16 + c], m[c], a);
                z[o] = a;
            }
        }
#pragma unroll
        for (int o = 0; o < 16; ++o) out[(n * 16 + o) * HW + q] = z[o];
    }
}

}  // namespace

extern "C" int ufm_image_to_nhwc(const void* img, int in_dtype, int in_layout, int B, int H, int W, const float* scale3, const float* shift3,
                                 void* out, int out_dtype, int Cpad, void* stream) {
    UFM_REQUIRE(img && out && scale3 && shift3, "ufm_image_to_nhwc: null pointer");
    UFM_REQUIRE(B > 0 && H > 0 && W > 0 && Cpad >= 4 && Cpad % 4 == 0, "ufm_image_to_nhwc: bad shape");
    UFM_REQUIRE(out_dtype == UFM_F32 || out_dtype == UFM_BF16X2, "ufm_image_to_nhwc: out_dtype must be UFM_F32 or UFM_BF16X2");
    Affine3 af;
    for (int c = 0; c < 3; ++c) af.scale[c] = scale3[c], af.shift[c] = shift3[c];
    const size_t work = (size_t)B * H * W * (Cpad / 4);
    if (out_dtype == UFM_BF16X2)
        hipLaunchKernelGGL(image_to_nhwc_kernel<1>, grid_for(work), dim3(256), 0, (hipStream_t)stream, img, in_dtype, in_layout, B, H, W, af, out, Cpad);
    else
        hipLaunchKernelGGL(image_to_nhwc_kernel<0>, grid_for(work), dim3(256), 0, (hipStream_t)stream, img, in_dtype, in_layout, B, H, W, af, out, Cpad);
    UFM_CHECK_LAUNCH("ufm_image_to_nhwc");
    return UFM_OK;
}

extern "C" int ufm_maxpool2x2_nhwc(const void* in, int dtype, int B, int H, int W, int C, void* out, void* stream) {
    UFM_REQUIRE(in && out && B > 0 && H >= 2 && W >= 2 && C > 0 && C % 8 == 0, "ufm_maxpool2x2_nhwc: bad shape (C %% 8 == 0)");
    UFM_REQUIRE(dtype == UFM_F32 || dtype == UFM_BF16X2, "ufm_maxpool2x2_nhwc: dtype must be UFM_F32 or UFM_BF16X2");
    const size_t work = (size_t)B * (H / 2) * (W / 2) * (C / (dtype == UFM_BF16X2 ? 8 : 4));
    if (dtype == UFM_BF16X2)
        hipLaunchKernelGGL(maxpool2_kernel<1>, grid_for(work), dim3(256), 0, (hipStream_t)stream, in, B, H, W, C, out);
    else
        hipLaunchKernelGGL(maxpool2_kernel<0>, grid_for(work), dim3(256), 0, (hipStream_t)stream, in, B, H, W, C, out);
    UFM_CHECK_LAUNCH("ufm_maxpool2x2_nhwc");
    return UFM_OK;
}

extern "C" int ufm_resize_nearest_nhwc(const void* in, int dtype, int B, int H, int W, int C, void* out, int Ho, int Wo, int ldc, int c_off,
                                       void* stream) {
    UFM_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0 && ldc >= c_off + C && ldc % 8 == 0 && c_off % 8 == 0,
                "ufm_resize_nearest_nhwc: bad shape (C, ldc, c_off multiples of 8)");
    UFM_REQUIRE(dtype == UFM_F32 || dtype == UFM_BF16X2, "ufm_resize_nearest_nhwc: dtype must be UFM_F32 or UFM_BF16X2");
    const size_t work = (size_t)B * Ho * Wo * (C / (dtype == UFM_BF16X2 ? 8 : 4));
    const float sy = (float)H / (float)Ho, sx = (float)W / (float)Wo;  // torch: scale = (float)input_size / output_size
    if (dtype == UFM_BF16X2)
        hipLaunchKernelGGL(resize_nearest_kernel<1>, grid_for(work), dim3(256), 0, (hipStream_t)stream, in, B, H, W, C, out, Ho, Wo, ldc, c_off, sy, sx);
    else
        hipLaunchKernelGGL(resize_nearest_kernel<0>, grid_for(work), dim3(256), 0, (hipStream_t)stream, in, B, H, W, C, out, Ho, Wo, ldc, c_off, sy, sx);
    UFM_CHECK_LAUNCH("ufm_resize_nearest_nhwc");
    return UFM_OK;
}

extern "C" int ufm_unet_combine(const float* cls, const void* unet, int unet_dtype, int N, int HW, int ldu, const float* w1, const float* b1,
                                const float* w2, const float* b2, int method, float* out, void* stream) {
    UFM_REQUIRE(cls && unet && w2 && b2 && out && (method == 1 || (w1 && b1)), "ufm_unet_combine: null pointer");
    UFM_REQUIRE(N > 0 && HW > 0 && ldu >= 16 && ldu % 4 == 0 && (method == 0 || method == 1), "ufm_unet_combine: bad shape");
    UFM_REQUIRE(unet_dtype == UFM_F32 || unet_dtype == UFM_BF16X2, "ufm_unet_combine: unet_dtype must be UFM_F32 or UFM_BF16X2");
    const size_t work = (size_t)N * HW;
    if (unet_dtype == UFM_BF16X2)
        hipLaunchKernelGGL(unet_combine_kernel<1>, grid_for(work), dim3(256), 0, (hipStream_t)stream, cls, unet, N, HW, ldu, w1, b1, w2, b2, method, out);
    else
        hipLaunchKernelGGL(unet_combine_kernel<0>, grid_for(work), dim3(256), 0, (hipStream_t)stream, cls, unet, N, HW, ldu, w1, b1, w2, b2, method, out);
    UFM_CHECK_LAUNCH("ufm_unet_combine");
    return UFM_OK;
}
