// Post-processing of the `ufm infer` runner: warp the target image into the source frame with the
// predicted flow -- [R] utils/viz.py:11-59 warp_image_with_flow (F.grid_sample bilinear, align_corners=False, zeros
// padding, on coordinates clip(x + flow_x, 0, Wt - 1) + 0.5) -- and the covisibility blend of cli.py:141-143.
#include "common.h"

namespace {

template <int U8>
__global__ __launch_bounds__(256) void warp_bilinear_kernel(const void* __restrict__ tgt_, int Ht, int Wt,
                                                            const float* __restrict__ flow, int H, int W,
                                                            const float* __restrict__ mask, int mask_mode, float fill,
                                                            float* __restrict__ out) {
    const int total = H * W;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int y = t / W, x = t - y * W;
        // the reference's normalise / un-normalise round trip returns exactly this pixel coordinate
        const float ix = fminf(fmaxf((float)x + flow[t], 0.0f), (float)(Wt - 1));
        const float iy = fminf(fmaxf((float)y + flow[total + t], 0.0f), (float)(Ht - 1));
        const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
        const float wx1 = ix - x0, wy1 = iy - y0, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int yy = y0 + dy, xx = x0 + dx;
                if (yy < 0 || yy >= Ht || xx < 0 || xx >= Wt) continue;  // zeros padding
                const float wgt = (dy ? wy1 : wy0) * (dx ? wx1 : wx0);
                const size_t o = ((size_t)yy * Wt + xx) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    acc[c] += wgt * (U8 ? (float)((const uint8_t*)tgt_)[o + c] : ((const float*)tgt_)[o + c]);
            }
        if (mask_mode == 1) {  // viz.py:56-57  warped * (mask > 0.5)
            const float m = mask[t] > 0.5f ? 1.f : 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] *= m;
        } else if (mask_mode == 2) {  // cli.py:141-143  cov * warped + (1 - cov) * fill
            const float m = mask[t];
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = m * acc[c] + (1.f - m) * fill;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(size_t)t * 3 + c] = acc[c];
    }
}

}  // namespace

extern "C" int ufm_warp_bilinear(const void* target, int tgt_dtype, int Ht, int Wt, const float* flow, int H, int W,
                                 const float* mask, int mask_mode, float fill, float* out, void* stream) {
    UFM_REQUIRE(target && flow && out, "ufm_warp_bilinear: null pointer");
    UFM_REQUIRE(Ht > 0 && Wt > 0 && H > 0 && W > 0 && (long long)H * W < (1ll << 30), "ufm_warp_bilinear: bad shape");
    UFM_REQUIRE(tgt_dtype == 0 || tgt_dtype == 1, "ufm_warp_bilinear: tgt_dtype must be 0 (uint8) or 1 (fp32)");
    UFM_REQUIRE(mask_mode >= 0 && mask_mode <= 2 && (mask_mode == 0 || mask), "ufm_warp_bilinear: bad mask_mode / null mask");
    const int blocks = (H * W + 255) / 256;
    if (tgt_dtype == 0)
        hipLaunchKernelGGL(warp_bilinear_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, target, Ht, Wt, flow, H, W, mask, mask_mode, fill, out);
    else
        hipLaunchKernelGGL(warp_bilinear_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, target, Ht, Wt, flow, H, W, mask, mask_mode, fill, out);
    UFM_CHECK_LAUNCH("ufm_warp_bilinear");
    return UFM_OK;
}
