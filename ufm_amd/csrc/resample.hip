// Resampling kernels: antialiased bilinear image resize (separable, torch "_upsample_bilinear2d_aa"
// semantics, utils/flow_resizing.py:313-326) and the fused UFM-Refine classification refinement
// (bicubic neighbourhood gather + P*P softmax + weighted offsets, models/ufm.py:1012-1178).
#include "common.h"

namespace {

struct Affine3 {
    float scale[3];
    float shift[3];
};

__device__ __forceinline__ float tri(float x) {
    x = fabsf(x);
    return x < 1.f ? 1.f - x : 0.f;
}

// antialias filter window for output index i along one axis (ATen _compute_indices_min_size_weights_aa)
__device__ __forceinline__ void aa_window(int i, int in_size, float scale, int& xmin, int& xsize, float& center,
                                          float& invscale) {
    const float support = scale >= 1.f ? scale : 1.f;
    invscale = scale >= 1.f ? 1.f / scale : 1.f;
    center = scale * (i + 0.5f);
    xmin = max((int)(center - support + 0.5f), 0);
    xsize = min((int)(center + support + 0.5f), in_size) - xmin;
}

// pass 1: horizontal. in: u8/f32, BHWC/BCHW -> tmp f32 [B][3][H][Wo]; normalisation applied on load.
__global__ __launch_bounds__(256) void aa_h_kernel(const void* img, int in_dtype, int in_layout, int B, int H, int W,
                                                   Affine3 af, float* __restrict__ tmp, int Wo, float scale) {
    const size_t total = (size_t)B * 3 * H * Wo;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(t % Wo), y = (int)((t / Wo) % H), c = (int)((t / ((size_t)Wo * H)) % 3);
        const int b = (int)(t / ((size_t)Wo * H * 3));
        int xmin, xsize;
        float center, inv;
        aa_window(ox, W, scale, xmin, xsize, center, inv);
        float wsum = 0.f;
        for (int j = 0; j < xsize; ++j) wsum += tri((j + xmin - center + 0.5f) * inv);
        float acc = 0.f;
        for (int j = 0; j < xsize; ++j) {
            const float wgt = tri((j + xmin - center + 0.5f) * inv) / wsum;
            const int x = xmin + j;
            const size_t idx = in_layout == 0 ? (((size_t)b * H + y) * W + x) * 3 + c : (((size_t)b * 3 + c) * H + y) * W + x;
            float v;
            if (in_dtype == 0)
                v = ((float)((const uint8_t*)img)[idx] / 255.0f - af.shift[c]) / af.scale[c];
            else
                v = ((const float*)img)[idx] * af.scale[c] + af.shift[c];
            acc += v * wgt;
        }
        tmp[t] = acc;
    }
}

// pass 2: vertical. tmp [B*3][H][Wo] -> out [B*3][Ho][Wo]
__global__ __launch_bounds__(256) void aa_v_kernel(const float* __restrict__ tmp, int BC, int H, int Wo,
                                                   float* __restrict__ out, int Ho, float scale) {
    const size_t total = (size_t)BC * Ho * Wo;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(t % Wo), oy = (int)((t / Wo) % Ho);
        const int bc = (int)(t / ((size_t)Wo * Ho));
        int ymin, ysize;
        float center, inv;
        aa_window(oy, H, scale, ymin, ysize, center, inv);
        float wsum = 0.f;
        for (int j = 0; j < ysize; ++j) wsum += tri((j + ymin - center + 0.5f) * inv);
        float acc = 0.f;
        for (int j = 0; j < ysize; ++j) {
            const float wgt = tri((j + ymin - center + 0.5f) * inv) / wsum;
            acc += tmp[((size_t)bc * H + ymin + j) * Wo + ox] * wgt;
        }
        out[t] = acc;
    }
}

// cubic convolution coefficients, A = -0.75 (ATen get_cubic_upsample_coefficients)
__device__ __forceinline__ void cubic_coef(float t, float* c) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
    c[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    c[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    c[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    c[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

// One thread per pixel.  The P*P sample points differ by integer offsets, so they share one set
// of cubic weights and one (P+3)^2 tap window per channel: horizontal filter (P+3 rows x P cols)
// then vertical -- the same association as grid_sample's nested cubic_interp1d.
template <int P>
__global__ __launch_bounds__(256) void refine_kernel(const float* __restrict__ flow, const float* __restrict__ feat,
                                                     int B, int C, int H, int W, float temperature,
                                                     const float* __restrict__ bias, float* __restrict__ residual,
                                                     float* __restrict__ logp) {
    constexpr int R = (P - 1) / 2, T = P + 3;
    const size_t HW = (size_t)H * W;
    const size_t total = (size_t)B * HW;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(t % W), y = (int)((t / W) % H), b = (int)(t / HW);
        const float fx = flow[((size_t)b * 2) * HW + (size_t)y * W + x] + (float)x;
        const float fy = flow[((size_t)b * 2 + 1) * HW + (size_t)y * W + x] + (float)y;
        // normalise / un-normalise round trip of ufm.py:1165-1166 + grid_sample(align_corners=False)
        const float ux = ((((fx + 0.5f) / (float)W) * 2.f - 1.f + 1.f) * (float)W - 1.f) / 2.f;
        const float uy = ((((fy + 0.5f) / (float)H) * 2.f - 1.f + 1.f) * (float)H - 1.f) / 2.f;
        const float flx = floorf(ux), fly = floorf(uy);
        float wx[4], wy[4];
        cubic_coef(ux - flx, wx);
        cubic_coef(uy - fly, wy);
        const int bx = (int)flx - 1 - R, by = (int)fly - 1 - R;  // top-left tap of the window
        float score[P * P];
#pragma unroll
        for (int i = 0; i < P * P; ++i) score[i] = 0.f;
        const float* f1 = feat + (size_t)b * C * HW + (size_t)y * W + x;
        const float* f2 = feat + (size_t)(B + b) * C * HW;
        for (int c = 0; c < C; ++c) {
            const float qv = f1[(size_t)c * HW];
            const float* img = f2 + (size_t)c * HW;
            float hrow[T][P];
#pragma unroll
            for (int r = 0; r < T; ++r) {
                const int yy = by + r;
                float v[T];
                const bool yok = (unsigned)yy < (unsigned)H;
#pragma unroll
                for (int k = 0; k < T; ++k) {
                    const int xx = bx + k;
                    v[k] = (yok && (unsigned)xx < (unsigned)W) ? img[(size_t)yy * W + xx] : 0.f;
                }
#pragma unroll
                for (int dj = 0; dj < P; ++dj)
                    hrow[r][dj] = ((v[dj] * wx[0] + v[dj + 1] * wx[1]) + v[dj + 2] * wx[2]) + v[dj + 3] * wx[3];
            }
#pragma unroll
            for (int di = 0; di < P; ++di)
#pragma unroll
                for (int dj = 0; dj < P; ++dj) {
                    float s = 0.f;
#pragma unroll
                    for (int a = 0; a < 4; ++a) s += hrow[di + a][dj] * wy[a];
                    score[di * P + dj] += qv * s;
                }
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int i = 0; i < P * P; ++i) {
            score[i] = score[i] / temperature + bias[i];
            mx = fmaxf(mx, score[i]);
        }
        float sum = 0.f, rx = 0.f, ry = 0.f;
#pragma unroll
        for (int i = 0; i < P * P; ++i) {
            const float e = expf(score[i] - mx);
            sum += e;
            rx += e * (float)(i % P - R);
            ry += e * (float)(i / P - R);
        }
        residual[((size_t)b * 2) * HW + (size_t)y * W + x] = rx / sum;
        residual[((size_t)b * 2 + 1) * HW + (size_t)y * W + x] = ry / sum;
        if (logp) {
            const float lse = mx + logf(sum);
#pragma unroll
            for (int i = 0; i < P * P; ++i) logp[t * (P * P) + i] = score[i] - lse;
        }
    }
}

inline dim3 stream_grid(size_t work_items) {
    size_t blocks = (work_items + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    return dim3((unsigned)blocks);
}

}  // namespace

extern "C" int ufm_resize_antialias(const void* img, int in_dtype, int in_layout, int B, int H, int W,
                                    const float* scale3, const float* shift3, float* out, int Ho, int Wo, float* tmp,
                                    void* stream) {
    UFM_REQUIRE(img && out && tmp && scale3 && shift3, "ufm_resize_antialias: null pointer");
    UFM_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "ufm_resize_antialias: bad shape");
    UFM_REQUIRE((in_dtype == 0 || in_dtype == 1) && (in_layout == 0 || in_layout == 1), "ufm_resize_antialias: bad dtype/layout");
    Affine3 af;
    for (int c = 0; c < 3; ++c) {
        af.scale[c] = scale3[c];
        af.shift[c] = shift3[c];
    }
    hipLaunchKernelGGL(aa_h_kernel, stream_grid((size_t)B * 3 * H * Wo), dim3(256), 0, (hipStream_t)stream, img, in_dtype, in_layout, B, H, W, af, tmp, Wo, (float)W / (float)Wo);
    hipLaunchKernelGGL(aa_v_kernel, stream_grid((size_t)B * 3 * Ho * Wo), dim3(256), 0, (hipStream_t)stream, tmp, B * 3, H, Wo, out, Ho, (float)H / (float)Ho);
    UFM_CHECK_LAUNCH("ufm_resize_antialias");
    return UFM_OK;
}

extern "C" int ufm_refine(const float* flow, const float* feat, int B, int C, int H, int W, int P, float temperature,
                          const float* bias, float* residual, float* log_softmax, void* stream) {
    UFM_REQUIRE(flow && feat && bias && residual, "ufm_refine: null pointer");
    UFM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && temperature != 0.f, "ufm_refine: bad shape");
    UFM_REQUIRE(P == 3 || P == 5 || P == 7, "ufm_refine: refinement_range P=%d not in {3,5,7}", P);
    const dim3 grid = stream_grid((size_t)B * H * W);
    const float it = temperature;
    if (P == 3)
        hipLaunchKernelGGL(refine_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, flow, feat, B, C, H, W, it, bias, residual, log_softmax);
    else if (P == 5)
        hipLaunchKernelGGL(refine_kernel<5>, grid, dim3(256), 0, (hipStream_t)stream, flow, feat, B, C, H, W, it, bias, residual, log_softmax);
    else
        hipLaunchKernelGGL(refine_kernel<7>, grid, dim3(256), 0, (hipStream_t)stream, flow, feat, B, C, H, W, it, bias, residual, log_softmax);
    UFM_CHECK_LAUNCH("ufm_refine");
    return UFM_OK;
}
