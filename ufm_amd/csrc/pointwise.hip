// HBM-bound pointwise / gather kernels of the UFM hot path: patchify (+normalise), bilinear
// upsample (align_corners=True, NHWC), head tail (1x1 conv + adaptor), flow / channel un-mapping,
// pixel shuffle, casts.  All are coalesced 8-16 B per lane, grid-strided, one pass over the data.
#include "common.h"

namespace {

struct Affine3 {
    float scale[3];
    float shift[3];
};

__device__ __forceinline__ float load_px(const void* img, int in_dtype, int in_layout, int b, int c, int y, int x,
                                         int H, int W) {
    size_t idx = in_layout == 0 ? (((size_t)b * H + y) * W + x) * 3 + c : (((size_t)b * 3 + c) * H + y) * W + x;
    return in_dtype == 0 ? (float)((const uint8_t*)img)[idx] : ((const float*)img)[idx];
}

// out[(b,py,px)][c*P*P + i*P + j], 4 consecutive k per thread.
template <int OUT_BF16>
__global__ __launch_bounds__(256) void patchify_kernel(const void* img, int in_dtype, int in_layout, int B, int H,
                                                       int W, int P, Affine3 af, void* out, int kpad) {
    const int gh = H / P, gw = W / P;
    const int kq = kpad >> 2;
    const size_t total = (size_t)B * gh * gw * kq;
    const int kreal = 3 * P * P;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int k4 = (int)(t % kq);
        const size_t row = t / kq;
        const int px = (int)(row % gw);
        const int py = (int)((row / gw) % gh);
        const int b = (int)(row / ((size_t)gw * gh));
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k4 * 4 + e;
            if (k < kreal) {
                const int c = k / (P * P), ij = k - c * P * P, i = ij / P, j = ij - i * P;
                const float raw = load_px(img, in_dtype, in_layout, b, c, py * P + i, px * P + j, H, W);
                v[e] = in_dtype == 0 ? (raw / 255.0f - af.shift[c]) / af.scale[c]  // (x/255 - mean)/std, base.py:228
                                     : raw * af.scale[c] + af.shift[c];
            } else {
                v[e] = 0.f;
            }
        }
        if (OUT_BF16) {
            u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            *(u32x2*)((uint16_t*)out + row * kpad + k4 * 4) = pk;
        } else {
            *(f32x4*)((float*)out + row * kpad + k4 * 4) = f32x4{v[0], v[1], v[2], v[3]};
        }
    }
}

__device__ __forceinline__ f32x4 ld_split4(const uint16_t* hi_ptr, size_t plane) {
    const u32x2 ph = *(const u32x2*)hi_ptr;
    const u32x2 pl = *(const u32x2*)(hi_ptr + plane);
    f32x4 v;
    v[0] = __uint_as_float(ph[0] << 16) + __uint_as_float(pl[0] << 16);
    v[1] = __uint_as_float(ph[0] & 0xffff0000u) + __uint_as_float(pl[0] & 0xffff0000u);
    v[2] = __uint_as_float(ph[1] << 16) + __uint_as_float(pl[1] << 16);
    v[3] = __uint_as_float(ph[1] & 0xffff0000u) + __uint_as_float(pl[1] & 0xffff0000u);
    return v;
}
__device__ __forceinline__ void st_split4(uint16_t* hi_ptr, size_t plane, const f32x4& v) {
    float h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = bf16_to_f32(f32_to_bf16(v[j]));
        l[j] = v[j] - h[j];
    }
    u32x2 ph = {pack_bf16x2(h[0], h[1]), pack_bf16x2(h[2], h[3])};
    u32x2 pl = {pack_bf16x2(l[0], l[1]), pack_bf16x2(l[2], l[3])};
    *(u32x2*)hi_ptr = ph;
    *(u32x2*)(hi_ptr + plane) = pl;
}

// align_corners=True bilinear, NHWC, float4 over channels.
// (Ho, Wo) is the stored extent; sy/sx come from the FULL output size (a cropped store, DPT refinenet4).
template <int SPLIT>
__global__ __launch_bounds__(256) void upsample_kernel(const void* __restrict__ in_, int B, int H, int W, int C,
                                                       void* __restrict__ out_, int Ho, int Wo, float sy, float sx) {
    const size_t in_plane = (size_t)B * H * W * C, out_plane = (size_t)B * Ho * Wo * C;
    const int cq = C >> 2;
    const size_t total = (size_t)B * Ho * Wo * cq;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(t % cq);
        size_t pix = t / cq;
        const int ox = (int)(pix % Wo);
        const int oy = (int)((pix / Wo) % Ho);
        const int b = (int)(pix / ((size_t)Wo * Ho));
        const float fy = sy * oy, fx = sx * ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly1 = fy - y0, lx1 = fx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const size_t boff = (size_t)b * H * W * C + c4 * 4;
        const size_t o00 = boff + ((size_t)y0 * W + x0) * C, o01 = boff + ((size_t)y0 * W + x1) * C;
        const size_t o10 = boff + ((size_t)y1 * W + x0) * C, o11 = boff + ((size_t)y1 * W + x1) * C;
        f32x4 v00, v01, v10, v11;
        if (SPLIT) {
            const uint16_t* in = (const uint16_t*)in_;
            v00 = ld_split4(in + o00, in_plane);
            v01 = ld_split4(in + o01, in_plane);
            v10 = ld_split4(in + o10, in_plane);
            v11 = ld_split4(in + o11, in_plane);
        } else {
            const float* in = (const float*)in_;
            v00 = *(const f32x4*)(in + o00);
            v01 = *(const f32x4*)(in + o01);
            v10 = *(const f32x4*)(in + o10);
            v11 = *(const f32x4*)(in + o11);
        }
        f32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = ly0 * (lx0 * v00[j] + lx1 * v01[j]) + ly1 * (lx0 * v10[j] + lx1 * v11[j]);
        if (SPLIT)
            st_split4((uint16_t*)out_ + pix * C + c4 * 4, out_plane, r);
        else
            *(f32x4*)((float*)out_ + pix * C + c4 * 4) = r;
    }
}


// split-format upsample, 8 channels (16 B per plane) per thread: half the memory instructions of the 4-channel form
__device__ __forceinline__ void ld_split8(const uint16_t* hi_ptr, size_t plane, float (&v)[8]) {
    const u32x4 ph = *(const u32x4*)hi_ptr;
    const u32x4 pl = *(const u32x4*)(hi_ptr + plane);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[2 * j] = __uint_as_float(ph[j] << 16) + __uint_as_float(pl[j] << 16);
        v[2 * j + 1] = __uint_as_float(ph[j] & 0xffff0000u) + __uint_as_float(pl[j] & 0xffff0000u);
    }
}
__global__ __launch_bounds__(256) void upsample_split8_kernel(const uint16_t* __restrict__ in, int B, int H, int W, int C,
                                                              uint16_t* __restrict__ out, int Ho, int Wo, float sy, float sx) {
    const size_t in_plane = (size_t)B * H * W * C, out_plane = (size_t)B * Ho * Wo * C;
    const int cq = C >> 3;
    const size_t total = (size_t)B * Ho * Wo * cq;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(t % cq);
        const size_t pix = t / cq;
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((size_t)Wo * Ho));
        const float fy = sy * oy, fx = sx * ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly1 = fy - y0, lx1 = fx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const size_t boff = (size_t)b * H * W * C + c8 * 8;
        float v00[8], v01[8], v10[8], v11[8];
        ld_split8(in + boff + ((size_t)y0 * W + x0) * C, in_plane, v00);
        ld_split8(in + boff + ((size_t)y0 * W + x1) * C, in_plane, v01);
        ld_split8(in + boff + ((size_t)y1 * W + x0) * C, in_plane, v10);
        ld_split8(in + boff + ((size_t)y1 * W + x1) * C, in_plane, v11);
        unsigned ph[4], pl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float r[2], h[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int k = 2 * j + e;
                r[e] = ly0 * (lx0 * v00[k] + lx1 * v01[k]) + ly1 * (lx0 * v10[k] + lx1 * v11[k]);
                h[e] = bf16_to_f32(f32_to_bf16(r[e]));
            }
            ph[j] = pack_bf16x2(h[0], h[1]);
            pl[j] = pack_bf16x2(r[0] - h[0], r[1] - h[1]);
        }
        uint16_t* o = out + pix * C + c8 * 8;
        *(u32x4*)o = u32x4{ph[0], ph[1], ph[2], ph[3]};
        *(u32x4*)(o + out_plane) = u32x4{pl[0], pl[1], pl[2], pl[3]};
    }
}

// Tiled form of the split-format upsample for ratios <= ~0.57 (the DPT x2 steps and the 296 -> 518 step): a block owns
// an 8 x 32 output tile x 64 channels.  Stage A unpacks the <= 8 x 20 source pixels the tile interpolates between into
// an fp32 LDS patch (each source value is loaded ONCE per tile: the one-thread-per-output form loads 4 corners x 2
// planes per output, ~9.5x more load instructions, and ran at 3.0 TB/s against a 6.2 TB/s streaming-write rate);
// stage B evaluates exactly the same expression as upsample_split8_kernel from the patch (bit-identical).
constexpr int UT_Y = 8, UT_X = 32, UT_C = 64, UP_R = 8, UP_C = 20;
__global__ __launch_bounds__(256) void upsample_split_tiled_kernel(const uint16_t* __restrict__ in, int B, int H, int W, int C,
                                                                   uint16_t* __restrict__ out, int Ho, int Wo, float sy, float sx) {
    __shared__ __attribute__((aligned(16))) float patch[UP_R * UP_C * UT_C];  // 40 KiB
    const size_t in_plane = (size_t)B * H * W * C, out_plane = (size_t)B * Ho * Wo * C;
    const int ntx = (Wo + UT_X - 1) / UT_X, nty = (Ho + UT_Y - 1) / UT_Y, ncc = C / UT_C;
    int bid = blockIdx.x;
    const int cc = bid % ncc;
    bid /= ncc;
    const int tx = bid % ntx;
    bid /= ntx;
    const int ty = bid % nty, b = bid / nty;
    const int oy0 = ty * UT_Y, ox0 = tx * UT_X;
    const int ybase = (int)(sy * oy0), xbase = (int)(sx * ox0);
    const int nr = min(H - 1, (int)(sy * min(oy0 + UT_Y - 1, Ho - 1)) + 1) - ybase + 1;  // <= UP_R (host check)
    const int nc = min(W - 1, (int)(sx * min(ox0 + UT_X - 1, Wo - 1)) + 1) - xbase + 1;  // <= UP_C
    const uint16_t* src = in + (size_t)b * H * W * C + cc * UT_C;
    for (int u = threadIdx.x; u < nr * nc * 8; u += 256) {
        const int c8 = u & 7, pc = (u >> 3) % nc, pr = (u >> 3) / nc;
        float v[8];
        ld_split8(src + ((size_t)(ybase + pr) * W + (xbase + pc)) * C + c8 * 8, in_plane, v);
        float* d = patch + (pr * UP_C + pc) * UT_C + c8 * 8;
        *(f32x4*)d = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)(d + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
    __syncthreads();
    for (int u = threadIdx.x; u < UT_Y * UT_X * 8; u += 256) {
        const int c8 = u & 7, px = (u >> 3) % UT_X, py = (u >> 3) / UT_X;
        const int oy = oy0 + py, ox = ox0 + px;
        if (oy >= Ho || ox >= Wo) continue;
        const float fy = sy * oy, fx = sx * ox;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly1 = fy - y0, lx1 = fx - x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float* p00 = patch + ((y0 - ybase) * UP_C + (x0 - xbase)) * UT_C + c8 * 8;
        const float* p01 = patch + ((y0 - ybase) * UP_C + (x1 - xbase)) * UT_C + c8 * 8;
        const float* p10 = patch + ((y1 - ybase) * UP_C + (x0 - xbase)) * UT_C + c8 * 8;
        const float* p11 = patch + ((y1 - ybase) * UP_C + (x1 - xbase)) * UT_C + c8 * 8;
        unsigned ph[4], pl[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const f32x4 v00 = *(const f32x4*)(p00 + 4 * j), v01 = *(const f32x4*)(p01 + 4 * j);
            const f32x4 v10 = *(const f32x4*)(p10 + 4 * j), v11 = *(const f32x4*)(p11 + 4 * j);
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                float r[2], h[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int k = 2 * e2 + e;
                    r[e] = ly0 * (lx0 * v00[k] + lx1 * v01[k]) + ly1 * (lx0 * v10[k] + lx1 * v11[k]);
                    h[e] = bf16_to_f32(f32_to_bf16(r[e]));
                }
                ph[2 * j + e2] = pack_bf16x2(h[0], h[1]);
                pl[2 * j + e2] = pack_bf16x2(r[0] - h[0], r[1] - h[1]);
            }
        }
        uint16_t* o = out + (((size_t)b * Ho + oy) * Wo + ox) * C + cc * UT_C + c8 * 8;
        *(u32x4*)o = u32x4{ph[0], ph[1], ph[2], ph[3]};
        *(u32x4*)(o + out_plane) = u32x4{pl[0], pl[1], pl[2], pl[3]};
    }
}

constexpr int TAIL_MAX = 8;  // output channels of a head tail (flow 2; mask 1 + covariance 3 + keypoint confidence 1 = 5)
struct TailArgs {
    int kind[TAIL_MAX];
    float a[TAIL_MAX];
    float d[TAIL_MAX];
};

// one thread per pixel; x row is Cin floats (Cin % 4 == 0, <= 64); w/b from global (L1/L2 resident).
template <int SPLIT, int CT>
__global__ __launch_bounds__(256) void head_tail_kernel(const void* __restrict__ x_, int P, int HW, int Cin,
                                                        const float* __restrict__ w, const float* __restrict__ bias,
                                                        int Cout, TailArgs ta, float* __restrict__ out,
                                                        float* __restrict__ out_logits) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        float acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = 0.f;
        for (int k = 0; k < Cin; k += 4) {
            const f32x4 xv = SPLIT ? ld_split4((const uint16_t*)x_ + (size_t)p * Cin + k, (size_t)P * Cin)
                                   : *(const f32x4*)((const float*)x_ + (size_t)p * Cin + k);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if (c < Cout) {
                    const f32x4 wv = *(const f32x4*)(w + c * Cin + k);
                    acc[c] += xv[0] * wv[0];
                    acc[c] += xv[1] * wv[1];
                    acc[c] += xv[2] * wv[2];
                    acc[c] += xv[3] * wv[3];
                }
            }
        }
        const int b = p / HW, q = p - b * HW;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (c < Cout) {
                const float y = acc[c] + bias[c];
                const size_t o = ((size_t)b * Cout + c) * HW + q;
                if (ta.kind[c] == 1) {
                    out[o] = 1.0f / (1.0f + expf(-y));
                    if (out_logits) out_logits[o] = y;
                } else {
                    out[o] = y * ta.a[c] + ta.d[c];
                }
            }
        }
    }
}

struct UnmapArgs {
    int rep0[4], src0[4], src1[4];
    float sscale[2], tscale[2];  // (x, y) -- float32 ratios, flow_resizing.py:832-853
    float cscale[4];
};

// torch bilinear (align_corners=False, no antialias) source index + weights for one axis
__device__ __forceinline__ void lin_coef(int dst, float scale, int in_size, int& i0, int& i1, float& l0, float& l1) {
    float src = scale * (dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = src - i0;
    l0 = 1.f - l1;
}

__global__ __launch_bounds__(256) void unmap_flow_kernel(const float* __restrict__ flow, int B, int h, int w,
                                                         UnmapArgs ua, int H0, int W0, float* __restrict__ out,
                                                         uint8_t* __restrict__ valid) {
    const int rh = ua.rep0[1] - ua.rep0[0], rw = ua.rep0[3] - ua.rep0[2];
    const int sh = ua.src0[1] - ua.src0[0], sw = ua.src0[3] - ua.src0[2];
    const float by = (float)rh / sh, bx = (float)rw / sw;  // bilinear + legacy-nearest scale = in/out
    const size_t total = (size_t)B * H0 * W0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(t % W0), y = (int)((t / W0) % H0), b = (int)(t / ((size_t)W0 * H0));
        float fx = 0.f, fy = 0.f;
        const bool in = y >= ua.src0[0] && y < ua.src0[1] && x >= ua.src0[2] && x < ua.src0[3];
        if (in) {
            const int yy = y - ua.src0[0], xx = x - ua.src0[2];
            int iy0, iy1, ix0, ix1;
            float ly0, ly1, lx0, lx1;
            lin_coef(yy, by, rh, iy0, iy1, ly0, ly1);
            lin_coef(xx, bx, rw, ix0, ix1, lx0, lx1);
            // pixel-centre grid (flow_resizing.py:788-800) interpolated like F.interpolate does it (2-D lerp)
            const float gx0 = ix0 + 0.5f, gx1 = ix1 + 0.5f, gy0 = iy0 + 0.5f, gy1 = iy1 + 0.5f;
            float sx = ly0 * (lx0 * gx0 + lx1 * gx1) + ly1 * (lx0 * gx0 + lx1 * gx1);
            float sy = ly0 * (lx0 * gy0 + lx1 * gy0) + ly1 * (lx0 * gy1 + lx1 * gy1);
            // legacy nearest: src = min(floor(dst * scale), in - 1)
            const int ny = min((int)floorf(yy * by), rh - 1), nx = min((int)floorf(xx * bx), rw - 1);
            const size_t fo = ((size_t)b * 2 * h + (ua.rep0[0] + ny)) * w + ua.rep0[2] + nx;
            float tx = flow[fo] + sx, ty = flow[fo + (size_t)h * w] + sy;
            sx = sx * ua.sscale[0];
            sy = sy * ua.sscale[1];
            tx = tx * ua.tscale[0];
            ty = ty * ua.tscale[1];
            sx += (float)ua.src0[2];
            sy += (float)ua.src0[0];
            tx += (float)ua.src1[2];
            ty += (float)ua.src1[0];
            fx = tx - sx;
            fy = ty - sy;
        }
        const size_t o = ((size_t)b * 2 * H0 + y) * W0 + x;
        out[o] = fx;
        out[o + (size_t)H0 * W0] = fy;
        if (valid) valid[t] = in ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void unmap_channels_kernel(const float* __restrict__ chan, int B, int C, int h, int w,
                                                             UnmapArgs ua, int H0, int W0, int use_scale,
                                                             float* __restrict__ out, uint8_t* __restrict__ valid) {
    const int rh = ua.rep0[1] - ua.rep0[0], rw = ua.rep0[3] - ua.rep0[2];
    const int sh = ua.src0[1] - ua.src0[0], sw = ua.src0[3] - ua.src0[2];
    const float by = (float)rh / sh, bx = (float)rw / sw;
    const size_t total = (size_t)B * H0 * W0;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(t % W0), y = (int)((t / W0) % H0), b = (int)(t / ((size_t)W0 * H0));
        const bool in = y >= ua.src0[0] && y < ua.src0[1] && x >= ua.src0[2] && x < ua.src0[3];
        int ny = 0, nx = 0;
        if (in) {
            ny = min((int)floorf((y - ua.src0[0]) * by), rh - 1);
            nx = min((int)floorf((x - ua.src0[2]) * bx), rw - 1);
        }
        for (int c = 0; c < C; ++c) {
            float v = 0.f;
            if (in) {
                v = chan[(((size_t)b * C + c) * h + ua.rep0[0] + ny) * w + ua.rep0[2] + nx];
                if (use_scale) v *= ua.cscale[c];
            }
            out[(((size_t)b * C + c) * H0 + y) * W0 + x] = v;
        }
        if (valid) valid[t] = in ? 1 : 0;
    }
}

// x [B*gh*gw][C*p*p] (col = (c,i,j)) -> out planar [B][C][gh*p][gw*p].  SPLIT: x is the UFM_BF16X2 pair of planes (value = hi + lo)
template <int SPLIT>
__global__ __launch_bounds__(256) void pixel_shuffle_kernel(const void* __restrict__ xv, int B, int gh, int gw, int C,
                                                            int p, float* __restrict__ out) {
    const int Ho = gh * p, Wo = gw * p;
    const size_t total = (size_t)B * C * Ho * Wo;
    const size_t plane = (size_t)B * gh * gw * C * p * p;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(t % Wo), oy = (int)((t / Wo) % Ho);
        const int c = (int)((t / ((size_t)Wo * Ho)) % C), b = (int)(t / ((size_t)Wo * Ho * C));
        const int gy = oy / p, i = oy - gy * p, gx = ox / p, j = ox - gx * p;
        const size_t src = (((size_t)b * gh + gy) * gw + gx) * ((size_t)C * p * p) + (c * p + i) * p + j;
        if (SPLIT) {
            const uint16_t* x = (const uint16_t*)xv;
            out[t] = bf16_to_f32(x[src]) + bf16_to_f32(x[plane + src]);
        } else {
            out[t] = ((const float*)xv)[src];
        }
    }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ in, uint16_t* __restrict__ out, size_t n4) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = *(const f32x4*)(in + t * 4);
        u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *(u32x2*)(out + t * 4) = pk;
    }
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, size_t n4) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (size_t)gridDim.x * blockDim.x) {
        const f32x4 va = *(const f32x4*)(a + t * 4), vb = *(const f32x4*)(b + t * 4);
        *(f32x4*)(out + t * 4) = va + vb;
    }
}

// out[orow(r)][:] = a[r][:] + tab[r % tab_mod][:]  (fp32 token assembly for the "parity" numerics mode)
__global__ __launch_bounds__(256) void add_rows_kernel(const float* __restrict__ a, int lda, const float* __restrict__ tab,
                                                       int ldtab, int tab_mod, float* __restrict__ out, int ldo,
                                                       int out_row_group, int rows, int D) {
    const int dq = D >> 2;
    const size_t total = (size_t)rows * dq;
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % dq);
        const int r = (int)(t / dq);
        const int orow = out_row_group > 0 ? (r / out_row_group) * (out_row_group + 1) + 1 + r % out_row_group : r;
        f32x4 v = *(const f32x4*)(a + (size_t)r * lda + c * 4);
        if (tab) v += *(const f32x4*)(tab + (size_t)(tab_mod > 0 ? r % tab_mod : r) * ldtab + c * 4);
        *(f32x4*)(out + (size_t)orow * ldo + c * 4) = v;
    }
}

inline dim3 stream_grid(size_t work_items) {
    size_t blocks = (work_items + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;  // Guideline 11: cap and grid-stride
    if (blocks < 1) blocks = 1;
    return dim3((unsigned)blocks);
}

}  // namespace

extern "C" int ufm_patchify(const void* img, int in_dtype, int in_layout, int B, int H, int W, int patch,
                            const float* scale3, const float* shift3, void* out, int out_dtype, int kpad,
                            void* stream) {
    UFM_REQUIRE(img && out && scale3 && shift3, "ufm_patchify: null pointer");
    UFM_REQUIRE(B > 0 && patch > 0 && H % patch == 0 && W % patch == 0, "ufm_patchify: H=%d W=%d not multiples of patch=%d", H, W, patch);
    UFM_REQUIRE(kpad % 4 == 0 && kpad >= 3 * patch * patch, "ufm_patchify: kpad=%d too small / not multiple of 4", kpad);
    UFM_REQUIRE((in_dtype == 0 || in_dtype == 1) && (in_layout == 0 || in_layout == 1), "ufm_patchify: bad dtype/layout");
    Affine3 af;
    for (int c = 0; c < 3; ++c) {
        af.scale[c] = scale3[c];
        af.shift[c] = shift3[c];
    }
    const size_t total = (size_t)B * (H / patch) * (W / patch) * (kpad / 4);
    if (out_dtype == UFM_BF16)
        hipLaunchKernelGGL(patchify_kernel<1>, stream_grid(total), dim3(256), 0, (hipStream_t)stream, img, in_dtype, in_layout, B, H, W, patch, af, out, kpad);
    else
        hipLaunchKernelGGL(patchify_kernel<0>, stream_grid(total), dim3(256), 0, (hipStream_t)stream, img, in_dtype, in_layout, B, H, W, patch, af, out, kpad);
    UFM_CHECK_LAUNCH("ufm_patchify");
    return UFM_OK;
}

static int g_upsample_tiled = 1;  // A/B and test hook below: bit 0 clear = one-thread-per-output kernels only
static int g_upsample_flags = 1;  // bit 1: ufm_dpt_tail_fused with the plain (2-way bank-conflicting) T image of rounds 1-4
int ufm_upsample_variant_flags() { return g_upsample_flags; }
extern "C" int ufm_debug_set_upsample_variant(int tiled) {
    UFM_REQUIRE((tiled & ~3) == 0, "ufm_debug_set_upsample_variant: %d has bits outside 0..1", tiled);
    g_upsample_flags = tiled;
    g_upsample_tiled = tiled & 1;
    return UFM_OK;
}

extern "C" int ufm_upsample_bilinear_nhwc(const void* in, int dtype, int B, int H, int W, int C, void* out, int Ho, int Wo,
                                          int crop_h, int crop_w, void* stream) {
    UFM_REQUIRE(dtype == UFM_F32 || dtype == UFM_BF16X2, "ufm_upsample_bilinear_nhwc: dtype must be UFM_F32 or UFM_BF16X2");
    UFM_REQUIRE(in && out, "ufm_upsample_bilinear_nhwc: null pointer");
    UFM_REQUIRE(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C % 4 == 0, "ufm_upsample_bilinear_nhwc: bad shape");
    UFM_REQUIRE(crop_h >= 0 && crop_h <= Ho && crop_w >= 0 && crop_w <= Wo, "ufm_upsample_bilinear_nhwc: bad crop");
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const int Hs = crop_h > 0 ? crop_h : Ho, Ws = crop_w > 0 ? crop_w : Wo;
    const size_t total = (size_t)B * Hs * Ws * (C / 4);
    if (dtype == UFM_BF16X2 && C % UT_C == 0 && g_upsample_tiled && sy * (UT_Y - 1) + 2.f <= (float)UP_R && sx * (UT_X - 1) + 2.f <= (float)UP_C &&
        (long long)B * ((Hs + UT_Y - 1) / UT_Y) * ((Ws + UT_X - 1) / UT_X) * (C / UT_C) < (1ll << 31)) {
        const unsigned blocks = (unsigned)((long long)B * ((Hs + UT_Y - 1) / UT_Y) * ((Ws + UT_X - 1) / UT_X) * (C / UT_C));
        hipLaunchKernelGGL(upsample_split_tiled_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in, B, H, W, C, (uint16_t*)out, Hs, Ws, sy, sx);
    } else if (dtype == UFM_BF16X2 && C % 8 == 0)
        hipLaunchKernelGGL(upsample_split8_kernel, stream_grid((size_t)B * Hs * Ws * (C / 8)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)in, B, H, W, C, (uint16_t*)out, Hs, Ws, sy, sx);
    else if (dtype == UFM_BF16X2)
        hipLaunchKernelGGL(upsample_kernel<1>, stream_grid(total), dim3(256), 0, (hipStream_t)stream, in, B, H, W, C, out, Hs, Ws, sy, sx);
    else
        hipLaunchKernelGGL(upsample_kernel<0>, stream_grid(total), dim3(256), 0, (hipStream_t)stream, in, B, H, W, C, out, Hs, Ws, sy, sx);
    UFM_CHECK_LAUNCH("ufm_upsample_bilinear_nhwc");
    return UFM_OK;
}

extern "C" int ufm_head_tail(const void* x, int in_dtype, int P, int HW, int Cin, const float* w, const float* b, int Cout,
                             const int32_t* kind_host, const float* a_host, const float* d_host, float* out,
                             float* out_logits, void* stream) {
    UFM_REQUIRE(x && w && b && out && kind_host && a_host && d_host, "ufm_head_tail: null pointer");
    UFM_REQUIRE(Cout >= 1 && Cout <= TAIL_MAX && Cin % 4 == 0 && Cin > 0 && P > 0 && HW > 0 && P % HW == 0, "ufm_head_tail: bad shape");
    TailArgs ta;
    for (int c = 0; c < TAIL_MAX; ++c) {
        ta.kind[c] = c < Cout ? kind_host[c] : 0;
        ta.a[c] = c < Cout ? a_host[c] : 1.f;
        ta.d[c] = c < Cout ? d_host[c] : 0.f;
    }
    UFM_REQUIRE(in_dtype == UFM_F32 || in_dtype == UFM_BF16X2, "ufm_head_tail: in_dtype must be UFM_F32 or UFM_BF16X2");
    if (in_dtype == UFM_BF16X2 && Cout <= 4)
        hipLaunchKernelGGL((head_tail_kernel<1, 4>), stream_grid((size_t)P), dim3(256), 0, (hipStream_t)stream, x, P, HW, Cin, w, b, Cout, ta, out, out_logits);
    else if (in_dtype == UFM_BF16X2)
        hipLaunchKernelGGL((head_tail_kernel<1, TAIL_MAX>), stream_grid((size_t)P), dim3(256), 0, (hipStream_t)stream, x, P, HW, Cin, w, b, Cout, ta, out, out_logits);
    else if (Cout <= 4)
        hipLaunchKernelGGL((head_tail_kernel<0, 4>), stream_grid((size_t)P), dim3(256), 0, (hipStream_t)stream, x, P, HW, Cin, w, b, Cout, ta, out, out_logits);
    else
        hipLaunchKernelGGL((head_tail_kernel<0, TAIL_MAX>), stream_grid((size_t)P), dim3(256), 0, (hipStream_t)stream, x, P, HW, Cin, w, b, Cout, ta, out, out_logits);
    UFM_CHECK_LAUNCH("ufm_head_tail");
    return UFM_OK;
}

// ---- output adaptors that are not a per-channel affine / sigmoid ([U] uniception prediction_heads.adaptors;
// call sites models/ufm.py:648-654).  Inputs are the raw decoded channels (ufm_head_tail kind 0 with a = 1, d = 0). ----
namespace {
__global__ __launch_bounds__(256) void covariance2d_kernel(const float* __restrict__ raw, size_t n, size_t HW, float* __restrict__ cov,
                                                           float* __restrict__ inv, float* __restrict__ logdet) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / HW, q = i - b * HW, o = b * 3 * HW + q;
        const float sx = expf(raw[o]), sy = expf(raw[o + HW]), rho = tanhf(raw[o + 2 * HW]) * 0.99f;
        const float cxx = sx * sx, cyy = sy * sy, cxy = rho * sx * sy;
        const float det = cxx * cyy - cxy * cxy;
        cov[o] = cxx, cov[o + HW] = cyy, cov[o + 2 * HW] = cxy;
        inv[o] = cyy / det, inv[o + HW] = cxx / det, inv[o + 2 * HW] = -cxy / det;
        logdet[b * HW + q] = logf(det);
    }
}
__global__ __launch_bounds__(256) void confidence_kernel(const float* __restrict__ raw, size_t n, int type, float vmin, float vmax,
                                                         float* __restrict__ out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = raw[i];
        float v = x;
        if (type == 0) v = fminf(vmin + expf(x), vmax);
        else if (type == 1) v = (vmax - vmin) * (1.0f / (1.0f + expf(-x))) + vmin;
        out[i] = v;
    }
}
}  // namespace

extern "C" int ufm_adaptor_covariance2d(const float* raw, int B, int HW, float* cov, float* inv_cov, float* log_det, void* stream) {
    UFM_REQUIRE(raw && cov && inv_cov && log_det && B > 0 && HW > 0, "ufm_adaptor_covariance2d: bad args");
    const size_t n = (size_t)B * HW;
    hipLaunchKernelGGL(covariance2d_kernel, stream_grid(n), dim3(256), 0, (hipStream_t)stream, raw, n, (size_t)HW, cov, inv_cov, log_det);
    UFM_CHECK_LAUNCH("ufm_adaptor_covariance2d");
    return UFM_OK;
}

extern "C" int ufm_adaptor_confidence(const float* raw, int64_t n, int type, float vmin, float vmax, float* out, void* stream) {
    UFM_REQUIRE(raw && out && n > 0 && type >= 0 && type <= 2, "ufm_adaptor_confidence: bad args");
    hipLaunchKernelGGL(confidence_kernel, stream_grid((size_t)n), dim3(256), 0, (hipStream_t)stream, raw, (size_t)n, type, vmin, vmax, out);
    UFM_CHECK_LAUNCH("ufm_adaptor_confidence");
    return UFM_OK;
}

static int fill_unmap(UnmapArgs& ua, const int32_t* rep0, const int32_t* src0, const int32_t* src1, int h, int w, int H0,
                      int W0) {
    for (int i = 0; i < 4; ++i) {
        ua.rep0[i] = rep0[i];
        ua.src0[i] = src0[i];
        ua.src1[i] = src1 ? src1[i] : src0[i];
    }
    if (!(rep0[0] >= 0 && rep0[1] <= h && rep0[0] < rep0[1] && rep0[2] >= 0 && rep0[3] <= w && rep0[2] < rep0[3])) return -1;
    if (!(src0[0] >= 0 && src0[1] <= H0 && src0[0] < src0[1] && src0[2] >= 0 && src0[3] <= W0 && src0[2] < src0[3])) return -1;
    const float rw = (float)(rep0[3] - rep0[2]), rh = (float)(rep0[1] - rep0[0]);
    ua.sscale[0] = (float)(ua.src0[3] - ua.src0[2]) / rw;
    ua.sscale[1] = (float)(ua.src0[1] - ua.src0[0]) / rh;
    ua.tscale[0] = (float)(ua.src1[3] - ua.src1[2]) / rw;
    ua.tscale[1] = (float)(ua.src1[1] - ua.src1[0]) / rh;
    for (int i = 0; i < 4; ++i) ua.cscale[i] = 1.f;
    return 0;
}

extern "C" int ufm_unmap_flow(const float* flow, int B, int h, int w, const int32_t* rep0, const int32_t* src0,
                              const int32_t* src1, int H0, int W0, float* out, uint8_t* valid, void* stream) {
    UFM_REQUIRE(flow && rep0 && src0 && src1 && out, "ufm_unmap_flow: null pointer");
    UFM_REQUIRE(B > 0 && h > 0 && w > 0 && H0 > 0 && W0 > 0, "ufm_unmap_flow: bad shape");
    UnmapArgs ua;
    UFM_REQUIRE(fill_unmap(ua, rep0, src0, src1, h, w, H0, W0) == 0, "ufm_unmap_flow: region outside its image");
    hipLaunchKernelGGL(unmap_flow_kernel, stream_grid((size_t)B * H0 * W0), dim3(256), 0, (hipStream_t)stream, flow, B, h, w, ua, H0, W0, out, valid);
    UFM_CHECK_LAUNCH("ufm_unmap_flow");
    return UFM_OK;
}

extern "C" int ufm_unmap_channels(const float* chan, int B, int C, int h, int w, const int32_t* rep0,
                                  const int32_t* src0, int H0, int W0, const float* chan_scale_host, float* out,
                                  uint8_t* valid, void* stream) {
    UFM_REQUIRE(chan && rep0 && src0 && out, "ufm_unmap_channels: null pointer");
    UFM_REQUIRE(B > 0 && C > 0 && C <= 4 && h > 0 && w > 0 && H0 > 0 && W0 > 0, "ufm_unmap_channels: bad shape (C<=4)");
    UnmapArgs ua;
    UFM_REQUIRE(fill_unmap(ua, rep0, src0, nullptr, h, w, H0, W0) == 0, "ufm_unmap_channels: region outside its image");
    if (chan_scale_host)
        for (int c = 0; c < C; ++c) ua.cscale[c] = chan_scale_host[c];
    hipLaunchKernelGGL(unmap_channels_kernel, stream_grid((size_t)B * H0 * W0), dim3(256), 0, (hipStream_t)stream, chan, B, C, h, w, ua, H0, W0, chan_scale_host ? 1 : 0, out, valid);
    UFM_CHECK_LAUNCH("ufm_unmap_channels");
    return UFM_OK;
}

extern "C" int ufm_pixel_shuffle_planar(const void* x, int in_dtype, int B, int gh, int gw, int C, int p, float* out, void* stream) {
    UFM_REQUIRE(x && out && B > 0 && gh > 0 && gw > 0 && C > 0 && p > 0, "ufm_pixel_shuffle_planar: bad args");
    UFM_REQUIRE(in_dtype == UFM_F32 || in_dtype == UFM_BF16X2, "ufm_pixel_shuffle_planar: bad in_dtype %d", in_dtype);
    if (in_dtype == UFM_BF16X2)
        hipLaunchKernelGGL(pixel_shuffle_kernel<1>, stream_grid((size_t)B * C * gh * p * gw * p), dim3(256), 0, (hipStream_t)stream, x, B, gh, gw, C, p, out);
    else
        hipLaunchKernelGGL(pixel_shuffle_kernel<0>, stream_grid((size_t)B * C * gh * p * gw * p), dim3(256), 0, (hipStream_t)stream, x, B, gh, gw, C, p, out);
    UFM_CHECK_LAUNCH("ufm_pixel_shuffle_planar");
    return UFM_OK;
}

extern "C" int ufm_cast_f32_to_bf16(const float* in, uint16_t* out, int64_t n, void* stream) {
    UFM_REQUIRE(in && out && n > 0 && n % 4 == 0, "ufm_cast_f32_to_bf16: n must be a positive multiple of 4");
    hipLaunchKernelGGL(cast_bf16_kernel, stream_grid((size_t)n / 4), dim3(256), 0, (hipStream_t)stream, in, out, (size_t)n / 4);
    UFM_CHECK_LAUNCH("ufm_cast_f32_to_bf16");
    return UFM_OK;
}

extern "C" int ufm_add_f32(const float* a, const float* b, float* out, int64_t n, void* stream) {
    UFM_REQUIRE(a && b && out && n > 0 && n % 4 == 0, "ufm_add_f32: n must be a positive multiple of 4");
    hipLaunchKernelGGL(add_kernel, stream_grid((size_t)n / 4), dim3(256), 0, (hipStream_t)stream, a, b, out, (size_t)n / 4);
    UFM_CHECK_LAUNCH("ufm_add_f32");
    return UFM_OK;
}

extern "C" int ufm_add_rows(const float* a, int lda, const float* tab, int ldtab, int tab_mod, float* out, int ldo,
                            int out_row_group, int rows, int D, void* stream) {
    UFM_REQUIRE(a && out && rows > 0 && D > 0 && D % 4 == 0 && lda % 4 == 0 && ldo % 4 == 0 && (!tab || ldtab % 4 == 0), "ufm_add_rows: bad args");
    hipLaunchKernelGGL(add_rows_kernel, stream_grid((size_t)rows * (D / 4)), dim3(256), 0, (hipStream_t)stream, a, lda, tab, ldtab, tab_mod, out, ldo, out_row_group, rows, D);
    UFM_CHECK_LAUNCH("ufm_add_rows");
    return UFM_OK;
}
