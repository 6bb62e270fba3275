// fp32 implicit-GEMM convolution on NHWC activations with exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// The reference keeps the DPT heads in fp32 (models/ufm.py:635 "the prediction need precision"),
// so this kernel computes in true fp32: gfx950 has no TF32/xf32, the f32-input MFMA is a k-ordered
// fmaf chain (cdna_hip_programming.md section 3 "FP32-input MFMA") at 157 TF peak.
//
// GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = KH*KW*Cin; the A operand is gathered on
// the fly (im2col never materialised): K-step kt covers 32 input channels of one filter tap.
//   * 128 x BN x 32 block tile (BN = 128 / 64 / 32 picked per layer), 4 waves, 32x32 MFMA tiles.
//   * both operands go global -> LDS by global_load_lds_dwordx4; the per-lane SOURCE address does
//     the gather, the halo (zero padding) is served from a caller-provided zero page, and the
//     16-byte chunk XOR ((row>>1)&7) swizzle is applied on the source side and on the ds_read_b128
//     fragment reads (conflict-free for the 32-row operand).
//   * one ds_read_b128 per operand row feeds FOUR k-steps of the 32x32x2 MFMA: lane (row, k-half h)
//     holds k = 8*step + 4*h + e for e = 0..3, used by MFMA e with the same assignment on both
//     operands (a permutation of the k sum).
//   * MFMA is issued as D[cout][pixel] so a lane's accumulator quads are 4 consecutive output
//     channels of one pixel: 16-byte epilogue accesses for bias / gamma / residuals / store.
//   * fused: ReLU on the input (ResidualConvUnit), bias, activation, LayerScale, two residual
//     adds (RCU skip + FeatureFusion skip), and a pixel-shuffle store for ConvTranspose(k == s).
#include "common.h"

namespace {

constexpr int BM = 128, BK = 32;

struct ConvArgs {
    const float* in;
    const float* w;
    const float* bias;
    const float* gamma;
    const float* res1;
    const float* res2;
    const float* zero;
    float* out;
    int B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, M;
    int relu_in, act, shuffle, Co;
    int replicate;  // padding_mode='replicate': out-of-range taps read the nearest edge pixel instead of zero
};

template <int BN>
__global__ __launch_bounds__(256, 2) void conv_f32_kernel(ConvArgs p) {
    constexpr int WN = BN >= 64 ? 2 : 1, WM = 4 / WN;
    constexpr int TM = (BM / WM) / 32, TN = (BN / WN) / 32;
    constexpr int A_BYTES = BM * BK * 4, STAGE_BYTES = (BM + BN) * BK * 4;
    constexpr int WPIECES = BN / 8 / 4;  // weight DMA pieces per wave (8 rows each)
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntn = p.Cout / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tmi = bid / ntn, tni = bid - tmi * ntn;
    const int m0 = tmi * BM, n0 = tni * BN;
    const int cpt = p.Cin / BK;  // K-steps per filter tap
    const int ntaps = p.KH * p.KW;
    const int nk = ntaps * cpt;
    const size_t ktot = (size_t)p.KH * p.KW * p.Cin;

    // ---- staging set-up: A pieces 4w..4w+3 (8 pixel rows each) ----
    const int srow = lane >> 3, slot = lane & 7;
    int a_iy0[4], a_ix0[4], a_chunk[4];
    size_t a_img[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + srow;
        a_chunk[i] = (slot ^ ((r >> 1) & 7)) * 4;
        const int m = min(m0 + r, p.M - 1);
        const int ox = m % p.Wo, t = m / p.Wo, oy = t % p.Ho, b = t / p.Ho;
        a_iy0[i] = oy * p.stride - p.pad;
        a_ix0[i] = ox * p.stride - p.pad;
        a_img[i] = (size_t)b * p.H * p.W;
    }
    const float* gw[WPIECES];
#pragma unroll
    for (int i = 0; i < WPIECES; ++i) {
        const int r = (wave * WPIECES + i) * 8 + srow;
        gw[i] = p.w + (size_t)(n0 + r) * ktot + (slot ^ ((r >> 1) & 7)) * 4;
    }
    auto stage = [&](int buf, int kt) {
        // channel-chunk outer, filter tap inner: the KH*KW taps of one 32-channel chunk are consecutive K-steps,
        // so the +-1-pixel-shifted re-reads of the same input lines hit L1/L2 (tap-major order re-fetched the
        // whole input 9x from beyond L2: 6.5 GB for the 296^2 layer, PMC FETCH_SIZE, profiles/r01)
        const int chunk = kt / ntaps, tap = kt - chunk * ntaps, c0 = chunk * BK;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const size_t koff = (size_t)tap * p.Cin + c0;  // weight layout stays [Cout][KH][KW][Cin]
        char* sa = smem + buf * STAGE_BYTES + wave * 4096;
        char* sb = smem + buf * STAGE_BYTES + A_BYTES + wave * (WPIECES * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
            if (p.replicate) iy = min(max(iy, 0), p.H - 1), ix = min(max(ix, 0), p.W - 1);
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const float* src = ok ? p.in + (a_img[i] + (size_t)iy * p.W + ix) * p.Cin + c0 + a_chunk[i] : p.zero + a_chunk[i];
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(sa + i * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WPIECES; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gw[i] + koff), LDS_PTR(sb + i * 1024), 16, 0, 0);
    };

    // ---- fragment offsets ----
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 31, fh = lane >> 5;
    int px_off[TM], w_off[TN], px_sw[TM], w_sw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * (TM * 32) + i * 32 + fr;
        px_off[i] = r * 128;
        px_sw[i] = (r >> 1) & 7;
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int r = wn * (TN * 32) + i * 32 + fr;
        w_off[i] = A_BYTES + r * 128;
        w_sw[i] = (r >> 1) & 7;
    }

    f32x16 acc[TN][TM];
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;

    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const char* s = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int chunk = 2 * st + fh;
            f32x4 pf[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                pf[i] = *(const f32x4*)(s + px_off[i] + ((chunk ^ px_sw[i]) << 4));
                if (p.relu_in) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) pf[i][e] = fmaxf(pf[i][e], 0.f);
                }
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) wf[i] = *(const f32x4*)(s + w_off[i] + ((chunk ^ w_sw[i]) << 4));
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int m = 0; m < TM; ++m)
                        acc[n][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[n][e], pf[m][e], acc[n][m], 0, 0, 0);
        }
    }

    // ---- epilogue ----
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        const int pix = m0 + wm * (TM * 32) + m * 32 + fr;
        if (pix >= p.M) continue;
        int sb = 0, sy = 0, sx = 0;
        if (p.shuffle) {
            sx = pix % p.Wo;
            const int t = pix / p.Wo;
            sy = t % p.Ho;
            sb = t / p.Ho;
        }
#pragma unroll
        for (int n = 0; n < TN; ++n) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cb = n0 + wn * (TN * 32) + n * 32 + 8 * g + 4 * fh;
                f32x4 v = {acc[n][m][4 * g], acc[n][m][4 * g + 1], acc[n][m][4 * g + 2], acc[n][m][4 * g + 3]};
                if (p.shuffle) {
                    const int tapo = cb / p.Co, co = cb - tapo * p.Co;
                    const int kh = tapo / p.shuffle, kw = tapo - kh * p.shuffle;
                    if (p.bias) v += *(const f32x4*)(p.bias + co);
                    const size_t o = (((size_t)sb * (p.Ho * p.shuffle) + sy * p.shuffle + kh) * (p.Wo * p.shuffle) + sx * p.shuffle + kw) * p.Co + co;
                    *(f32x4*)(p.out + o) = v;
                    continue;
                }
                if (p.bias) v += *(const f32x4*)(p.bias + cb);
                if (p.act != UFM_ACT_NONE) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], p.act);
                }
                if (p.gamma) v *= *(const f32x4*)(p.gamma + cb);
                const size_t o = (size_t)pix * p.Cout + cb;
                if (p.res1) v += *(const f32x4*)(p.res1 + o);
                if (p.res2) v += *(const f32x4*)(p.res2 + o);
                *(f32x4*)(p.out + o) = v;
            }
        }
    }
}

}  // namespace

extern "C" int ufm_conv2d_nhwc_f32(const float* in, int B, int H, int W, int Cin, const float* weight, int Cout, int KH,
                                   int KW, int stride, int pad, int relu_in, const float* bias, int act,
                                   const float* gamma, const float* res1, const float* res2, int shuffle, float* out,
                                   int out_dtype_unused, const float* zero_page, void* stream) {
    (void)out_dtype_unused;
    UFM_REQUIRE(in && weight && out && zero_page, "ufm_conv2d_nhwc_f32: null pointer");
    UFM_REQUIRE(B > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, "ufm_conv2d_nhwc_f32: bad geometry");
    UFM_REQUIRE(Cin % BK == 0 && Cin > 0, "ufm_conv2d_nhwc_f32: Cin=%d must be a multiple of %d", Cin, BK);
    UFM_REQUIRE(Cout % 32 == 0 && Cout > 0, "ufm_conv2d_nhwc_f32: Cout=%d must be a multiple of 32", Cout);
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    UFM_REQUIRE(Ho > 0 && Wo > 0, "ufm_conv2d_nhwc_f32: empty output");
    int Co = Cout;
    if (shuffle) {
        UFM_REQUIRE(KH == 1 && KW == 1 && stride == 1 && pad == 0, "ufm_conv2d_nhwc_f32: shuffle mode needs a 1x1 geometry");
        UFM_REQUIRE(Cout % (shuffle * shuffle) == 0, "ufm_conv2d_nhwc_f32: Cout not divisible by shuffle^2");
        Co = Cout / (shuffle * shuffle);
        UFM_REQUIRE(Co % 4 == 0, "ufm_conv2d_nhwc_f32: Co=%d must be a multiple of 4 in shuffle mode", Co);
        UFM_REQUIRE(!res1 && !res2 && !gamma && act == UFM_ACT_NONE, "ufm_conv2d_nhwc_f32: shuffle mode supports bias only");
    }
    const long long M = (long long)B * Ho * Wo;
    UFM_REQUIRE(M < (1ll << 31) && (long long)B * H * W * Cin < (1ll << 40), "ufm_conv2d_nhwc_f32: problem too large");
    ConvArgs p{in, weight, bias, gamma, res1, res2, zero_page, out, B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, (int)M, relu_in & 1, act, shuffle, Co, (relu_in >> 1) & 1};
    const int ntm = (int)((M + BM - 1) / BM);
    if (Cout % 128 == 0) {
        hipLaunchKernelGGL(conv_f32_kernel<128>, dim3(ntm * (Cout / 128)), dim3(256), 0, (hipStream_t)stream, p);
    } else if (Cout % 64 == 0) {
        hipLaunchKernelGGL(conv_f32_kernel<64>, dim3(ntm * (Cout / 64)), dim3(256), 0, (hipStream_t)stream, p);
    } else {
        hipLaunchKernelGGL(conv_f32_kernel<32>, dim3(ntm * (Cout / 32)), dim3(256), 0, (hipStream_t)stream, p);
    }
    UFM_CHECK_LAUNCH("ufm_conv2d_nhwc_f32");
    return UFM_OK;
}
