// Fused full-resolution tail of the DPT regression processor ("fast" numerics):
//     bilinear(align_corners) h x w -> H x W   ([U] DPTRegressionProcessor: interpolate to the target size)
//  -> conv 3x3 pad 1, 128 -> 32, bias, ReLU      (conv2[0], conv2[1])
//  -> conv 1x1, 32 -> Ct <= 8, bias              (conv2[2])
//  -> FlowAdaptor / MaskAdaptor                  (models/ufm.py:644-660)
// in ONE kernel.  Unfused (ufm_upsample_bilinear_nhwc -> ufm_conv2d_nhwc_bf16x3 -> ufm_head_tail) the 128-channel
// full-resolution map is written once (1.1 GB at B = 8, 518^2) and re-read 9x by the implicit GEMM's tap gathers
// (that layer was L2-bound: 861 us), plus a 32-channel map written and re-read.  Here neither map ever leaves the CU.
//
// One block = one 16 x 16 output tile of one image, 4 waves, each wave 4 tile rows (= 4 M-fragments of 16 pixels) x
// 32 output channels.  The 128 input channels are processed in four passes of 32 (77.5 KiB of LDS: two blocks per
// CU, so one block's fill (VALU + loads) overlaps the other's MFMA pass):
//   fill : the 18 x 18 halo tile of the UPSAMPLED map is computed straight from the h x w source (same fp32
//          operations in the same order as upsample_split8_kernel, evaluated separably: horizontal lerps of the
//          <= 13 source rows first), split into (hi, lo) bf16 and parked in LDS: [324 px][hi 64 B | lo 64 B],
//          16-byte chunk index XOR (px & 7) (conflict-free ds_read_b128 for 16 consecutive pixels); zero outside the
//          image (= the convolution's padding);
//   mma  : K order = 32-channel chunk outer, filter tap inner (conv_bf16x3.hip's order); the A fragment of a tap is
//          the same LDS tile read at a shifted pixel index -- no im2col re-fetch; the pass's 36 KiB of weights are
//          loaded while the fill runs and parked in LDS for the four waves (no barrier inside the 9-tap loop);
//          3 MFMA per (hi, lo) product as in the bf16x3 convolution.
// The epilogue reproduces the unfused arithmetic exactly (bias, ReLU, the (hi, lo) round trip of the 32-channel map,
// the sequential fp32 dot of ufm_head_tail, the adaptor), so fused and unfused outputs are BIT-IDENTICAL (tested).
#include "common.h"

int ufm_upsample_variant_flags();  // pointwise.hip: ufm_debug_set_upsample_variant

namespace {

constexpr int TS = 16, HS = TS + 2, NPX = HS * HS;  // 16 x 16 tile, 18 x 18 halo
constexpr int CIN = 128, CMID = 32, CQ = 32;         // channels per pass (= one K-chunk of the bf16x3 convolution)
constexpr int HALO_B = NPX * 128;                    // per pixel 128 B: [hi: 32 ch | lo: 32 ch] -> 41,472 B
constexpr int WQ_B = 9 * 2 * CMID * 64;              // the 3x3 weights of one 32-channel pass: [tap][plane][32 cout][64 B] = 36 KiB
constexpr int SMEM_B = HALO_B + WQ_B;                // 77.5 KiB: two blocks per CU
constexpr int TROWS = 13;                            // source rows of the separable fill: T[13][18][32] fp32 = 29,952 B <= WQ_B

__device__ __forceinline__ int swz64(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }  // 64-B rows (conv_bf16x3.hip)

struct TailFusedArgs {
    const uint16_t* in;   // [2][B][h][w][128] split planes (p_conv1 output)
    const uint16_t* w2;   // [2][32][3][3][128] split planes
    const float* b2;      // [32]
    const float* wt;      // [Ct][32]
    const float* bt;      // [Ct]
    float* out;           // [B][Ct][H][W]
    float* out_logits;    // optional, same shape
    long long in_plane, w_plane;
    int B, h, w, H, W, Ct;
    float sy, sx;
    int t_swap;  // 4: the bank-conflict-free T image (default); 0: the plain image of rounds 1-4 (A/B: ufm_debug_set_upsample_variant bit 1)
    int kind[8];
    float a[8], d[8];
};

__device__ __forceinline__ void ld_split8(const uint16_t* hi_ptr, long long plane, float (&v)[8]) {
    const u32x4 ph = *(const u32x4*)hi_ptr;
    const u32x4 pl = *(const u32x4*)(hi_ptr + plane);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[2 * j] = __uint_as_float(ph[j] << 16) + __uint_as_float(pl[j] << 16);
        v[2 * j + 1] = __uint_as_float(ph[j] & 0xffff0000u) + __uint_as_float(pl[j] & 0xffff0000u);
    }
}

template <int CT>  // compile-time bound on the tail's output channels (4: flow / mask heads; 8: mask + covariance + confidence)
__global__ __launch_bounds__(256, 2) void dpt_tail_fused_kernel(TailFusedArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[SMEM_B];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntx = (p.W + TS - 1) / TS, nty = (p.H + TS - 1) / TS;
    const int bid = blockIdx.x;
    const int b = bid / (ntx * nty), t = bid - b * (ntx * nty);
    const int ty0 = (t / ntx) * TS, tx0 = (t % ntx) * TS;
    const int fr = lane & 15, fq = lane >> 4;

    f32x4 acc[2][4];  // [n][m]: lane holds couts n*16 + fq*4 + j of pixel (row 4*wave + m, col fr)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    int base_p[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) base_p[m] = (4 * wave + m) * HS + fr;
    // W staging: piece i = it * 256 + tid of a pass = (tap it, plane, cout row, 16-byte chunk); every block re-reads the
    // same 147 KB from L2 -- through LDS once per block and pass instead of once per wave and tap (5 GB -> 1.25 GB)
    const int w_plane_i = tid >> 7, w_row = (tid >> 2) & 31, w_chunk = tid & 3;
    const uint16_t* w_src = p.w2 + (size_t)w_plane_i * p.w_plane + (size_t)w_row * (9 * CIN) + w_chunk * 8;
    char* w_dst = smem + HALO_B + w_plane_i * 2048 + w_row * 64 + ((w_chunk ^ swz64(w_row)) << 4);
    int w_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) w_off[j] = HALO_B + (j * 16 + fr) * 64 + ((fq ^ swz64(j * 16 + fr)) << 4);

    // source rows this tile's 18 halo rows interpolate between (host guarantees <= TROWS of them)
    const int ybase = (int)(p.sy * max(ty0 - 1, 0));
    const int ytop = min(p.h - 1, (int)(p.sy * min(ty0 + TS, p.H - 1)) + 1);
    const int nrows = ytop - ybase + 1;
    // Stage-A work items of this thread: item k = (source row r, halo column hx, 8-channel group c8) with index tid + 256 k
    // (nrows * 72 <= 936 of them: at most AK = 4 per thread).  Their geometry does not depend on the pass, only the channel
    // offset does, so it is computed once; the 16 global loads of a pass are issued TOGETHER and one pass AHEAD (right
    // before the previous pass's tap loop), so their latency runs under 216 MFMAs instead of being exposed once per item
    // (the un-pipelined loop waited vmcnt(0) behind every item's four loads: ~16 exposed L2/HBM latencies per block).
    constexpr int AK = (TROWS * HS * 4 + 255) / 256;
    size_t a_src0[AK], a_src1[AK];
    int a_dst[AK];
    float a_lx1[AK];
    bool a_ok[AK];
#pragma unroll
    for (int k = 0; k < AK; ++k) {
        const int u = tid + 256 * k;
        const int r = u / (HS * 4), rem = u - r * (HS * 4);
        const int hx = rem >> 2, c8 = rem & 3;
        const int ox = tx0 - 1 + hx;
        a_ok[k] = u < nrows * (HS * 4) && (unsigned)ox < (unsigned)p.W;  // columns outside the image are never read
        const float fx = p.sx * ox;
        const int x0 = a_ok[k] ? (int)fx : 0;
        const int x1 = x0 + (x0 < p.w - 1 ? 1 : 0);
        a_lx1[k] = fx - x0;
        const size_t row = ((size_t)b * p.h + min(ybase + r, p.h - 1)) * p.w;
        a_src0[k] = (row + x0) * CIN + c8 * 8;
        a_src1[k] = (row + x1) * CIN + c8 * 8;
        a_dst[k] = (r * HS + hx) * 32 + c8 * 8;
    }
    u32x4 ald[AK][4];  // [item][x0 hi, x0 lo, x1 hi, x1 lo]
    auto issue_a = [&](int q) {
#pragma unroll
        for (int k = 0; k < AK; ++k) {
            if (a_ok[k]) {
                const uint16_t* s0 = p.in + a_src0[k] + q * CQ;
                const uint16_t* s1 = p.in + a_src1[k] + q * CQ;
                ald[k][0] = *(const u32x4*)s0;
                ald[k][1] = *(const u32x4*)(s0 + p.in_plane);
                ald[k][2] = *(const u32x4*)s1;
                ald[k][3] = *(const u32x4*)(s1 + p.in_plane);
            }
        }
    };
    issue_a(0);
    for (int q = 0; q < CIN / CQ; ++q) {
        if (q) __syncthreads();  // every wave is done reading the previous pass's tile and weights
        // ---- fill, separable.  Stage A: T[r][hx] = lx0 * v[y][x0] + lx1 * v[y][x1] for the <= 13 source rows y the tile
        // touches and its 18 halo columns (fp32, parked in the weight region, which is idle until the fill is done).
        // Stage B: halo[hy][hx] = split(ly0 * T[r0][hx] + ly1 * T[r1][hx]).  Same operations in the same order as the
        // one-step form ly0 * (lx0 v00 + lx1 v01) + ly1 * (lx0 v10 + lx1 v11), so the result is bit-identical, with
        // 2.8x fewer global loads and ~1/3 less VALU work (each horizontal lerp is shared by ~1.75 output rows). ----
        float* T = (float*)(smem + HALO_B);
#pragma unroll
        for (int k = 0; k < AK; ++k) {
            if (!a_ok[k]) continue;
            float v0[8], v1[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // (= ld_split8)
                v0[2 * j] = __uint_as_float(ald[k][0][j] << 16) + __uint_as_float(ald[k][1][j] << 16);
                v0[2 * j + 1] = __uint_as_float(ald[k][0][j] & 0xffff0000u) + __uint_as_float(ald[k][1][j] & 0xffff0000u);
                v1[2 * j] = __uint_as_float(ald[k][2][j] << 16) + __uint_as_float(ald[k][3][j] << 16);
                v1[2 * j + 1] = __uint_as_float(ald[k][2][j] & 0xffff0000u) + __uint_as_float(ald[k][3][j] & 0xffff0000u);
            }
            const float lx1 = a_lx1[k], lx0 = 1.f - lx1;
            f32x4 t0, t1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t0[j] = lx0 * v0[j] + lx1 * v1[j];
                t1[j] = lx0 * v0[4 + j] + lx1 * v1[4 + j];
            }
            // T entry e = (r, hx) is 128 B = half the LDS's bank row: entries e and e + 2 start on the same bank.  Stage B's
            // ds_read_b128 lane groups take four consecutive entries (four lanes x 16 B at a 32-B stride each), so e / e + 2 met on the
            // same banks: the 2-way conflicts PMC counted (30 M cycles per launch, rounds 2-4).  The two 16-byte halves of every 32-B
            // pair are swapped in entries with bit 1 set: e + 2's lanes then fill the holes e's lanes leave.  Same values, same bits.
            float* dst = T + a_dst[k];
            const int tsw = ((a_dst[k] >> 6) & 1) ? p.t_swap : 0;  // (entry >> 1) & 1 -> 4 floats
            *(f32x4*)(dst + tsw) = t0;
            *(f32x4*)(dst + (4 ^ tsw)) = t1;
        }
        // this pass's weights: requested now, parked in LDS after stage B (their latency runs under stage B, which only
        // touches LDS); not earlier, so that they are never live beside the 64 registers of stage-A loads
        u32x4 wreg[9];
#pragma unroll
        for (int it = 0; it < 9; ++it) wreg[it] = *(const u32x4*)(w_src + it * CIN + q * CQ);
        __syncthreads();
        for (int u = tid; u < NPX * 4; u += 256) {
            const int px = u >> 2, c8 = u & 3;
            const int hy = px / HS, hx = px - hy * HS;
            const int oy = ty0 - 1 + hy, ox = tx0 - 1 + hx;
            u32x4 ph = {0u, 0u, 0u, 0u}, pl = {0u, 0u, 0u, 0u};
            if ((unsigned)oy < (unsigned)p.H && (unsigned)ox < (unsigned)p.W) {
                const float fy = p.sy * oy;
                const int y0 = (int)fy;
                const int y1 = y0 + (y0 < p.h - 1 ? 1 : 0);
                const float ly1 = fy - y0, ly0 = 1.f - ly1;
                const int ea = (y0 - ybase) * HS + hx, ec = (y1 - ybase) * HS + hx;
                const float* a = T + ea * 32 + c8 * 8;
                const float* c = T + ec * 32 + c8 * 8;
                const int sa = ((ea >> 1) & 1) ? p.t_swap : 0, sc = ((ec >> 1) & 1) ? p.t_swap : 0;  // stage A's half swap
                const f32x4 a0 = *(const f32x4*)(a + sa), a1 = *(const f32x4*)(a + (4 ^ sa)), c0 = *(const f32x4*)(c + sc), c1 = *(const f32x4*)(c + (4 ^ sc));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float r[2], hh[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int k = 2 * j + e;
                        r[e] = ly0 * (k < 4 ? a0[k & 3] : a1[k & 3]) + ly1 * (k < 4 ? c0[k & 3] : c1[k & 3]);
                        hh[e] = bf16_to_f32(f32_to_bf16(r[e]));
                    }
                    ph[j] = pack_bf16x2(hh[0], hh[1]);
                    pl[j] = pack_bf16x2(r[0] - hh[0], r[1] - hh[1]);
                }
            }
            char* dst = smem + px * 128;
            *(u32x4*)(dst + ((c8 ^ (px & 7)) << 4)) = ph;
            *(u32x4*)(dst + (((4 + c8) ^ (px & 7)) << 4)) = pl;
        }
        __syncthreads();  // T (in the weight region) has been consumed
#pragma unroll
        for (int it = 0; it < 9; ++it) *(u32x4*)(w_dst + it * 4096) = wreg[it];
        __syncthreads();
        if (q + 1 < CIN / CQ) issue_a(q + 1);  // the next pass's stage-A loads fly under this pass's tap loop
        // ---- mma: 9 taps of this 32-channel chunk ----
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - kh * 3;
            bf16x8 wh[2], wl[2], ah[4], al[4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                wh[j] = *(const bf16x8*)(smem + w_off[j] + tap * 4096);
                wl[j] = *(const bf16x8*)(smem + w_off[j] + tap * 4096 + 2048);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int pp = base_p[m] + kh * HS + kw;
                const char* s = smem + pp * 128;
                ah[m] = *(const bf16x8*)(s + ((fq ^ (pp & 7)) << 4));
                al[m] = *(const bf16x8*)(s + (((4 + fq) ^ (pp & 7)) << 4));
            }
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[n], ah[m], acc[n][m], 0, 0, 0);
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[n], al[m], acc[n][m], 0, 0, 0);
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[n], ah[m], acc[n][m], 0, 0, 0);
                }
        }
    }
    __syncthreads();  // the epilogue's staging slices alias the halo tile

    // ---- epilogue.  Park relu(acc + bias) -- after the (hi, lo) round trip the unfused 32-channel map goes through --
    // as [64 px][32 ch] fp32 in the wave's own LDS slice, then one lane per pixel does ufm_head_tail's sequential dot ----
    char* ws = smem + wave * (64 * CMID * 4);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const f32x4 bv = *(const f32x4*)(p.b2 + n * 16 + fq * 4);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int r = m * 16 + fr;
            f32x4 v = acc[n][m] + bv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = fmaxf(v[j], 0.0f);
                const float hi = bf16_to_f32(f32_to_bf16(y));
                v[j] = hi + bf16_to_f32(f32_to_bf16(y - hi));
            }
            *(f32x4*)(ws + r * 128 + (((n * 4 + fq) ^ (r & 7)) << 4)) = v;
        }
    }
    const int oy = ty0 + 4 * wave + (lane >> 4), ox = tx0 + (lane & 15);
    float o[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) o[c] = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < 8; ++k4) {
        const f32x4 xv = *(const f32x4*)(ws + lane * 128 + ((k4 ^ (lane & 7)) << 4));
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (c < p.Ct) {
                const f32x4 wv = *(const f32x4*)(p.wt + c * CMID + k4 * 4);
                o[c] += xv[0] * wv[0];
                o[c] += xv[1] * wv[1];
                o[c] += xv[2] * wv[2];
                o[c] += xv[3] * wv[3];
            }
        }
    }
    if (oy < p.H && ox < p.W) {
        const size_t HW = (size_t)p.H * p.W, q = (size_t)oy * p.W + ox;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (c < p.Ct) {
                const float y = o[c] + p.bt[c];
                const size_t oo = ((size_t)b * p.Ct + c) * HW + q;
                if (p.kind[c] == 1) {
                    p.out[oo] = 1.0f / (1.0f + expf(-y));
                    if (p.out_logits) p.out_logits[oo] = y;
                } else {
                    p.out[oo] = y * p.a[c] + p.d[c];
                }
            }
        }
    }
}

}  // namespace

extern "C" int ufm_dpt_tail_fused(const uint16_t* in, int B, int h, int w, int Cin, const uint16_t* w2, const float* b2,
                                  int Cmid, int H, int W, const float* wt, const float* bt, int Ct,
                                  const int32_t* kind_host, const float* a_host, const float* d_host, float* out,
                                  float* out_logits, int64_t in_plane, void* stream) {
    UFM_REQUIRE(in && w2 && b2 && wt && bt && out && kind_host && a_host && d_host, "ufm_dpt_tail_fused: null pointer");
    UFM_REQUIRE(Cin == CIN && Cmid == CMID, "ufm_dpt_tail_fused: built for 128 -> 32 channels, got %d -> %d", Cin, Cmid);
    UFM_REQUIRE(B > 0 && h > 1 && w > 1 && H > 1 && W > 1 && Ct >= 1 && Ct <= 8, "ufm_dpt_tail_fused: bad shape");
    UFM_REQUIRE((double)(h - 1) / (H - 1) * (TS + 1) + 2.0 <= TROWS && h <= H && w <= W,
                "ufm_dpt_tail_fused: built for up-sampling ratios (h-1)/(H-1) <= 0.64 (got %d -> %d)", h, H);
    UFM_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)w2 % 16) == 0 && ((uintptr_t)wt % 16) == 0 && ((uintptr_t)b2 % 16) == 0,
                "ufm_dpt_tail_fused: misaligned pointer");
    TailFusedArgs p{};
    p.t_swap = (ufm_upsample_variant_flags() & 2) ? 0 : 4;
    p.in = in, p.w2 = w2, p.b2 = b2, p.wt = wt, p.bt = bt, p.out = out, p.out_logits = out_logits;
    UFM_REQUIRE(in_plane == 0 || in_plane >= (int64_t)B * h * w * CIN, "ufm_dpt_tail_fused: in_plane is shorter than the %d images", B);
    p.in_plane = in_plane ? (long long)in_plane : (long long)B * h * w * CIN, p.w_plane = (long long)CMID * 9 * CIN;
    p.B = B, p.h = h, p.w = w, p.H = H, p.W = W, p.Ct = Ct;
    p.sy = (float)(h - 1) / (float)(H - 1), p.sx = (float)(w - 1) / (float)(W - 1);  // as ufm_upsample_bilinear_nhwc
    for (int c = 0; c < 8; ++c) {
        p.kind[c] = c < Ct ? kind_host[c] : 0;
        p.a[c] = c < Ct ? a_host[c] : 1.f;
        p.d[c] = c < Ct ? d_host[c] : 0.f;
    }
    const long long tiles = (long long)B * ((H + TS - 1) / TS) * ((W + TS - 1) / TS);
    UFM_REQUIRE(tiles < (1ll << 31), "ufm_dpt_tail_fused: problem too large");
    if (Ct <= 4) hipLaunchKernelGGL(dpt_tail_fused_kernel<4>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(dpt_tail_fused_kernel<8>, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, p);
    UFM_CHECK_LAUNCH("ufm_dpt_tail_fused");
    return UFM_OK;
}
