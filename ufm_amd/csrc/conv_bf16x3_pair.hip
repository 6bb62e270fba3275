// 256 px x 128 cout x 32 ch "pair" bf16x3 implicit-GEMM convolution (round 5): the 8-phase wave schedule of conv_bf16x3_8ph.hip in a
// FOUR-wave workgroup with the ring layout of gemm_bf16_pair.hip, so that two workgroups share a CU -- for the layers whose Cout is a
// multiple of 128 but not of 256 (the DPT heads' p_conv1, 296^2 x 256 -> 128: the largest single convolution of a step, until now on
// the two-stage 128 x 128 kernel, which is LDS-bandwidth bound by construction: per K-step its two resident workgroups read 16 KiB x 8
// waves and receive 64 KiB by DMA = 1536 LDS cycles against 1536 matrix-pipe cycles).  Per wave 128 px x 64 cout as in the 8-phase kernel:
// a third fewer LDS read bytes per MFMA than the 64 x 64 wave tile, a quarter less DMA per FLOP than the 128 x 128 tile.
//
//   * A K-tile (32 channels of one filter tap) is staged as four half-tiles in the order the wave's four 64 px x 32 cout quadrants consume
//     them: kind 0 = W-lo (couts nh = 0 of both wave columns: [2 planes][64 rows][64 B] = 8 KiB), 1 = X-lo (pixels mh = 0 of both wave
//     rows: [2][128][64 B] = 16 KiB), 2 = W-hi, 3 = X-hi.  X and W half-tiles each live in a ring of THREE buffers (48 + 24 KiB = 72 KiB):
//     half-tile s + 6 takes the buffer of s.
//   * One phase = {8 / 4 / 8 / 4 ds_read_b128, the DMA of half-tile phase + 6 (4 or 2 global_load_lds_dwordx4 per wave: hi and lo plane of
//     the wave's row pieces; the A rows gathered through the per-lane source address, halo from the zero page), s_waitcnt vmcnt(12),
//     lgkmcnt(0), ONE s_barrier, 24 MFMA (wl*ah, wh*al, wh*ah per fragment pair)}.
//   * Ordering exactly as gemm_bf16_pair.hip: RAW -- half-tile s is waited for in phase s - 2 in front of that phase's barrier and read in
//     phase s - 1; WAR -- the DMA of s + 6 into the buffer of s is issued in phase s, behind the barrier of phase s - 1, in front of which
//     every wave retired (lgkmcnt(0)) its reads of s.  A second prologue barrier orders phase 0's re-fill of W slot 0 behind W-lo(0)'s read.
//   * Same arithmetic in the same order as conv_x3_kernel (K-tile = 32 channels of one tap, channel-chunk major / tap inner): BIT-IDENTICAL
//     to the 128-row kernels (tests/test_kernels_gpu.py).
#include "conv_x3_common.h"

namespace {

constexpr int XPLANE = 128 * 64, WPLANE = 64 * 64;  // bytes of one plane of an X / W half-tile
constexpr int XH = 2 * XPLANE, WH = 2 * WPLANE;     // 16 KiB, 8 KiB
constexpr int W_RING = 3 * XH;
constexpr int PAIR_LDS = 3 * XH + 3 * WH;           // 72 KiB

template <int K>
using IC = std::integral_constant<int, K>;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct TapIter {  // walks the K-tiles of one half-tile kind: channel-chunk outer, filter tap inner
    int tap, kh, kw, c0;
};

__global__ __launch_bounds__(256, 2) void conv_x3_pair_kernel(ConvX3Args p) {
    __shared__ __attribute__((aligned(16))) char smem[PAIR_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int ntn = p.Cout / 128;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tmi_all = bid / ntn, tni = bid - tmi_all * ntn;
    int grp, tmi;
    conv_x3_group_of(p, tmi_all, (p.M - p.m_begin + 255) / 256, grp, tmi);
    const uint16_t* const in_g = p.in + (size_t)grp * p.in_group;
    const uint16_t* const w_g = p.w + (size_t)grp * p.w_group;
    const int m0 = p.m_begin + tmi * 256, n0 = tni * 128;
    const int ntaps = p.KH * p.KW;
    const int nt = ntaps * (p.Cin >> 5);
    const unsigned ktot = (unsigned)(ntaps * p.Cin);

    // ---- DMA sources.  X half-tile: 8 row pieces of 16 pixels -> wave w stages row pieces w and 4 + w, hi and lo plane;
    // W half-tile: 4 row pieces of 16 couts -> wave w stages row piece w, both planes ----
    const int srow = lane >> 2, slot = lane & 3;
    int x_iy0[2][2], x_ix0[2][2];
    unsigned x_img[2][2], x_chunk[2], w_src[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lr = (i * 4 + wave) * 16 + srow;
        x_chunk[i] = (unsigned)((slot ^ swz(lr)) * 8);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int brow = (lr >> 6) * 128 + h * 64 + (lr & 63);  // X half h: pixel rows mh = h of both wave rows
            const int m = min(m0 + brow, p.M - 1);
            const int ox = m % p.Wo, t = m / p.Wo, oy = t % p.Ho, b = t / p.Ho;
            x_iy0[h][i] = oy * p.stride - p.pad;
            x_ix0[h][i] = ox * p.stride - p.pad;
            x_img[h][i] = (unsigned)b * (unsigned)(p.H * p.W);
        }
    }
    const int wlr = wave * 16 + srow;
    const unsigned w_chunk = (unsigned)((slot ^ swz(wlr)) * 8);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int bcol = (wlr >> 5) * 64 + h * 32 + (wlr & 31);     // W half h: couts nh = h of both wave columns
        w_src[h] = (unsigned)(n0 + bcol) * ktot + w_chunk;
    }
    TapIter it[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    auto stage = [&](auto kind, auto buf) {  // kind: 0 W-lo, 1 X-lo, 2 W-hi, 3 X-hi; issues this kind's NEXT K-tile into ring slot buf
        constexpr int KIND = decltype(kind)::value, BUF = decltype(buf)::value;
        constexpr int H = KIND >> 1;
        TapIter& ti = it[KIND];
        if constexpr (KIND & 1) {
            char* base = smem + BUF * XH;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int iy = x_iy0[H][i] + ti.kh, ix = x_ix0[H][i] + ti.kw;
                if (p.replicate) iy = min(max(iy, 0), p.H - 1), ix = min(max(ix, 0), p.W - 1);
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const uint16_t* src = ok ? in_g + ((x_img[H][i] + (unsigned)(iy * p.W + ix)) * (unsigned)p.Cin + (unsigned)ti.c0 + x_chunk[i]) : p.zero + x_chunk[i];
                const uint16_t* src_lo = ok ? src + p.in_plane : src;
                char* dst = base + (i * 4 + wave) * 1024;
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(dst), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src_lo), LDS_PTR(dst + XPLANE), 16, 0, 0);
            }
        } else {
            const uint16_t* src = w_g + (w_src[H] + (unsigned)(ti.tap * p.Cin + ti.c0));
            char* dst = smem + W_RING + BUF * WH + wave * 1024;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(dst), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + p.w_plane), LDS_PTR(dst + WPLANE), 16, 0, 0);
        }
        if (++ti.kw == p.KW) ti.kw = 0, ++ti.kh;
        if (++ti.tap == ntaps) ti.tap = 0, ti.kh = 0, ti.kw = 0, ti.c0 += 32;
    };

    // ---- fragment read offsets (16x16x32: lane (fr, fq) reads row fr, 16-byte chunk fq) ----
    const int fr = lane & 15, fq = lane >> 4;
    int x_off[4], w_off[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wr * 64 + i * 16 + fr;
        x_off[i] = r * 64 + ((fq ^ swz(r)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = wc * 32 + j * 16 + fr;
        w_off[j] = r * 64 + ((fq ^ swz(r)) << 4);
    }

    f32x4 acc[2][4][4];  // [mh][n][m]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[h][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[4][2], wa[2][2], wb[2][2];  // [frag][plane]

    auto read_x = [&](auto buf) {
        const char* s = smem + decltype(buf)::value * XH;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xf[i][0] = *(const bf16x8*)(s + x_off[i]);
            xf[i][1] = *(const bf16x8*)(s + XPLANE + x_off[i]);
        }
    };
    auto read_w = [&](bf16x8 (&w)[2][2], auto buf) {
        const char* s = smem + W_RING + decltype(buf)::value * WH;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            w[j][0] = *(const bf16x8*)(s + w_off[j]);
            w[j][1] = *(const bf16x8*)(s + WPLANE + w_off[j]);
        }
    };
    auto mma = [&](auto mh_, auto nh_, bf16x8 (&w)[2][2], bool fresh_x) {
        constexpr int MH = decltype(mh_)::value, NH = decltype(nh_)::value;
        if (fresh_x && p.relu_in) {  // ReLU on the input: the sign of hi decides for both halves
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x8 neg = xf[i][0] >> 15;
                xf[i][0] &= ~neg;
                xf[i][1] &= ~neg;
            }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4& a = acc[MH][NH * 2 + j][i];
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][1], xf[i][0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][0], xf[i][1], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][0], xf[i][0], a, 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // end of the load slot of phase ph = 4 tile + I: issue half-tile ph + 6 into the ring slot of half-tile ph, wait for half-tile ph + 2,
    // retire this phase's fragment reads, barrier.  B2 = ring slot of the tile's lo halves = (2 tile) % 3.
    const int nhalf = 4 * nt;
    auto l_end = [&](int tile, auto i_, auto b2_) {
        constexpr int I = decltype(i_)::value, B2 = decltype(b2_)::value;
        const int ph = 4 * tile + I;
        if (ph + 6 < nhalf) {
            stage(IC<(I + 2) & 3>{}, IC<(I < 2) ? B2 : (B2 + 1) % 3>{});
            wait_vmcnt<12>();  // one younger half-tile of each kind: 2 + 4 + 2 + 4 pieces
        } else {
            const int inflight = nhalf - ph - 3;  // the half-tiles younger than ph + 2 are the LAST ones of the stream: kinds .. 1, 2, 3
            if (inflight >= 3) wait_vmcnt<10>();
            else if (inflight == 2) wait_vmcnt<6>();
            else if (inflight == 1) wait_vmcnt<4>();
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this phase's fragment reads are retired in front of the barrier (WAR)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto tile_body = [&](int t, auto b2_, bf16x8 (&wcur)[2][2], bf16x8 (&wnxt)[2][2]) {  // wcur holds W-lo(t) on entry
        constexpr int B2 = decltype(b2_)::value, B2H = (B2 + 1) % 3, B2N = (B2 + 2) % 3;
        read_x(IC<B2>{});
        l_end(t, IC<0>{}, b2_);
        mma(IC<0>{}, IC<0>{}, wcur, true);
        read_w(wnxt, IC<B2H>{});
        l_end(t, IC<1>{}, b2_);
        mma(IC<0>{}, IC<1>{}, wnxt, false);
        read_x(IC<B2H>{});
        l_end(t, IC<2>{}, b2_);
        mma(IC<1>{}, IC<1>{}, wnxt, true);
        if (t + 1 < nt) read_w(wnxt, IC<B2N>{});  // W-lo of the next K-tile into the set W-hi(t) just vacated
        l_end(t, IC<3>{}, b2_);
        mma(IC<1>{}, IC<0>{}, wcur, false);
    };

    // ---- prologue: half-tiles 0..5 (host guarantees nt >= 2): tile 0 in slots 0 / 1, the lo halves of tile 1 in slot 2 ----
    stage(IC<0>{}, IC<0>{});
    stage(IC<1>{}, IC<0>{});
    stage(IC<2>{}, IC<1>{});
    stage(IC<3>{}, IC<1>{});
    stage(IC<0>{}, IC<2>{});
    stage(IC<1>{}, IC<2>{});
    wait_vmcnt<12>();  // half-tiles 0 (W-lo) and 1 (X-lo) of tile 0 have landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_w(wb, IC<0>{});
    __builtin_amdgcn_s_waitcnt(0xC07F);  // phase 0 re-fills W slot 0 (half-tile 6): every wave's read of half-tile 0 is retired first
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    for (int t = 0;;) {  // (2 t) % 3 has period 3, the W register sets swap every tile: six tile bodies per trip
        tile_body(t, IC<0>{}, wb, wa);
        if (++t >= nt) break;
        tile_body(t, IC<2>{}, wa, wb);
        if (++t >= nt) break;
        tile_body(t, IC<1>{}, wb, wa);
        if (++t >= nt) break;
        tile_body(t, IC<0>{}, wa, wb);
        if (++t >= nt) break;
        tile_body(t, IC<2>{}, wb, wa);
        if (++t >= nt) break;
        tile_body(t, IC<1>{}, wa, wb);
        if (++t >= nt) break;
    }
    // every wave retired its last ds_read in front of the last phase's barrier and every DMA has landed: the rings are free for the
    // epilogue staging (16 KiB per wave, two 64 x 64 slices)
    conv_x3_epilogue<4, 4>(p, acc[0], smem + wave * 16384, m0 + wr * 128, n0 + wc * 64, lane, grp);
    conv_x3_epilogue<4, 4>(p, acc[1], smem + wave * 16384, m0 + wr * 128 + 64, n0 + wc * 64, lane, grp);
}

}  // namespace

int ufm_launch_conv_x3_pair(const ConvX3Args& p, hipStream_t stream) {
    const int ntm = (p.M - p.m_begin + 255) / 256 * p.groups;
    hipLaunchKernelGGL(conv_x3_pair_kernel, dim3(ntm * (p.Cout / 128)), dim3(256), 0, stream, p);
    return 0;
}
