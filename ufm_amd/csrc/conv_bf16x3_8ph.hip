// 256 px x 256 cout (or 512 px x 128 cout) x 32 ch "8-phase" bf16x3 implicit-GEMM convolution: the schedule of gemm_bf16_8ph.hip (ping-pong
// wave groups, counted vmcnt, four half-tiles in flight, no vmcnt(0)/__syncthreads in the loop) applied to the
// split-precision convolution of conv_bf16x3.hip.  Same arithmetic in the same order (K-tile = 32 channels of one
// filter tap, channel-chunk major / tap inner; per K-tile wl*ah, wh*al, wh*ah), so its results are BIT-IDENTICAL
// to conv_x3_kernel's -- the parity test checks exactly that.
//
// What changes against the GEMM form:
//   * a half-tile is [2 planes (hi, lo)][128 rows][64 B] = 16 KiB: the two planes play the role of the GEMM's two
//     32-deep k-substeps, a phase is 8 x 3 = 24 MFMA, and the reads per phase stay 8 / 4 / 8 / 4 ds_read_b128;
//   * a wave's two DMA pieces of a half-tile are the hi and lo plane of the same 16 rows; the A rows are gathered
//     (output pixel + filter tap, halo from the zero page) through the per-lane source address;
//   * 64-B LDS rows: 16-byte chunk index XOR g[(row >> 2) & 3], g = {0,2,3,1} (conv_bf16x3.hip);
//   * Cout = 256 is ONE column tile: every input pixel is gathered once (the 128x128 kernel gathers it twice).
#include "conv_x3_common.h"

namespace {

// Wave layouts (WR x WC waves, 128 px x 64 cout per wave in both):
//   2 x 4 : 256 px x 256 cout  -- half-tiles X 128 rows (16 KiB), W 128 rows (16 KiB); 128 KiB of LDS
//   4 x 2 : 512 px x 128 cout  -- half-tiles X 256 rows (32 KiB), W  64 rows ( 8 KiB); 160 KiB of LDS (all of it)
// A half-tile is [plane (hi, lo)][rows][64 B]; a DMA piece is 16 rows of one plane (1 KiB).

template <int K>
using IC = std::integral_constant<int, K>;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct TapIter {  // walks the K-tiles of one half-tile kind: channel-chunk outer, filter tap inner
    int tap, kh, kw, c0;
};

// NF = 16-row fragments per wave row (5..8; WR = 2 only): the tile is WR x 16 NF pixels high, as in gemm_bf16_8ph.hip -- a wave's second
// 64-row half has NF - 4 fragments; staging, schedule and every accumulator's order are those of NF = 8 (bit-identical).  Lower tiles
// let the Linear layers of numerics "precise" (ufm_gemm_bf16x3: M = 21 920 rows x N = 1024 is 344 tiles of 256 rows on 256 CUs) fill
// whole rounds of the chip.  (Round 5 also tried the phase's 24 MFMAs product-major over its eight accumulators instead of three
// per accumulator back to back: neutral, 478.7 vs 479.2 us on the 148^2 layer, tools/lab/conv_stamps.py -- not kept.)
// ABL: timing ablations of the loop (ConvX3Args::ablate; a diagnostic instantiation of its own, so that the stamped kernel stays the shipped loop)
// IL (round 6): the Linear form on INTERLEAVED split operands -- A [M][K / 32][hi 32 | lo 32], W [N][K / 32][hi 32 | lo 32] -- so that every
// LDS-DMA row is a whole 128-byte line (8 lanes x 16 B: the hi and the lo half of one 32-channel chunk of one row) instead of two 64-byte
// halves 2 M K bytes apart.  LDS half-tile = [128 rows][128 B], 16-byte chunk c = 4 plane + fq stored at position c ^ (row & 7) (the bf16
// GEMM's layout, with the planes in the place of its two k-substeps).  Same K order, same products: bit-identical to the planar form.
template <int WR, int WC, bool STAMP = false, int NF = 8, bool ABL = false, bool IL = false>
__global__ __launch_bounds__(512, 1) void conv_x3_8ph_kernel(ConvX3Args p) {
    static_assert(WR * WC == 8, "eight waves");
    static_assert(!IL || (WR == 2 && !ABL), "the interleaved form is built for the 2 x 4 layout");
    static_assert(NF >= 5 && NF <= 8 && (NF == 8 || WR == 2), "NF");
    constexpr int RW = 16 * NF;  // pixel rows of one wave row
    GemmStamps stamps;  // (diagnostic instantiation only: the shipped kernel executes no stamp)
    if constexpr (STAMP) stamps.entry(), stamps.t_prologue = stamps.t_entry;
    constexpr int XROWS = WR * 64, WROWS = WC * 32;           // rows of an X / W half-tile
    constexpr int XPLANE = XROWS * 64, WPLANE = WROWS * 64;   // bytes of one plane
    constexpr int XHALF = 2 * XPLANE, WHALF = 2 * WPLANE;
    constexpr int KTILE = 2 * (XHALF + WHALF);                // one K-tile buffer: [W-lo][X-lo][W-hi][X-hi]
    constexpr int XP = XROWS / 16 * 2 / 8, WP = WROWS / 16 * 2 / 8;  // DMA pieces per wave and half-tile: 2|4 and 2|1
    constexpr int INFLIGHT = 2 * (XP + WP);                   // pieces of the four half-tiles kept in flight
    __shared__ __attribute__((aligned(16))) char smem[2 * KTILE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    auto half_off = [](int kind) { return kind == 0 ? 0 : kind == 1 ? WHALF : kind == 2 ? WHALF + XHALF : 2 * WHALF + XHALF; };

    const int ntn = p.Cout / (WC * 64);
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tmi_all = bid / ntn, tni = bid - tmi_all * ntn;
    int grp, tmi;
    conv_x3_group_of(p, tmi_all, (p.M - p.m_begin + WR * RW - 1) / (WR * RW), grp, tmi);
    const uint16_t* const in_g = p.in + (size_t)grp * p.in_group;
    const uint16_t* const w_g = p.w + (size_t)grp * p.w_group;
    const int m0 = p.m_begin + tmi * (WR * RW), n0 = tni * (WC * 64);
    const int ntaps = p.KH * p.KW;
    const int nt = ntaps * (p.Cin >> 5);
    const unsigned ktot = (unsigned)(ntaps * p.Cin);

    // ---- DMA sources.  X half-tile: wave w stages row-pieces w (and 8 + w when there are 16), hi and lo plane.
    // W half-tile: 8 row-pieces -> wave w stages piece w, both planes; 4 row-pieces -> piece w & 3 of plane w >> 2 ----
    const int srow = lane >> 2, slot = lane & 3;
    constexpr int XPR = XP / 2;  // X row-pieces per wave
    int x_iy0[2][XPR], x_ix0[2][XPR];
    unsigned x_img[2][XPR], x_chunk[XPR], w_src[2];
#pragma unroll
    for (int i = 0; i < XPR; ++i) {
        const int lr = (i * 8 + wave) * 16 + srow;
        x_chunk[i] = (unsigned)((slot ^ swz(lr)) * 8);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int brow = (lr >> 6) * RW + h * 64 + (lr & 63);  // X half h: pixel rows mh = h of every wave row (rows past RW: unused)
            const int m = min(m0 + brow, p.M - 1);
            const int ox = m % p.Wo, t = m / p.Wo, oy = t % p.Ho, b = t / p.Ho;
            x_iy0[h][i] = oy * p.stride - p.pad;
            x_ix0[h][i] = ox * p.stride - p.pad;
            x_img[h][i] = (unsigned)b * (unsigned)(p.H * p.W);
        }
    }
    const int w_piece = WP == 2 ? wave : (wave & 3), w_plane_sel = WP == 2 ? 0 : (wave >> 2);
    const int wlr = w_piece * 16 + srow;
    const unsigned w_chunk = (unsigned)((slot ^ swz(wlr)) * 8);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int bcol = (wlr >> 5) * 64 + h * 32 + (wlr & 31);     // W half h: couts nh = h of every wave column
        w_src[h] = (unsigned)(n0 + bcol) * ktot + w_chunk;
    }
    // IL: piece = 8 rows x 128 B; wave w stages pieces w and 8 + w of every half-tile; lane (l >> 3, l & 7) = row, 16-byte position
    unsigned xil_src[2][2], wil_src[2][2];  // [half][piece]: element offsets
    if constexpr (IL) {
        const int prow = lane >> 3, ppos = lane & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = (i * 8 + wave) * 8 + prow;  // row of the half-tile, 0..127
            const unsigned chunk = (unsigned)((ppos ^ (lr & 7)) * 8);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int brow = (lr >> 6) * RW + h * 64 + (lr & 63);
                const int bcol = (lr >> 5) * 64 + h * 32 + (lr & 31);
                xil_src[h][i] = (unsigned)min(m0 + brow, p.M - 1) * (unsigned)(2 * p.Cin) + chunk;
                wil_src[h][i] = (unsigned)(n0 + bcol) * (unsigned)(2 * p.Cin) + chunk;
            }
        }
    }
    TapIter it[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    auto stage = [&](auto kind, int tile) {  // kind: 0 W-lo, 1 X-lo, 2 W-hi, 3 X-hi ; issues this kind's NEXT K-tile
        constexpr int KIND = decltype(kind)::value;
        constexpr int H = KIND >> 1;
        TapIter& ti = it[KIND];
        char* base = smem + (tile & 1) * KTILE + half_off(KIND);
        if constexpr (ABL) {
            if (p.ablate & 1) return;  // timing ablation: no LDS-DMA (the counted waits fall through)
            if ((p.ablate & 64) && (KIND & 1) && ti.tap != 0) {  // timing emulation of a halo tile: the X pieces of one tap in nine only
                if (++ti.kw == p.KW) ti.kw = 0, ++ti.kh;
                if (++ti.tap == ntaps) ti.tap = 0, ti.kh = 0, ti.kw = 0, ti.c0 += 32;
                return;
            }
        }
        if constexpr (IL) {  // 1x1 on interleaved operands: K-tile = chunk c0 / 32 = 64 elements [hi 32 | lo 32] of every row
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint16_t* src = (KIND & 1) ? in_g + (xil_src[H][i] + (unsigned)(2 * ti.c0)) : w_g + (wil_src[H][i] + (unsigned)(2 * ti.c0));
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(base + (i * 8 + wave) * 1024), 16, 0, 0);
            }
            ti.c0 += 32;
            return;
        }
        if (KIND & 1) {
#pragma unroll
            for (int i = 0; i < XPR; ++i) {
                int iy = x_iy0[H][i] + ti.kh, ix = x_ix0[H][i] + ti.kw;
                if (p.replicate) iy = min(max(iy, 0), p.H - 1), ix = min(max(ix, 0), p.W - 1);
                const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const uint16_t* src = ok ? in_g + ((x_img[H][i] + (unsigned)(iy * p.W + ix)) * (unsigned)p.Cin + (unsigned)ti.c0 + x_chunk[i]) : p.zero + x_chunk[i];
                const uint16_t* src_lo = ok ? src + p.in_plane : src;
                char* dst = base + (i * 8 + wave) * 1024;
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(dst), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src_lo), LDS_PTR(dst + XPLANE), 16, 0, 0);
            }
        } else {
            const uint16_t* src = w_g + (w_src[H] + (unsigned)(ti.tap * p.Cin + ti.c0));
            char* dst = base + w_piece * 1024;
            if (WP == 2) {
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(dst), 16, 0, 0);
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + p.w_plane), LDS_PTR(dst + WPLANE), 16, 0, 0);
            } else {
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + (w_plane_sel ? p.w_plane : 0)), LDS_PTR(dst + w_plane_sel * WPLANE), 16, 0, 0);
            }
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

    // IL: row r of a half-tile at r * 128 B; plane pl, k-quarter fq at 16-byte position (4 pl + fq) ^ (r & 7); (r & 7) = fr & 7 for every fragment
    int xil_off[2], wil_off[2];  // [plane]: fragment 0 of this wave; fragment i: + i * 2048
    if constexpr (IL) {
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            xil_off[pl] = (wr * 64 + fr) * 128 + (((4 * pl + fq) ^ (fr & 7)) << 4);
            wil_off[pl] = (wc * 32 + fr) * 128 + (((4 * pl + fq) ^ (fr & 7)) << 4);
        }
    }
    f32x4 acc[2][4][4];  // [mh][n][m]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[h][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[4][2], wa[2][2], wb[2][2];  // [frag][plane]

    auto read_x = [&](int tile, int mh) {
        if constexpr (ABL) {
            if (p.ablate & 2) return;  // timing ablation: no fragment reads
        }
        const char* s = smem + (tile & 1) * KTILE + half_off(1 + 2 * mh);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (mh == 1 && i >= NF - 4) continue;
            if constexpr (IL) {
                xf[i][0] = *(const bf16x8*)(s + xil_off[0] + i * 2048);
                xf[i][1] = *(const bf16x8*)(s + xil_off[1] + i * 2048);
            } else {
                xf[i][0] = *(const bf16x8*)(s + x_off[i]);
                xf[i][1] = *(const bf16x8*)(s + XPLANE + x_off[i]);
            }
        }
    };
    auto read_w = [&](bf16x8 (&w)[2][2], int tile, int nh) {
        if constexpr (ABL) {
            if (p.ablate & 2) return;
        }
        const char* s = smem + (tile & 1) * KTILE + half_off(2 * nh);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (IL) {
                w[j][0] = *(const bf16x8*)(s + wil_off[0] + j * 2048);
                w[j][1] = *(const bf16x8*)(s + wil_off[1] + j * 2048);
            } else {
                w[j][0] = *(const bf16x8*)(s + w_off[j]);
                w[j][1] = *(const bf16x8*)(s + WPLANE + w_off[j]);
            }
        }
    };
    auto mma = [&](auto mh_, auto nh_, bf16x8 (&w)[2][2], bool fresh_x) {
        constexpr int MH = decltype(mh_)::value, NH = decltype(nh_)::value;
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        if (fresh_x && p.relu_in) {  // ReLU on the input: the sign of hi decides for both halves
#pragma unroll
            for (int i = 0; i < (MH == 0 ? 4 : NF - 4); ++i) {
                const bf16x8 neg = xf[i][0] >> 15;
                xf[i][0] &= ~neg;
                xf[i][1] &= ~neg;
            }
        }
        __builtin_amdgcn_s_setprio(1);
        bool do_mfma = true, do_bar = true;
        if constexpr (ABL) do_mfma = !(p.ablate & 4), do_bar = !(p.ablate & 8) && !((p.ablate & 32) && wr == 0);  // timing ablations; 32: one rendezvous per phase (group 0 keeps the l_end barriers, group 1 the mma barriers) -- NOT a legal schedule: a half-tile is read one phase after its landing wait, the mid-phase rendezvous is what publishes it
        if (do_mfma) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < (MH == 0 ? 4 : NF - 4); ++i) {
                    f32x4& a = acc[MH][NH * 2 + j][i];
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][1], xf[i][0], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][0], xf[i][1], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][0], xf[i][0], a, 0, 0, 0);
                }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (do_bar) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // end of the L slot of phase ph (= 4 * tile + i): issue half-tile ph + 6, wait for half-tile ph + 2, barrier
    const int nhalf = 4 * nt;
    auto l_end = [&](int tile, auto i_) {
        constexpr int I = decltype(i_)::value;
        const int ph = 4 * tile + I;
        if (ph + 6 < nhalf) {
            stage(IC<(I + 2) & 3>{}, tile + (I + 6) / 4);
            wait_vmcnt<INFLIGHT>();  // the four youngest half-tiles are always two X and two W
        } else {
            const int inflight = nhalf - ph - 3;  // half-tiles issued after half-tile ph + 2: the LAST ones of the stream
            if (inflight >= 3) wait_vmcnt<2 * XP + WP>();   // X-lo, W-hi, X-hi
            else if (inflight == 2) wait_vmcnt<XP + WP>();  // W-hi, X-hi
            else if (inflight == 1) wait_vmcnt<XP>();       // X-hi
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_sched_barrier(0);
        bool do_bar = true;
        if constexpr (ABL) do_bar = !(p.ablate & 8) && !((p.ablate & 32) && wr == 1);
        if (do_bar) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto tile_body = [&](int t, bf16x8 (&wcur)[2][2], bf16x8 (&wnxt)[2][2]) {  // wcur holds W-lo(t) on entry
        read_x(t, 0);
        l_end(t, IC<0>{});
        mma(IC<0>{}, IC<0>{}, wcur, true);
        read_w(wnxt, t, 1);
        l_end(t, IC<1>{});
        mma(IC<0>{}, IC<1>{}, wnxt, false);
        read_x(t, 1);
        l_end(t, IC<2>{});
        mma(IC<1>{}, IC<1>{}, wnxt, true);
        if (t + 1 < nt) read_w(wnxt, t + 1, 0);  // W-lo of the next K-tile into the set W-hi(t) just vacated
        l_end(t, IC<3>{});
        mma(IC<1>{}, IC<0>{}, wcur, false);
    };

    // ---- prologue: half-tiles 0..5 (host guarantees nt >= 2) ----
    stage(IC<0>{}, 0);
    stage(IC<1>{}, 0);
    stage(IC<2>{}, 0);
    stage(IC<3>{}, 0);
    stage(IC<0>{}, 1);
    stage(IC<1>{}, 1);
    wait_vmcnt<INFLIGHT>();  // half-tiles 0 (W-lo) and 1 (X-lo) of tile 0 have landed: W-hi, X-hi, W-lo, X-lo may be in flight
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_w(wb, 0, 0);
    if (wr == 1) {  // stagger: the wr = 1 group runs one slot behind
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }

    int t = 0;
    for (; t + 1 < nt; t += 2) {
        tile_body(t, wb, wa);
        tile_body(t + 1, wa, wb);
    }
    if (t < nt) tile_body(t, wb, wa);

    if (wr == 0) __builtin_amdgcn_s_barrier();  // pairs with the last M-slot barrier of the wr = 1 group
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (STAMP) stamps.t_loop = gemm_stamp();
    // two explicit calls: a loop over h that hipcc declines to unroll would index acc[h] at run time -> scratch (rule 20)
    conv_x3_epilogue<4, 4>(p, acc[0], smem + wave * 16384, m0 + wr * RW, n0 + wc * 64, lane, grp);
    conv_x3_epilogue<4, 4, NF - 4>(p, acc[1], smem + wave * 16384, m0 + wr * RW + 64, n0 + wc * 64, lane, grp);
    if constexpr (STAMP) stamps.finish(p.stamps, p.stamp_rows);
}

}  // namespace

// The Linear form on interleaved split operands (ConvX3Args of a 1x1 layer over M rows; in = A_il, w = W_il; in_plane / w_plane unused)
int ufm_launch_gemm_x3_il_8ph(const ConvX3Args& p, hipStream_t stream, int nf) {
    const int rows = 32 * nf;
    const dim3 grid((p.M - p.m_begin + rows - 1) / rows * (p.Cout / 256)), block(512);
    if (nf == 5) hipLaunchKernelGGL((conv_x3_8ph_kernel<2, 4, false, 5, false, true>), grid, block, 0, stream, p);
    else if (nf == 6) hipLaunchKernelGGL((conv_x3_8ph_kernel<2, 4, false, 6, false, true>), grid, block, 0, stream, p);
    else if (nf == 7) hipLaunchKernelGGL((conv_x3_8ph_kernel<2, 4, false, 7, false, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((conv_x3_8ph_kernel<2, 4, false, 8, false, true>), grid, block, 0, stream, p);
    return 0;
}

int ufm_launch_conv_x3_8ph(const ConvX3Args& p, hipStream_t stream, int nf) {
    // 256 px x 256 cout tiles (wave layout 2 x 4), 32 nf px high.  The kernel is templated on the wave layout; the 4 x 2 layout (512 px x
    // 128 cout for Cout = 128) measured slower than the 128-row kernels (296^2 256->128: 1328 vs 1130 us) and is no longer instantiated.
    const int rows = 32 * nf;
    const int ntm = (p.M - p.m_begin + rows - 1) / rows * p.groups;
    const dim3 grid(ntm * (p.Cout / 256)), block(512);
    if (p.stamps && nf == 8 && p.ablate) hipLaunchKernelGGL((conv_x3_8ph_kernel<2, 4, true, 8, true>), grid, block, 0, stream, p);  // timing ablations
    else if (p.stamps && nf == 8) hipLaunchKernelGGL((conv_x3_8ph_kernel<2, 4, true>), grid, block, 0, stream, p);  // diagnostic build
    else if (nf == 5) hipLaunchKernelGGL((conv_x3_8ph_kernel<2, 4, false, 5>), grid, block, 0, stream, p);
    else if (nf == 6) hipLaunchKernelGGL((conv_x3_8ph_kernel<2, 4, false, 6>), grid, block, 0, stream, p);
    else if (nf == 7) hipLaunchKernelGGL((conv_x3_8ph_kernel<2, 4, false, 7>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((conv_x3_8ph_kernel<2, 4>), grid, block, 0, stream, p);
    return 0;
}
