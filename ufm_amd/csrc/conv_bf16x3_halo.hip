// 3x3 / stride 1 / pad 1 bf16x3 implicit-GEMM convolution with a ROW-WINDOW HALO TILE (round 6): the 8-phase 256 px x 256 cout kernel of
// conv_bf16x3_8ph.hip with the input fetched ONCE PER FILTER ROW instead of once per tap.
//
// conv_bf16x3_8ph.hip gathers, for every K-tile (32 channels of one tap), the 256 input pixels of that tap by LDS-DMA: the three taps
// (kh, 0), (kh, 1), (kh, 2) of one filter row read the SAME pixels shifted by one -- 96 KiB of LDS-DMA per filter row and channel chunk
// where 34 KiB are distinct.  The timing ablations of round 5 (profiles/r05/conv_loop_ablation.log) price the loop's load slot as longer
// than its MFMA slot and the X pieces as the part of it that a halo removes.  Here:
//
//   * The tile's 2 RW output pixels are consecutive in the flat (image, y, x) order, so the inputs of filter row kh are the 2 RW + 2
//     consecutive input pixels  q = m0 + (kh - 1) W - 1 + e,  e in [0, 2 RW + 2)  -- ONE "window" per (channel chunk, kh), staged as
//     [plane (hi, lo)][NRP row pieces of 16 px][64 B], NRP = 2 NF + 1, in two ping-pong buffers (68 KiB); tap (kh, kw) of output pixel j
//     reads window pixel j + kw.  A "super-tile" = the three K-tiles of one window = 12 phases.
//   * Borders.  Rows: window pixel e belongs to image row y0 + kh - 1, y0 = row of flat pixel m0 - 1 + e; it is a proper input of filter
//     row kh iff that row lies in the same image (kh = 0: y0 != 0, kh = 2: y0 != H - 1) -- decided when the window is STAGED: the void
//     pixels of a window are ONE scalar interval (the nearest image boundary +- a row), and a void lane issues its piece with an out-of-range buffer
//     offset, which writes zeros to its LDS slot (tools/lab/buffer_lds_oob.hip) -- exactly the zero padding of the gather.  Columns: output pixel x = 0 (kw = 0) / x = W - 1 (kw = 2) would read
//     the neighbouring row's end through the flat shift: those fragment rows are zeroed in registers, on the fragments that contain such a
//     pixel only (wave-uniform bit test; a 16-pixel fragment holds a row boundary in 11 % of the cases at W = 148).  A zero operand is what
//     the gather's zero page supplied: same products, same order.  All operands come in by raw buffer loads to LDS (32-bit lane offsets into
//     wave-uniform descriptors): no 64-bit lane addresses, no per-lane flag registers -- the kernel must be spill-free (a scratch reload counts in vmcnt).
//   * K order, product order (wl*ah, wh*al, wh*ah), wave layout, W half-tiles, the two-group ping-pong schedule and the epilogue are those of
//     conv_bf16x3_8ph.hip: results are BIT-IDENTICAL (tests/test_kernels_gpu.py::test_conv2d_bf16x3_halo_bit_identical).
//   * Vector-memory schedule of one wave over the 12 phases of super-tile u (l_end of phase p): p even: the 2 pieces of a W half-tile as
//     before (W-hi(t+1) at I = 0, W-lo(t+2) at I = 2); p = 1, 3: hi + lo piece of window u + 1's row pieces w and 8 + w; p = 5, 7 (NF = 8
//     only): hi / lo of row piece 16 (its 2 real pixels: wave 0; the other waves' pieces go to a 1-KiB scrap area with an out-of-range offset so that
//     every wave counts the same operations; the same happens to every operation scheduled past the end of the K range).  6 instead of 12 X operations per wave and filter row, 34 instead of 96 KiB.
//     Counted waits (all compile-time, `Sched` below): at even p the W half-tile issued four phases earlier = everything but the
//     operations of phases p-3..p; at p = 11 window u + 1 = everything but the operations issued since its last piece.
//   * Hazards.  RAW: a window is waited for by every wave at l_end(12 u + 11) and first read at the start of phase 12 (u + 1), i.e. after
//     group 0 passed its mma barrier of phase 12 u + 11, which group 1 reaches only behind ITS l_end(12 u + 11) wait (the half-tile rule of
//     the 8-phase kernel).  WAR: window u + 1 re-fills the buffer of window u - 1, last read at phase 12 u - 2; its first piece is issued at
//     l_end(12 u + 1), six barrier instances later.  W ring: unchanged.
#include "conv_x3_common.h"

namespace {

template <int K>
using IC = std::integral_constant<int, K>;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// The lane id, recomputed where it is used (volatile: not hoisted).  The loop below is register-tight -- 128 accumulator + 64 fragment registers --
// and every loop-invariant lane constant hipcc keeps alive is a spill whose scratch reload counts in vmcnt.
__device__ __forceinline__ unsigned lane_id_now() {
    unsigned l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// vector-memory operations one wave issues at l_end of phase p of a super-tile (X3: the window has a 17th row piece).  EVERY super-tile issues
// them -- past the end of the K range they are scrap operations (out-of-range offset, scrap destination) -- so one set of counted waits
// serves the whole loop and the loop has no specialised last iteration.
template <bool X3>
struct Sched {
    static constexpr int c(int p) { return (p & 1) == 0 ? 2 : (p == 1 || p == 3) ? 2 : ((p == 5 || p == 7) && X3) ? 1 : 0; }
    // operations younger than the W half-tile issued at phase p - 4
    static constexpr int younger_w(int p) {
        int n = 0;
        for (int q = p - 3; q <= p; ++q) n += c((q + 12) % 12);
        return n;
    }
    // operations younger than the last piece of window u + 1 (issued at phase 7 with X3, else 3), seen from l_end(11)
    static constexpr int younger_x() {
        int n = 0;
        for (int q = (X3 ? 8 : 4); q <= 11; ++q) n += c(q);
        return n;
    }
};

// WIL: the weights are INTERLEAVED per 32-channel chunk, [Cout][9 Cin / 32][hi 32 | lo 32] (ConvX3Args::w_il, packed beside the planar copy): a W piece
// is then 8 couts x one whole 128-byte line instead of 16 couts x 64 B of one plane; LDS half-tile [128 couts][128 B], chunk 4 plane + fq at
// position ^ (cout & 7) (the layout of ufm_gemm_bf16x3_il).  Same operands in the same MFMAs.
template <int NF = 8, bool STAMP = false, bool WIL = false>
__global__ __launch_bounds__(512, 1) void conv_x3_halo_kernel(ConvX3Args p) {
    static_assert(NF >= 5 && NF <= 8, "NF");
    constexpr int WC = 4;
    constexpr int RW = 16 * NF;                      // pixel rows of one wave row; the tile is 2 RW pixels
    constexpr int WPLANE = 128 * 64, WHALF = 2 * WPLANE, WTILE = 2 * WHALF;  // W: [K-tile & 1][lo | hi half][plane][128 couts][64 B]
    constexpr int NRP = 2 * NF + 1;                  // row pieces (16 px) of a window of 2 RW + 2 pixels
    constexpr bool X3 = NRP > 16;
    constexpr int XPLANE = NRP * 1024, WIN = 2 * XPLANE;
    constexpr int WIN0 = 2 * WTILE, SCRAP = WIN0 + 2 * WIN;
    constexpr int LDS_BYTES = SCRAP + 1024 > 8 * 16384 ? SCRAP + 1024 : 8 * 16384;  // (the epilogue stages 8 x 16 KiB)
    GemmStamps stamps;
    if constexpr (STAMP) stamps.entry(), stamps.t_prologue = stamps.t_entry;
    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;

    const int ntn = p.Cout >> 8;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tmi_all = bid / ntn, tni = bid - tmi_all * ntn;
    int grp, tmi;
    conv_x3_group_of(p, tmi_all, (p.M - p.m_begin + 2 * RW - 1) / (2 * RW), grp, tmi);
    const uint16_t* const in_g = p.in + (size_t)grp * p.in_group;
    const uint16_t* const w_g = p.w + (size_t)grp * p.w_group;
    const int m0 = p.m_begin + tmi * (2 * RW), n0 = tni * 256;
    const int nchunks = p.Cin >> 5;
    const int nu = 3 * nchunks;  // super-tiles: one per (channel chunk, filter row); 3 K-tiles each
    const unsigned ktot = (unsigned)(9 * p.Cin);

    // ---- DMA sources: raw buffer loads to LDS (`buffer_load_dwordx4 ... offen lds`).  A lane is a 32-bit byte offset into a wave-uniform
    // descriptor, and a lane whose offset is out of the descriptor's range writes ZEROS to its LDS slot (tools/lab/buffer_lds_oob.hip
    // probes exactly that on gfx950): the zero padding costs one select per piece, no zero page, no 64-bit address arithmetic -- the
    // global_load_lds form of this kernel kept a dozen 64-bit lane addresses live through the loop and spilled (scratch reloads count in
    // vmcnt: fatal for the counted waits). ----
    // every valid voffset + soffset stays below 2^31 (host check: plane sizes), the void offset plus any soffset stays below 2^32: a void lane is
    // out of range whether or not the hardware includes soffset in its range check, and nothing wraps
    constexpr unsigned OOB = 0x80000000u, NUM_RECORDS = 0x80000000u;
    const int srow = lane >> 2, slot = lane & 3;
    // X window: wave w stages row pieces w, 8 + w and (X3) 16 + w; lane (srow, slot) = window pixel e = 16 rp + srow, 16-byte chunk slot ^ swz(e)
    // ONE offset register per operand: the row pieces of a wave are 8 pieces = 128 pixels apart (a scalar offset), the W halves 32 couts
    const unsigned x_src = 2u * ((unsigned)(m0 - 1 + wave * 16 + srow) * (unsigned)p.Cin + (unsigned)((slot ^ swz(srow)) * 8));  // byte offset of (flat pixel m0 - 1 + e, chunk), slot 0, hi plane (wraps for pixels below 0: flagged)
    const unsigned x_slot_b = 2u * 128u * (unsigned)p.Cin;
    // Row borders as ONE scalar interval per window (no per-lane flag register): a window pixel is void for filter row kh iff its flat index f
    // lies in [b HW - lo_kh, b HW + hi_kh) for an image boundary b HW, (lo, hi) = (0, W) / (0, 0) / (W, 0) for kh = 0 / 1 / 2 (the first row
    // of image b / nothing / the last row of image b - 1), open-ended below for b = 0 and above for b = B (outside the tensor).  Boundaries are
    // HW >= 2 W + 272 pixels apart (host check), a window is at most 258: at most ONE boundary's interval can meet a window -- the largest b
    // with b HW - W <= f0 + 257.  In window coordinates e = f - f0, f0 = m0 - 1:
    const int HW = p.H * p.W;
    const int b_near = (m0 - 1 + 257 + p.W) / HW;          // (m0 >= 0: non-negative)
    const int e_bound = b_near * HW - (m0 - 1);              // window coordinate of that image boundary (may lie outside the window)
    const bool b_first = b_near == 0, b_last = b_near == p.B;
    const int wlr = wave * 16 + srow;
    // WIL: piece = 8 couts x 128 B, wave w stages pieces w and 8 + w of a half: lane (l >> 3, l & 7) = cout row 8 piece + (l >> 3), position l & 7
    const int wil_lr = wave * 8 + (lane >> 3);  // row of piece `wave` (piece 8 + wave: + 64 rows = + 2 wave columns = + 128 couts of the tile)
    const unsigned w_src = WIL ? 2u * ((unsigned)(n0 + (wil_lr >> 5) * 64 + (wil_lr & 31)) * (2u * ktot) + (unsigned)(((lane & 7) ^ (wil_lr & 7)) * 8))
                               : 2u * ((unsigned)(n0 + (wlr >> 5) * 64 + (wlr & 31)) * ktot + (unsigned)((slot ^ swz(wlr)) * 8));  // W half 0 (couts nh = 0 of every wave column); half 1: + 32 couts
    const unsigned w_half_b = 2u * 32u * ktot * (WIL ? 2u : 1u);
    const uint16_t* const w_base_g = WIL ? p.w_il + (size_t)grp * 2 * p.w_group : w_g;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w_base_g, 0, NUM_RECORDS, 0x00020000);
    const unsigned w_plane_b = (unsigned)(2 * p.w_plane), x_plane_b = (unsigned)(2 * p.in_plane);
    struct TapIter {
        int tap, c0;
    };
    TapIter it[2] = {{0, 0}, {0, 0}};
    // (the DMA destinations are scalar -- M0 -- so their buffer parity may be a run-time value; the ds_read addresses below take it at compile time:
    // with a run-time (tile & 1) hipcc keeps both variants of every lane address live)
    auto stage_w = [&](auto half_, int tpar) {  // issues this half's NEXT K-tile (channel-chunk outer, tap inner) into W buffer tpar & 1
        constexpr int H = decltype(half_)::value;
        TapIter& ti = it[H];
        const bool live = ti.c0 < p.Cin;  // past the last K-tile: a scrap operation (keeps the counted waits uniform)
        char* dst = smem + (tpar & 1) * WTILE + H * WHALF + wave * 1024;
        char* dst_lo = dst + WPLANE;
        if (!live) dst = dst_lo = smem + SCRAP;
        const unsigned voff = live ? w_src : OOB;
        if constexpr (WIL) {  // the K-tile's chunk is 64 elements [hi 32 | lo 32] of every cout row; second piece: couts + 128 (rows 64..127 of the half)
            const unsigned soff = 4u * (unsigned)(ti.tap * p.Cin + ti.c0) + H * w_half_b;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, LDS_PTR(dst), 16, voff, soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, LDS_PTR(live ? dst + 8192 : dst), 16, voff, soff + 4u * w_half_b, 0, 0);
        } else {
            const unsigned soff = 2u * (unsigned)(ti.tap * p.Cin + ti.c0) + H * w_half_b;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, LDS_PTR(dst), 16, voff, soff, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, LDS_PTR(dst_lo), 16, voff, soff + w_plane_b, 0, 0);
        }
        if (++ti.tap == 9) ti.tap = 0, ti.c0 += 32;
    };
    // hi and lo piece of slot s in {0, 1, 2} of window v = (chunk v / 3, filter row v % 3); PL: 0 = hi only, 1 = lo only, 2 = both
    auto stage_x = [&](auto s_, auto pl_, int vpar, int v) {
        constexpr int S = decltype(s_)::value, PL = decltype(pl_)::value;
        const int ch = v / 3, kh = v - 3 * ch;
        // the descriptor's base carries the window's uniform shift: filter row (kh - 1) W pixels, channel chunk (before the tensor for kh = 0:
        // the lanes that would reach below its start are exactly the ones flagged "image row 0" or "does not exist")
        const uint16_t* base = in_g + ((long long)(kh - 1) * p.W * p.Cin + ch * 32);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, NUM_RECORDS, 0x00020000);
        // void interval [lo, lo + wd) of this window in e, clamped to [0, 512); this lane's pixel is e = 128 S + 16 wave + srow
        int lo = b_first ? 0 : e_bound - (kh == 2 ? p.W : 0), hi = b_last ? 512 : e_bound + (kh == 0 ? p.W : 0);
        lo = min(max(lo, 0), 512), hi = min(max(hi, 0), 512);
        if (v >= nu) lo = 0, hi = 512;  // past the last window: a scrap operation
        const unsigned wd = (unsigned)max(hi - lo, 0);
        // (the slot's 128-pixel step goes into the LANE offset, not soffset: x_src wraps below zero for the pixel in front of the tensor, and
        //  the range check looks at the lane offset alone -- found as five wrong pixels of tile 0 by tools/lab/conv_halo_debug.py)
        const unsigned voff = (lane_id_now() >> 2) + (unsigned)(S * 128 + wave * 16 - lo) < wd ? OOB : x_src + S * x_slot_b;
        char* dst_hi = smem + WIN0 + (vpar & 1) * WIN + (S * 8 + wave) * 1024;
        char* dst_lo = dst_hi + XPLANE;
        if (S * 8 + wave >= NRP || v >= nu) dst_hi = dst_lo = smem + SCRAP;  // this row piece does not exist (NF = 8: 17..23): same operation count, scrap destination
        if constexpr (PL != 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst_hi), 16, voff, 0, 0, 0);
        if constexpr (PL != 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst_lo), 16, voff, x_plane_b, 0, 0);
    };

    // ---- fragment read offsets (16x16x32: lane (fr, fq) reads row fr, 16-byte chunk fq) ----
    const int fr = lane & 15, fq = lane >> 4;
    int x_off[3], w_off;  // x_off[kw]: window pixel wr RW + fr + kw (+ 64 mh + 16 i as immediates); w_off: cout row wc 32 + fr (+ 16 j)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int e = wr * RW + fr + kw;
        x_off[kw] = WIN0 + e * 64 + ((fq ^ swz(e)) << 4);
    }
    {
        const int r = wc * 32 + fr;
        w_off = r * 64 + ((fq ^ swz(r)) << 4);  // (swz looks at bits 2..3 of the row: the same for rows r and r + 16)
    }
    int wil_off[2];  // WIL: [plane]: row r at r * 128 B, chunk 4 plane + fq at position ^ (r & 7); (r & 7) = fr & 7 for both fragments
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) wil_off[pl] = (wc * 32 + fr) * 128 + (((4 * pl + fq) ^ (fr & 7)) << 4);
    // ---- column borders, wave-uniform: the x = 0 pixels of this wave row's RW pixels are W apart (W >= 32: at most one per 16-pixel fragment),
    // the x = W - 1 pixels sit right before them.  pres0 / pres1: bit f = fragment f (= 4 mh + i) holds an x = 0 / x = W - 1 pixel;
    // pos0 / pos1: its row fr in 4 bits per fragment.  A lane's mask is one compare of fr with a scalar. ----
    unsigned pres0 = 0, pres1 = 0, pos0 = 0, pos1 = 0;
    {
        const int base = m0 + wr * RW;
        int j = (p.W - base % p.W) % p.W;  // first pixel of the wave row with x = 0
        for (; j <= RW; j += p.W) {        // (j = RW: only its left neighbour RW - 1 is ours)
            if (j < 16 * NF) pres0 |= 1u << (j >> 4), pos0 |= (unsigned)(j & 15) << ((j >> 4) * 4);
            if (j >= 1 && j - 1 < 16 * NF) pres1 |= 1u << ((j - 1) >> 4), pos1 |= (unsigned)((j - 1) & 15) << (((j - 1) >> 4) * 4);
        }
    }
    pres0 = __builtin_amdgcn_readfirstlane(pres0), pres1 = __builtin_amdgcn_readfirstlane(pres1);
    pos0 = __builtin_amdgcn_readfirstlane(pos0), pos1 = __builtin_amdgcn_readfirstlane(pos1);

    f32x4 acc[2][4][4];  // [mh][n][m]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[h][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[4][2], wa[2][2], wb[2][2];  // [frag][plane]

    auto read_x = [&](auto vpar_, auto kw_, auto mh_) {  // (parities compile-time: see stage_w)
        constexpr int KW = decltype(kw_)::value, MH = decltype(mh_)::value, VP = decltype(vpar_)::value & 1;
        const char* s = smem + VP * WIN + x_off[KW] + MH * 4096;
#pragma unroll
        for (int i = 0; i < (MH == 0 ? 4 : NF - 4); ++i) {
            xf[i][0] = *(const bf16x8*)(s + i * 1024);
            xf[i][1] = *(const bf16x8*)(s + XPLANE + i * 1024);
        }
    };
    auto read_w = [&](bf16x8 (&w)[2][2], auto tpar_, auto nh_) {
        constexpr int TP = decltype(tpar_)::value & 1, NH = decltype(nh_)::value;
        const char* s = smem + TP * WTILE + NH * WHALF;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (WIL) {
                w[j][0] = *(const bf16x8*)(s + wil_off[0] + j * 2048);
                w[j][1] = *(const bf16x8*)(s + wil_off[1] + j * 2048);
            } else {
                w[j][0] = *(const bf16x8*)(s + w_off + j * 1024);
                w[j][1] = *(const bf16x8*)(s + WPLANE + w_off + j * 1024);
            }
        }
    };
    auto mma = [&](auto mh_, auto nh_, auto kw_, bf16x8 (&w)[2][2], bool fresh_x) {
        constexpr int MH = decltype(mh_)::value, NH = decltype(nh_)::value, KW = decltype(kw_)::value;
        constexpr int NI = MH == 0 ? 4 : NF - 4;
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        if (fresh_x) {
            if constexpr (KW != 1) {  // column border of this tap: zero the fragment row that would read across a row end
                const unsigned pres = KW == 0 ? pres0 : pres1, pos = KW == 0 ? pos0 : pos1;
                if (pres != 0) {
                    const unsigned frv = lane_id_now() & 15u;  // (recomputed: a loop-invariant lane constant would be one more register to spill)
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const int f = MH * 4 + i;
                        if ((pres >> f) & 1u) {
                            const unsigned k = frv == ((pos >> (4 * f)) & 15u) ? 0u : 0xFFFFFFFFu;
                            const u32x4 keep = {k, k, k, k};
                            xf[i][0] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, xf[i][0]) & keep);
                            xf[i][1] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, xf[i][1]) & keep);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                f32x4& a = acc[MH][NH * 2 + j][i];
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][1], xf[i][0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][0], xf[i][1], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][0], xf[i][0], a, 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // end of the L slot of phase P (0..11) of super-tile u (window / first W buffer parity PAR): this phase's operations, the counted wait, the barrier
    auto l_end = [&](auto par_, auto p_, int u) {
        constexpr int P = decltype(p_)::value, PAR = decltype(par_)::value;
        constexpr int TP = PAR + P / 4;  // parity of K-tile t = 3 u + P / 4
        using S = Sched<X3>;
        if constexpr ((P & 3) == 0) stage_w(IC<1>{}, TP + 1);        // W-hi(t + 1)
        else if constexpr ((P & 3) == 2) stage_w(IC<0>{}, TP + 2);   // W-lo(t + 2)
        else if constexpr (P == 1) stage_x(IC<0>{}, IC<2>{}, PAR + 1, u + 1);
        else if constexpr (P == 3) stage_x(IC<1>{}, IC<2>{}, PAR + 1, u + 1);
        else if constexpr (P == 5 && X3) stage_x(IC<2>{}, IC<0>{}, PAR + 1, u + 1);
        else if constexpr (P == 7 && X3) stage_x(IC<2>{}, IC<1>{}, PAR + 1, u + 1);
        if constexpr ((P & 1) == 0) wait_vmcnt<S::younger_w(P)>();  // the W half-tile read in the next phase
        else if constexpr (P == 11) wait_vmcnt<S::younger_x()>();    // window u + 1
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // K-tile j (= kw) of super-tile u; wcur holds W-lo(t) on entry
    auto tile_body = [&](auto par_, auto j_, int u, bf16x8 (&wcur)[2][2], bf16x8 (&wnxt)[2][2]) {
        constexpr int J = decltype(j_)::value, PAR = decltype(par_)::value;
        constexpr int TP = PAR + J;
        read_x(par_, j_, IC<0>{});
        l_end(par_, IC<4 * J + 0>{}, u);
        mma(IC<0>{}, IC<0>{}, j_, wcur, true);
        read_w(wnxt, IC<TP>{}, IC<1>{});
        l_end(par_, IC<4 * J + 1>{}, u);
        mma(IC<0>{}, IC<1>{}, j_, wnxt, false);
        read_x(par_, j_, IC<1>{});
        l_end(par_, IC<4 * J + 2>{}, u);
        mma(IC<1>{}, IC<1>{}, j_, wnxt, true);
        read_w(wnxt, IC<TP + 1>{}, IC<0>{});  // W-lo of the next K-tile into the set W-hi(t) just vacated (behind the last K-tile: unused bytes)
        l_end(par_, IC<4 * J + 3>{}, u);
        mma(IC<1>{}, IC<0>{}, j_, wcur, false);
    };
    auto super_tile = [&](auto par_, int u, bf16x8 (&w0)[2][2], bf16x8 (&w1)[2][2]) {  // w0 holds W-lo(3 u) on entry, w1 W-lo(3 u + 3) on exit
        tile_body(par_, IC<0>{}, u, w0, w1);
        tile_body(par_, IC<1>{}, u, w1, w0);
        tile_body(par_, IC<2>{}, u, w0, w1);
    };

    // The loop body is a PAIR of super-tiles (buffer parities 0 then 1, the W register sets swapping roles).  An odd count (Cin / 32 odd) enters
    // the first pair at its second half: super-tile u then lives in the buffers of parity (u + q) & 1, q = nu & 1.
    const int q = nu & 1;
    // ---- prologue: window 0, then W-lo(0), W-hi(0), W-lo(1) (nt >= 9) ----
    stage_x(IC<0>{}, IC<2>{}, q, 0);
    stage_x(IC<1>{}, IC<2>{}, q, 0);
    if constexpr (X3) stage_x(IC<2>{}, IC<2>{}, q, 0);
    stage_w(IC<0>{}, q);
    stage_w(IC<1>{}, q);
    stage_w(IC<0>{}, q + 1);
    wait_vmcnt<4>();  // window 0 and W-lo(0) have landed; W-hi(0), W-lo(1) may be in flight
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (q) read_w(wa, IC<1>{}, IC<0>{});
    else read_w(wb, IC<0>{}, IC<0>{});
    if (wr == 1) {  // stagger: the wr = 1 group runs one slot behind
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (STAMP) stamps.t_prologue = gemm_stamp();

    int u = 0;
    bool skip = q != 0;
    while (u < nu) {
        if (!skip) {
            super_tile(IC<0>{}, u, wb, wa);
            ++u;
        }
        skip = false;
        super_tile(IC<1>{}, u, wa, wb);
        ++u;
    }

    if (wr == 0) __builtin_amdgcn_s_barrier();  // pairs with the last M-slot barrier of the wr = 1 group
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (STAMP) stamps.t_loop = gemm_stamp();
    conv_x3_epilogue<4, 4>(p, acc[0], smem + wave * 16384, m0 + wr * RW, n0 + wc * 64, lane, grp);
    conv_x3_epilogue<4, 4, NF - 4>(p, acc[1], smem + wave * 16384, m0 + wr * RW + 64, n0 + wc * 64, lane, grp);
    if constexpr (STAMP) stamps.finish(p.stamps, p.stamp_rows);
}

}  // namespace

// 3x3 / stride 1 / pad 1 / zero padding, Cout % 256 == 0, 32-bit operand offsets (the caller checks): tiles of 32 nf pixels x 256 cout
int ufm_launch_conv_x3_halo(const ConvX3Args& p, hipStream_t stream, int nf) {
    const int rows = 32 * nf;
    const int ntm = (p.M - p.m_begin + rows - 1) / rows * p.groups;
    const dim3 grid(ntm * (p.Cout / 256)), block(512);
    if (p.stamps && nf == 8) hipLaunchKernelGGL((conv_x3_halo_kernel<8, true>), grid, block, 0, stream, p);  // diagnostic build
    else if (p.w_il && nf == 5) hipLaunchKernelGGL((conv_x3_halo_kernel<5, false, true>), grid, block, 0, stream, p);
    else if (p.w_il && nf == 6) hipLaunchKernelGGL((conv_x3_halo_kernel<6, false, true>), grid, block, 0, stream, p);
    else if (p.w_il && nf == 7) hipLaunchKernelGGL((conv_x3_halo_kernel<7, false, true>), grid, block, 0, stream, p);
    else if (p.w_il) hipLaunchKernelGGL((conv_x3_halo_kernel<8, false, true>), grid, block, 0, stream, p);
    else if (nf == 5) hipLaunchKernelGGL((conv_x3_halo_kernel<5>), grid, block, 0, stream, p);
    else if (nf == 6) hipLaunchKernelGGL((conv_x3_halo_kernel<6>), grid, block, 0, stream, p);
    else if (nf == 7) hipLaunchKernelGGL((conv_x3_halo_kernel<7>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((conv_x3_halo_kernel<8>), grid, block, 0, stream, p);
    return 0;
}
