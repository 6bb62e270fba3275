// Shared pieces of the bf16x3 convolution kernels (conv_bf16x3.hip, conv_bf16x3_8ph.hip).
#pragma once
#include "common.h"
#include "lab_flags.h"

struct ConvX3Args {
    const uint16_t* in;   // [2][B*H*W][Cin]
    const uint16_t* w;    // [2][Cout][KH*KW*Cin]
    const float* bias;
    const uint16_t* res1;  // [2][M][Cout] or null
    const uint16_t* res2;
    const uint16_t* zero;
    uint16_t* out;         // [2][Mout][Co]
    uint16_t* out_relu;    // optional second output: relu(out), same layout (the next RCU conv's input; the raw `out` feeds its skip)
    long long in_plane, w_plane, out_plane;
    int B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, M;
    int relu_in, act, shuffle, Co;
    int m_begin;  // the launch covers output pixels [m_begin, M) (hybrid 8-phase + 128-row split of one conv)
    // Linear-layer epilogue of ufm_gemm_bf16x3 (numerics "precise": the transformer trunk on the split format)
    const float* gamma;    // per-output-channel scale after the activation (LayerScale / Q pre-scale) or null
    const float* res_f32;  // fp32 residual [M][Cout] added after gamma (may alias out_f32) or null
    float* out_f32;        // non-null: the result is stored as fp32 [M][Cout] instead of the split planes
    int replicate;         // padding_mode='replicate': out-of-range taps read the nearest edge pixel instead of the zero page
    // Grouped launch (ufm_conv2d_nhwc_bf16x3_grouped): `groups` independent convolutions of identical geometry in ONE grid --
    // the two DPT heads (ufm.py:553-556, 637-642: the same graph with different weights).  Group g reads the weights at
    // w + g * w_group, the bias at bias + g * (Co or Cout) and the input images at in + g * in_group (0: all groups share one
    // input); its output / residual rows follow group g - 1's: row g * Mg + m of out / res1 / res2 / out_relu.  M and m_begin
    // stay PER GROUP (rows [m_begin, M) of every group are covered by the launch).  groups = 1: Mg = rows of the one problem.
    int groups, Mg;
    long long in_group, w_group;
    // Deterministic split-K (conv_bf16x3.hip, the 128- / 64-row kernels): the K loop of a tile is cut into `splitk` consecutive
    // ranges of K-tiles, one workgroup each; every workgroup parks its fp32 partial tile in `slab`, the LAST one to arrive (an
    // agent-scope arrival counter per tile) adds the partials IN SLICE ORDER and runs the epilogue.  splitk depends on the layer's
    // geometry only (never on the batch), so a pixel's result is the same bits at any batch size.  splitk = 1: off.
    int splitk;
    float* slab;          // [tiles][splitk][BM * BN] partial tiles (fragment order)
    unsigned* counters;   // [tiles], zero between launches (the last arriver resets its tile's word)
    const uint16_t* w_il; // round 6: the same weights INTERLEAVED, [groups][Cout][KH KW Cin / 32][hi 32 | lo 32] (or null): read by the halo kernel in place of `w`
    int out_il;           // round 6: the split output is stored INTERLEAVED, [row][Cout / 32][hi 32 | lo 32] (UFM_BF16X2_IL; no residual, no out_relu, no shuffle)
    int il;               // round 6, ufm_gemm_bf16x3_il: `in` and `w` are INTERLEAVED split operands [rows][K / 32][hi 32 | lo 32] (the 8-phase Linear form only)
    int serial_epilogue;  // A/B hook (ufm_debug_set_conv_variant bit 4): the per-pass residual read-out of rounds 1-4
    // diagnostic build only (ufm_debug_set_conv_stamps; the STAMP = true instantiation of the 8-phase kernel): 8 x uint64 per workgroup
    unsigned long long* stamps;
    int stamp_rows;
    int ablate;  // diagnostic build only (ufm_debug_set_conv_variant bits 12..17): 1 no LDS-DMA, 2 no fragment reads, 4 no MFMA, 8 no slot barriers -- TIMING ONLY, results wrong; 16 nothing removed; 32 one rendezvous per phase (timing only: not a legal schedule); 64 the X pieces of one tap in nine (the DMA count of a halo tile)
};

// tile row index over all groups -> (group, tile row inside the group); tiles_pg = row tiles per group of this launch
static __device__ __forceinline__ void conv_x3_group_of(const ConvX3Args& p, int tmi_all, int tiles_pg, int& g, int& tmi) {
    g = 0, tmi = tmi_all;
    if (p.groups > 1) g = tmi_all / tiles_pg, tmi = tmi_all - g * tiles_pg;
}

static __device__ __forceinline__ int swz(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }

static __device__ __forceinline__ void split_store4(uint16_t* hi_ptr, long long plane, const f32x4& v) {
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

static __device__ __forceinline__ f32x4 split_load4(const uint16_t* hi_ptr, long long plane) {
    const u32x2 ph = *(const u32x2*)hi_ptr;
    const u32x2 pl = *(const u32x2*)(hi_ptr + plane);
    f32x4 v;
    v[0] = __uint_as_float(ph[0] << 16) + __uint_as_float(pl[0] << 16);
    v[1] = __uint_as_float(ph[0] & 0xffff0000u) + __uint_as_float(pl[0] & 0xffff0000u);
    v[2] = __uint_as_float(ph[1] << 16) + __uint_as_float(pl[1] << 16);
    v[3] = __uint_as_float(ph[1] & 0xffff0000u) + __uint_as_float(pl[1] & 0xffff0000u);
    return v;
}


// Epilogue of one wave's (TM*16) x (TN*16) fp32 tile, staged through `ws` (TM*16 rows x TN*64 B of LDS owned by the
// wave).  The direct form touched 16 pixel rows x 8 B per wave instruction for each of: hi store, lo store, and up to
// four residual plane loads; staged, a wave instruction covers whole pixel rows (TN*16 consecutive output channels =
// 128/64 B per plane, contiguous).  Chunk index XOR row keeps both the accumulator-shaped writes and the row-shaped
// reads conflict-free.  LDS ops of one wave execute in order: no barrier between the writes and the reads.
// TMU <= TM: only the first TMU 16-row fragments of the accumulator array exist (8-phase tiles lower than 256 rows).
// ReLU on the bit pattern (negative floats are negative integers): NaN stays NaN (fmaxf(NaN, 0) = 0), -0.0 -> +0.0
static __device__ __forceinline__ float relu_bits(float v) { return __int_as_float(max(__float_as_int(v), 0)); }

template <int TM, int TN, int TMU = TM>
static __device__ __forceinline__ void conv_x3_epilogue(const ConvX3Args& p, f32x4 (&acc)[TN][TM], char* ws, int pix0, int cb0, int lane, int g = 0) {
    static_assert(TMU >= 1 && TMU <= TM, "TMU");
    // pix0 = first row of the tile INSIDE group g; rows of out / res* are numbered over all groups (row0 = g * Mg)
    const float* bias_g = p.bias ? p.bias + (size_t)g * (p.shuffle ? p.Co : p.Cout) : nullptr;
    const int row0 = g * p.Mg;
    constexpr int ROWB = TN * 64;              // bytes per staged row (fp32)
    constexpr int NCH = TN * 4;                // 16-byte chunks per row
    constexpr int LPR = NCH;                   // lanes per row on the way out
    constexpr int RPI = 64 / LPR;              // rows per wave instruction
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int n = 0; n < TN; ++n) {
        const int cb = cb0 + n * 16 + fq * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias_g) bv = *(const f32x4*)(bias_g + (p.shuffle ? cb % p.Co : cb));
#pragma unroll
        for (int m = 0; m < TMU; ++m) {
            const int r = m * 16 + fr;
            *(f32x4*)(ws + r * ROWB + (((n * 4 + fq) ^ (r & (NCH - 1))) << 4)) = acc[n][m] + bv;
        }
    }
    const int orr = lane / LPR, oc = lane % LPR;
    // Round 5: residual epilogues with the loads taken OUT of the store chain.  The general loop below compiles to {residual
    // loads, s_waitcnt vmcnt(0), stores} per pass -- `continue` / run-time switches keep hipcc from moving a load above the
    // previous pass's stores, and a vmcnt(0) also waits for those STORES (vmcnt counts them): 16 serial memory round trips per
    // 64-row slice.  For whole slices (no ragged rows, no pixel shuffle) of the three residual forms -- fp32 read-modify-write
    // (ufm_gemm_bf16x3's proj / fc2), one or two split residuals (the RCU / fusion-block convolutions) -- the loads of a group of
    // passes are issued first, then the group's arithmetic and stores.  Same operations in the same order per element: bit-identical.
    constexpr int PASSES = TMU * 16 / RPI;
    // mode 4 (round 5, second step): split output WITHOUT a residual -- most layers of the heads and the precise-mode QKV / fc1.  No loads
    // to pipeline there, but the general loop's per-pass `continue` and run-time switches still cost: the stamps show 44 k cycles of
    // epilogue per 256 x 256 tile for a store-only layer against 28 k for the grouped residual form (tools/lab/conv_stamps.py).
    const bool gelu = p.act == UFM_ACT_GELU;
    const int mode = (p.shuffle || pix0 + TMU * 16 > p.M || (gelu && (p.out_f32 || p.res1))) ? 0
                     : (p.out_f32 ? (p.res_f32 ? 3 : 0) : (p.res1 ? (p.res2 ? 2 : 1) : 4));
    if (mode != 0 && !p.serial_epilogue) {
        const int cb = cb0 + oc * 4;
        // ReLU / no activation as ONE signed-integer max on the bit pattern (negative floats are negative integers): floor 0 = ReLU, floor
        // INT_MIN = identity.  Round 6: the float form fmaxf(v, -inf) turned a NaN accumulator into -inf (fmaxf returns its
        // non-NaN operand) where the serial path kept it; the integer form is the identity on every bit pattern, NaN included.
        const int lo = (p.act == UFM_ACT_RELU) ? 0 : (int)0x80000000;
        f32x4 gv = {1.f, 1.f, 1.f, 1.f};
        __builtin_amdgcn_s_waitcnt(0x0F70);  // compiler-visible vmcnt(0): clears the K loop's LDS-DMA from hipcc's scoreboard (all landed)
        if (p.gamma) gv = *(const f32x4*)(p.gamma + cb);
        auto body = [&](auto mode_c, auto gelu_c) {
            constexpr int MODE = decltype(mode_c)::value;
            constexpr bool GELU = decltype(gelu_c)::value;
            constexpr int G = MODE == 2 ? (PASSES < 8 ? PASSES : 8) : PASSES;  // passes per group: <= 64 registers of loads in flight
#pragma unroll
            for (int g0 = 0; g0 < PASSES; g0 += G) {
                f32x4 rf[G];
                u32x2 h1[G], l1[G], h2[MODE == 2 ? G : 1], l2[MODE == 2 ? G : 1];
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    if (g0 + i >= PASSES) continue;  // (PASSES = 12 with groups of 8: the lower 8-phase tiles)
                    const size_t o = (size_t)(row0 + pix0 + (g0 + i) * RPI + orr) * p.Cout + cb;
                    if constexpr (MODE == 3) rf[i] = *(const f32x4*)(p.res_f32 + o);
                    if constexpr (MODE == 1 || MODE == 2) h1[i] = *(const u32x2*)(p.res1 + o), l1[i] = *(const u32x2*)(p.res1 + o + p.out_plane);
                    if constexpr (MODE == 2) h2[i] = *(const u32x2*)(p.res2 + o), l2[i] = *(const u32x2*)(p.res2 + o + p.out_plane);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    if (g0 + i >= PASSES) continue;
                    const int r = (g0 + i) * RPI + orr;
                    const size_t o = (size_t)(row0 + pix0 + r) * p.Cout + cb;
                    f32x4 v = *(const f32x4*)(ws + r * ROWB + ((oc ^ (r & (NCH - 1))) << 4));
                    if constexpr (GELU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = gelu_erf_fast(v[j]);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = __int_as_float(max(__float_as_int(v[j]), lo));
                    }
                    v *= gv;
                    auto unsplit = [](const u32x2& ph, const u32x2& pl) {
                        f32x4 x;
                        x[0] = __uint_as_float(ph[0] << 16) + __uint_as_float(pl[0] << 16);
                        x[1] = __uint_as_float(ph[0] & 0xffff0000u) + __uint_as_float(pl[0] & 0xffff0000u);
                        x[2] = __uint_as_float(ph[1] << 16) + __uint_as_float(pl[1] << 16);
                        x[3] = __uint_as_float(ph[1] & 0xffff0000u) + __uint_as_float(pl[1] & 0xffff0000u);
                        return x;
                    };
                    if constexpr (MODE == 3) {
                        v += rf[i];
                        *(f32x4*)(p.out_f32 + o) = v;
                    } else {
                        if constexpr (MODE == 1 || MODE == 2) v += unsplit(h1[i], l1[i]);
                        if constexpr (MODE == 2) v += unsplit(h2[i], l2[i]);
                        if (MODE == 4 && p.out_il) split_store4(p.out + ((size_t)(row0 + pix0 + r) * (2 * p.Cout) + ((cb >> 5) << 6) + (cb & 31)), 32, v);
                        else split_store4(p.out + o, p.out_plane, v);
                        if (p.out_relu) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = relu_bits(v[j]);
                            split_store4(p.out_relu + o, p.out_plane, v);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        if (mode == 3) body(std::integral_constant<int, 3>{}, F_{});
        else if (mode == 2) body(std::integral_constant<int, 2>{}, F_{});
        else if (mode == 1) body(std::integral_constant<int, 1>{}, F_{});
        else if (gelu) body(std::integral_constant<int, 4>{}, T_{});
        else body(std::integral_constant<int, 4>{}, F_{});
        return;
    }
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        const int r = pass * RPI + orr;
        f32x4 v = *(const f32x4*)(ws + r * ROWB + ((oc ^ (r & (NCH - 1))) << 4));
        if (pix0 + r >= p.M) continue;
        const int pix = row0 + pix0 + r;
        const int cb = cb0 + oc * 4;
        if (p.shuffle) {
            const int sx = pix % p.Wo, t = pix / p.Wo, sy = t % p.Ho, sb = t / p.Ho;
            const int tapo = cb / p.Co, co = cb - tapo * p.Co;
            const int kh = tapo / p.shuffle, kw = tapo - kh * p.shuffle;
            const size_t o = (((size_t)sb * (p.Ho * p.shuffle) + sy * p.shuffle + kh) * (p.Wo * p.shuffle) + sx * p.shuffle + kw) * p.Co + co;
            split_store4(p.out + o, p.out_plane, v);
            continue;
        }
        if (p.act == UFM_ACT_GELU) {
            // branch-free erf (|abs err| <= 1.5e-7: below the 2^-17 relative resolution of the split format the value is
            // stored in); libm's branchy erff made the GELU epilogue of the precise-mode fc1 GEMM 15 % of its launch
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = gelu_erf_fast(v[j]);
        } else if (p.act == UFM_ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = relu_bits(v[j]);
        }
        const size_t o = (size_t)pix * p.Cout + cb;
        if (p.gamma) v *= *(const f32x4*)(p.gamma + cb);
        if (p.out_f32) {  // fp32 residual stream, read-modify-write in place
            if (p.res_f32) v += *(const f32x4*)(p.res_f32 + o);
            *(f32x4*)(p.out_f32 + o) = v;
            continue;
        }
        if (p.res1) v += split_load4(p.res1 + o, p.out_plane);
        if (p.res2) v += split_load4(p.res2 + o, p.out_plane);
        if (p.out_il) split_store4(p.out + ((size_t)pix * (2 * p.Cout) + ((cb >> 5) << 6) + (cb & 31)), 32, v);
        else split_store4(p.out + o, p.out_plane, v);
        if (p.out_relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = relu_bits(v[j]);
            split_store4(p.out_relu + o, p.out_plane, v);
        }
    }
}

// conv_bf16x3_8ph.hip: 8-phase kernels, 256 px x 256 cout (Cout % 256 == 0) or 512 px x 128 cout (Cout % 128 == 0); 32-bit operand offsets
// conv_bf16x3_pair.hip: 256 px x 128 cout, four waves, two resident workgroups per CU (Cout % 128 == 0, at least 2 K-tiles, 32-bit operand offsets)
int ufm_launch_conv_x3_pair(const ConvX3Args& p, hipStream_t stream);
int ufm_launch_gemm_x3_il_8ph(const ConvX3Args& p, hipStream_t stream, int nf = 8);  // conv_bf16x3_8ph.hip: the Linear form on interleaved operands
// conv_bf16x3_halo.hip (round 6): the 8-phase 256-cout kernel for 3x3 / stride 1 / pad 1 / zero padding with the input staged once per filter
// row (a row-window halo tile) instead of once per tap; same tile heights (32 nf pixels), bit-identical
int ufm_launch_conv_x3_halo(const ConvX3Args& p, hipStream_t stream, int nf = 8);
int ufm_launch_conv_x3_8ph(const ConvX3Args& p, hipStream_t stream, int nf = 8);  // nf = 16-row fragments per wave row (5..8): tiles of 32 nf pixels
