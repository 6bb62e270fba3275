// Split-precision (numerics "precise") flash attention, head_dim 64 -- round 5 rebuild of attention_bf16x3.hip.
//
// Same tiling as the round-1 kernel (workgroup = 4 waves x 32 query rows, two workgroups per CU = two waves per SIMD, 64-key tiles,
// swapped QK^T so one lane owns one query column, P in registers, V^T by ds_read_b64_tr_b16) with the two things that kept it at
// 0.37 of the / 3 peak removed:
//   * K / V (hi and lo planes) come in by LDS-DMA (global_load_lds_dwordx4, swizzle on the source address) one tile ahead:
//     no register staging (16 VGPRs, 8 global loads + 8 ds_write_b128 per thread and tile, a vmcnt(0) in front of the writes) and
//     one raw s_barrier per tile instead of __syncthreads;
//   * the wave's three phases no longer run in sequence.  QK^T of tile t + 1 (24 MFMA: lo*hi, hi*lo, hi*hi per fragment pair) is
//     issued INTERLEAVED with the softmax of tile t (running maximum, 32 exp2, row sum, the (hi, lo) split of P: ~1100 VALU issue
//     cycles against 768 matrix-pipe cycles), in one basic block shaped by sched_group_barrier; P.V of tile t (24 MFMA) follows.
//     Before, a wave's matrix pipe idled through its own softmax and its VALU through its own MFMAs, and the overlap was left to
//     the second resident workgroup, which runs the same program in the same phase.
//
//   top of iteration t:  wait for the own DMA pieces of K(t + 1) and V(t) (issued a whole iteration ago: vmcnt(0) is free), s_barrier,
//                        issue K(t + 2) -> K stage t & 1 (last read: iteration t - 1) and V(t + 1) -> V stage (t + 1) & 1 (ditto)
//   slot X:              S(t + 1) = K(t + 1) . Q^T      ||  max / alpha, P(t) = exp2(c S(t) - c m), row sum, split
//   (rescale O while a row's maximum moved)
//   slot Y:              O += V(t)^T . P(t)^T
//
// Arithmetic: per accumulator exactly the MFMA sequence, softmax operations and accumulation order of attention_bf16x3.hip, so the
// two kernels agree BIT FOR BIT (tests/test_kernels_gpu.py); the round-1 kernel stays as the reference (ufm_debug_set_attn_variant 2).
#include "common.h"

namespace {

// NW = waves per workgroup: 4 (128 query rows, two workgroups per CU) or 8 (256 rows, one workgroup per CU: a K / V tile is staged
// once for eight waves -- half the LDS-DMA instructions per wave and tile -- at the price of an eight-wave barrier)
constexpr int KB = 64;         // keys per tile
constexpr int PLANE = 8192;    // one 64-key x 64-d bf16 tile
constexpr int STG = 2 * PLANE; // a K or V stage: [hi][lo]
constexpr int V_RING = 2 * STG;
constexpr int X3PW_LDS = 4 * STG;  // K stages 0 / 1, V stages 0 / 1: 64 KiB
constexpr float NEG_BIG = -1.0e30f;

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;
__device__ __forceinline__ bf16x4 tr_read(const char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_ptr)LDS_PTR(p)); }

template <int K>
using IC = std::integral_constant<int, K>;

// LDS-DMA as inline asm (as attention_bf16_pw.hip): with the builtin hipcc keeps the transfer on its own scoreboard and puts an
// s_waitcnt vmcnt(0) in front of the next ds_read of the iteration -- i.e. waits for the tile it has just requested, every tile.
// M0 = LDS byte address of the 1-KiB piece (wave uniform), written in the statement that uses it; nothing else in this kernel uses
// M0 (tools/check_attn_isa.py --x3 audits the assembly for that).  16 bytes per lane: LDS[m0 + 16 lane] = gbase[voff .. voff + 15].
__device__ __forceinline__ void glds16(const char* gbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(gbase) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const char* p) { return (unsigned)(uintptr_t)LDS_PTR(p); }

// FIXREF (round 6): the softmax reference of a query row is FIXED after the first key tile (its maximum there) instead of following the running
// maximum: -m_ref rides in as the C operand of every tile's first QK^T MFMA, P = exp2(c (S - m_ref)) needs no per-tile maximum (32 fmax, the
// cross-half exchange, alpha) and O no per-tile rescale; only if a tile's weights pass 2^54 (keys scoring far above the first tile's) the reference is moved up
// to that tile's maximum in a cold, wave-uniform path in front of P.V (any jump is handled: the tile's softmax is redone).  The bf16 kernel's scheme (attention_bf16_pw.hip); PMC put this kernel at 7.1 VALU instructions per MFMA,
// issue-bound -- this takes 1.2 of them out.  The same softmax in exact arithmetic, other roundings: NOT bitwise the round-1 kernel
// (FIXREF = false stays that, ufm_debug_set_attn_variant bit 1 or the ufm_attention_bf16x3 test hook selects it); tested against fp64.
// PRESC (with FIXREF): q arrives pre-scaled by softmax_scale * log2(e) (the QKV Linear's epilogue multiplies the Q columns before the (hi, lo)
// split, as the bf16 path does): P = exp2(S') with no multiply per score -- one more VALU instruction out of the issue-bound gap.
template <int NW, bool FIXREF = false, bool PRESC = false>
__global__ __launch_bounds__(NW * 64, 8 / NW) void attn_x3_pw_kernel(const uint16_t* __restrict__ qp_, int ldq, long long q_plane,
                                                            const uint16_t* __restrict__ kp_, const uint16_t* __restrict__ vp_, int ldkv, long long in_plane,
                                                            uint16_t* __restrict__ out, int ldo, long long out_plane, int Nq, int N, int H, float c, int out_il) {
    __shared__ __attribute__((aligned(16))) char smem[X3PW_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int QB = NW * 32;  // query rows per workgroup
    const int nqb = (Nq + QB - 1) / QB;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);  // all query blocks of one (image, head) share an XCD's L2
    const int qblk = lid % nqb, head = (lid / nqb) % H, b = lid / (nqb * H);
    const uint16_t* qbase = qp_ + (size_t)b * Nq * ldq + head * 64;
    const char* kp = (const char*)(kp_ + (size_t)b * N * ldkv + head * 64);
    const char* vp = (const char*)(vp_ + (size_t)b * N * ldkv + head * 64);
    const int ql_ = lane & 31, hh = lane >> 5;
    const int q = qblk * QB + wave * 32 + ql_;

    bf16x8 qh[4], qlo[4];
    {
        const uint16_t* qr = qbase + (size_t)min(q, Nq - 1) * ldq + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qh[s] = *(const bf16x8*)(qr + 16 * s);
            qlo[s] = *(const bf16x8*)(qr + q_plane + 16 * s);
        }
    }

    // ---- LDS-DMA staging: piece = 8 rows x 128 B; wave w issues pieces w, w + NW, .. of both planes of a K or V tile ----
    const int srow = lane >> 3, slot = lane & 7;
    const long long lo_bytes = 2 * in_plane;
    auto stage = [&](auto is_v, int t) {  // K(t) -> K stage t & 1, V(t) -> V stage t & 1
        constexpr bool ISV = decltype(is_v)::value != 0;
        char* d = smem + (ISV ? V_RING : 0) + (t & 1) * STG + wave * 1024;
        const char* src = ISV ? vp : kp;
#pragma unroll
        for (int i = 0; i < 8 / NW; ++i) {
            const int r = (wave + NW * i) * 8 + srow;                                          // row of the tile
            const unsigned row_off = (unsigned)min(t * KB + r, N - 1) * (unsigned)ldkv * 2u;   // clamp: masked below
            const unsigned o = row_off + ((slot ^ (ISV ? (((r >> 1) & 1) << 2) : ((r >> 1) & 7))) << 4);  // the chunk that must land in this lane's LDS slot
            glds16(src, o, lds_addr_of(d + i * NW * 1024));
            glds16(src + lo_bytes, o, lds_addr_of(d + PLANE + i * NW * 1024));
        }
    };

    // ---- fragment read offsets (the images of attention_bf16x3.hip) ----
    int k_off[2][4];  // [key half][k-step]
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int key = kt * 32 + ql_;
            k_off[kt][s] = key * 128 + (((2 * s + hh) ^ ((key >> 1) & 7)) << 4);
        }
    const int ti = lane & 15, tq = ti >> 2, tp = ti & 3;
    const int tdc = 16 * ((lane >> 4) & 1) + 4 * tp;
    int v_off[2][2][2][2];  // [dt][kt][s2][half of the fragment]
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int row = kt * 32 + 16 * s2 + 4 * hh + 8 * e + tq;
                    const int dcol = dt * 32 + tdc;
                    v_off[dt][kt][s2][e] = row * 128 + ((((dcol >> 3)) ^ (((row >> 1) & 1) << 2)) << 4) + ((dcol & 7) << 1);
                }

    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;
    bf16x8 ph[2][2], pl[2][2];  // P^T fragments (B operand of P.V), hi / lo
    float alpha = 1.f;
    bool moved = false;

    auto k_frag = [&](bf16x8& fh, bf16x8& fl, const char* ks, int kt, int s_) {
        fh = *(const bf16x8*)(ks + k_off[kt][s_]);
        fl = *(const bf16x8*)(ks + PLANE + k_off[kt][s_]);
    };
    auto qk_plain = [&](f32x16 (&st)[2], const char* ks) {  // S^T = K . Q^T, small terms first (the un-overlapped first tile)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                bf16x8 kfh, kfl;
                k_frag(kfh, kfl, ks, kt, s_);
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfl, qh[s_], st[kt], 0, 0, 0);
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, qlo[s_], st[kt], 0, 0, 0);
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, qh[s_], st[kt], 0, 0, 0);
            }
        }
    };
    auto mask_tail = [&](f32x16 (&st)[2], int t) {  // ragged last tile (block-uniform branch at the call site)
        asm volatile("" ::: "memory");  // keeps the branch a branch: if-converted, its 32 compare / select pairs would run on every tile
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * KB + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (key >= N) st[kt][r] = NEG_BIG;
            }
    };

    // The online softmax of one tile (per lane = per query column), the round-1 kernel's operations in its order, cut into 24 pieces
    // that go into the 24 MFMA gaps of slot X:  pieces 0..7 = the running maximum (four fmax each; piece 7 also the cross-half
    // exchange, alpha and the exponent offset),  pieces 8..23 = one pair of scores each: exp2, row sum, (hi, lo) split.
    float mloc = 0.f, mc = 0.f, lsum = 0.f;
    unsigned pkh[8], pkl[8];
    f32x16 negm;  // FIXREF: -m_ref in every element (the C operand of a tile's first QK^T MFMA)
    float l_prev = 0.f;  // FIXREF: the row sum in front of the current tile (the cold re-reference path restarts from it)
    auto softmax_piece = [&](f32x16 (&st)[2], auto j_) {
        constexpr int J = decltype(j_)::value;
        if constexpr (FIXREF) {  // 16 score pairs over the 24 gaps (pair P in gap 3 P / 2): scale, exp2, row sum, (hi, lo) split
            if constexpr (J % 3 != 2) {
                constexpr int P = (J / 3) * 2 + (J % 3), kt = P >> 3, r = 2 * (P & 7);
                if constexpr (P == 0) lsum = 0.f;
                const float p0 = __builtin_amdgcn_exp2f(PRESC ? st[kt][r] : st[kt][r] * c);
                const float p1 = __builtin_amdgcn_exp2f(PRESC ? st[kt][r + 1] : st[kt][r + 1] * c);
                lsum += p0 + p1;
                const unsigned h = pack_bf16x2(p0, p1);
                pkh[r >> 1] = h;
                pkl[r >> 1] = pack_bf16x2(p0 - __uint_as_float(h << 16), p1 - __uint_as_float(h & 0xffff0000u));
                if constexpr ((P & 7) == 7) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        u32x4 wh = {pkh[4 * s2], pkh[4 * s2 + 1], pkh[4 * s2 + 2], pkh[4 * s2 + 3]};
                        u32x4 wl = {pkl[4 * s2], pkl[4 * s2 + 1], pkl[4 * s2 + 2], pkl[4 * s2 + 3]};
                        ph[kt][s2] = __builtin_bit_cast(bf16x8, wh);
                        pl[kt][s2] = __builtin_bit_cast(bf16x8, wl);
                    }
                }
                if constexpr (P == 15) l_prev = l_run, l_run += lsum;
                asm volatile("" : "+v"(pkh[r >> 1]), "+v"(pkl[r >> 1]), "+v"(lsum));
            }
        } else if constexpr (J < 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = 4 * J + i;
                const float v = st[idx >> 4][idx & 15];
                mloc = (J == 0 && i == 0) ? v : fmaxf(mloc, v);
            }
            if constexpr (J == 7) {
                mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
                const float m_new = fmaxf(m_run, mloc);
                alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
                moved = m_new != m_run;
                m_run = m_new;
                mc = m_new * c;
                lsum = 0.f;
            }
        } else {
            constexpr int P = J - 8, kt = P >> 3, r = 2 * (P & 7);
            const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][r], c, -mc));
            const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][r + 1], c, -mc));
            lsum += p0 + p1;
            const unsigned h = pack_bf16x2(p0, p1);
            pkh[r >> 1] = h;
            pkl[r >> 1] = pack_bf16x2(p0 - __uint_as_float(h << 16), p1 - __uint_as_float(h & 0xffff0000u));
            if constexpr ((P & 7) == 7) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4 wh = {pkh[4 * s2], pkh[4 * s2 + 1], pkh[4 * s2 + 2], pkh[4 * s2 + 3]};
                    u32x4 wl = {pkl[4 * s2], pkl[4 * s2 + 1], pkl[4 * s2 + 2], pkl[4 * s2 + 3]};
                    ph[kt][s2] = __builtin_bit_cast(bf16x8, wh);
                    pl[kt][s2] = __builtin_bit_cast(bf16x8, wl);
                }
            }
            if constexpr (P == 15) l_run = l_run * alpha + lsum;
            // produced HERE: without a pin LLVM sinks the whole piece to its first use (slot Y), out of the MFMA gap it is meant to fill
            asm volatile("" : "+v"(pkh[r >> 1]), "+v"(pkl[r >> 1]), "+v"(lsum));
        }
        if constexpr (!FIXREF && J < 8) asm volatile("" : "+v"(mloc));
    };
    // slot X: S(t + 1) = K(t + 1) . Q^T (eight fragment pairs x three MFMAs, the next pair's LDS reads issued a pair ahead) with one
    // piece of softmax(S(t)) behind every MFMA; sched_barrier pins each piece into its gap (left alone, hipcc ran the MFMAs in two
    // bursts and put two thirds of the softmax behind them)
    auto slot_x = [&](f32x16 (&cur)[2], f32x16 (&nxt)[2], const char* ks) {
        bf16x8 fh[2], fl[2];
        k_frag(fh[0], fl[0], ks, 0, 0);
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // the inline constant 0 as C: no 32 v_mov per tile
        auto group = [&](auto g_) {
            constexpr int G = decltype(g_)::value, kt = G >> 2, s_ = G & 3, b_ = G & 1;
            if constexpr (G + 1 < 8) k_frag(fh[b_ ^ 1], fl[b_ ^ 1], ks, (G + 1) >> 2, (G + 1) & 3);
            nxt[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[b_], qh[s_], s_ == 0 ? (FIXREF ? negm : zero) : nxt[kt], 0, 0, 0);
            softmax_piece(cur, IC<3 * G>{});
            __builtin_amdgcn_sched_barrier(0);
            nxt[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[b_], qlo[s_], nxt[kt], 0, 0, 0);
            softmax_piece(cur, IC<3 * G + 1>{});
            __builtin_amdgcn_sched_barrier(0);
            nxt[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[b_], qh[s_], nxt[kt], 0, 0, 0);
            softmax_piece(cur, IC<3 * G + 2>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        group(IC<0>{}); group(IC<1>{}); group(IC<2>{}); group(IC<3>{});
        group(IC<4>{}); group(IC<5>{}); group(IC<6>{}); group(IC<7>{});
    };
    auto softmax_alone = [&](f32x16 (&cur)[2]) {  // the last tile: nothing left to overlap with
        softmax_piece(cur, IC<0>{}); softmax_piece(cur, IC<1>{}); softmax_piece(cur, IC<2>{}); softmax_piece(cur, IC<3>{});
        softmax_piece(cur, IC<4>{}); softmax_piece(cur, IC<5>{}); softmax_piece(cur, IC<6>{}); softmax_piece(cur, IC<7>{});
        softmax_piece(cur, IC<8>{}); softmax_piece(cur, IC<9>{}); softmax_piece(cur, IC<10>{}); softmax_piece(cur, IC<11>{});
        softmax_piece(cur, IC<12>{}); softmax_piece(cur, IC<13>{}); softmax_piece(cur, IC<14>{}); softmax_piece(cur, IC<15>{});
        softmax_piece(cur, IC<16>{}); softmax_piece(cur, IC<17>{}); softmax_piece(cur, IC<18>{}); softmax_piece(cur, IC<19>{});
        softmax_piece(cur, IC<20>{}); softmax_piece(cur, IC<21>{}); softmax_piece(cur, IC<22>{}); softmax_piece(cur, IC<23>{});
    };
    // slot Y: O^T += V^T . P^T, small terms first; the next fragment pair's transposed reads a pair ahead
    auto slot_y = [&](const char* vs) {
        bf16x4 a0[2], a1[2], b0[2], b1[2];
        auto v_frag = [&](int buf, int dt, int kt, int s2) {
            a0[buf] = tr_read(vs + v_off[dt][kt][s2][0]);
            a1[buf] = tr_read(vs + v_off[dt][kt][s2][1]);
            b0[buf] = tr_read(vs + PLANE + v_off[dt][kt][s2][0]);
            b1[buf] = tr_read(vs + PLANE + v_off[dt][kt][s2][1]);
        };
        v_frag(0, 0, 0, 0);
        auto group = [&](auto g_) {
            constexpr int G = decltype(g_)::value, dt = G >> 2, kt = (G >> 1) & 1, s2 = G & 1, b_ = G & 1;
            if constexpr (G + 1 < 8) v_frag(b_ ^ 1, (G + 1) >> 2, ((G + 1) >> 1) & 1, (G + 1) & 1);
            const bf16x8 vh = {a0[b_][0], a0[b_][1], a0[b_][2], a0[b_][3], a1[b_][0], a1[b_][1], a1[b_][2], a1[b_][3]};
            const bf16x8 vl = {b0[b_][0], b0[b_][1], b0[b_][2], b0[b_][3], b1[b_][0], b1[b_][1], b1[b_][2], b1[b_][3]};
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph[kt][s2], oacc[dt], 0, 0, 0);
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl[kt][s2], oacc[dt], 0, 0, 0);
            oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph[kt][s2], oacc[dt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        group(IC<0>{}); group(IC<1>{}); group(IC<2>{}); group(IC<3>{});
        group(IC<4>{}); group(IC<5>{}); group(IC<6>{}); group(IC<7>{});
    };

    const int nt = (N + KB - 1) / KB;
    // iteration t: S(t) in `cur`, S(t + 1) built into `nxt`
    auto iteration = [&](int t, f32x16 (&cur)[2], f32x16 (&nxt)[2]) {
        // own DMA pieces of K(t + 1) and V(t), issued one iteration ago, have landed; behind the barrier everyone's have, and every
        // wave is done reading K(t) (slot X of iteration t - 1) and V(t - 1) (slot Y of iteration t - 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (t + 2 < nt) stage(IC<0>{}, t + 2);
        if (t + 1 < nt) stage(IC<1>{}, t + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nt) {
            slot_x(cur, nxt, smem + ((t + 1) & 1) * STG);
        } else {
            if ((t + 1) * KB > N) mask_tail(cur, t);  // only the last tile can be ragged, and its softmax runs here, outside the hot loop
            softmax_alone(cur);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FIXREF) {
            // cold path (wave-uniform): this tile holds keys scoring far above the row's reference -- a weight beyond 2^54, or already inf / NaN.
            // Nothing of the tile has reached O yet and S'(t) is still in `cur`: move the reference up to the tile's own maximum (rows that stay
            // below theirs keep it), scale O and the row sum down accordingly (they may underflow to 0: then they ARE negligible), correct the
            // scores slot X has already built for tile t + 1, and redo this tile's softmax.  Any jump is handled, at the price of one compare
            // and one ballot per tile in the hot path.
            if (__any(!(lsum <= 1.152921504606847e18f))) {
                float tm = cur[0][0];
#pragma unroll
                for (int idx = 1; idx < 32; ++idx) tm = fmaxf(tm, cur[idx >> 4][idx & 15]);
                tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
                const float dm = fmaxf(tm, 0.0f);
                const float f = __builtin_amdgcn_exp2f(-dm * c);
                l_run = l_prev * f;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= f;
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[r] -= dm;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) cur[kt][r] -= dm, nxt[kt][r] -= dm;
                softmax_alone(cur);
            }
        }
        if constexpr (!FIXREF) {
            if (__any(moved)) {  // wave-uniform: after the first tiles the running maxima rarely move (alpha == 1 exactly otherwise)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        slot_y(smem + V_RING + (t & 1) * STG);
    };

    f32x16 sa[2], sb[2];
    stage(IC<0>{}, 0);
    stage(IC<1>{}, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);  // the same wait, visible to hipcc: the Q loads are done -- no compiler vmcnt wait from here on (audited)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (nt > 1) stage(IC<0>{}, 1);
    qk_plain(sa, smem);
    if constexpr (FIXREF) {  // the reference: this row's maximum over the first key tile (both halves of the column), then S(0) -= m_ref
        if (nt == 1 && KB > N) mask_tail(sa, 0);
        float m0 = sa[0][0];
#pragma unroll
        for (int idx = 1; idx < 32; ++idx) m0 = fmaxf(m0, sa[idx >> 4][idx & 15]);
        m0 = fmaxf(m0, __shfl_xor(m0, 32, 64));
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -m0;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) sa[kt][r] -= m0;
    }
    __builtin_amdgcn_sched_barrier(0);
    // (the iteration's top barrier also orders the K(1) DMA above behind every wave's K(0) reads: K(2) is the first to reuse stage 0)
    int t = 0;
    for (; t + 1 < nt; t += 2) {
        iteration(t, sa, sb);
        iteration(t + 1, sb, sa);
    }
    if (t < nt) iteration(t, sa, sb);

    // ---- epilogue: O[q][d] = O^T[d][q] / l, stored as (hi, lo) planes ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q < Nq) {
        // out_il (round 6): O stored INTERLEAVED per 32-channel chunk, [row][ldo / 32][hi 32 | lo 32] -- the A operand of ufm_gemm_bf16x3_il (proj);
        // a head is two chunks (dt = 0, 1): the same values at other addresses
        uint16_t* orow = out_il ? out + ((size_t)b * Nq + q) * (2 * ldo) + head * 128 + 4 * hh : out + ((size_t)b * Nq + q) * ldo + head * 64 + 4 * hh;
        const long long dstep = out_il ? 64 : 32, lo_off = out_il ? 32 : out_plane;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4], hi[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = oacc[dt][4 * g + j] * inv;
                const unsigned h0 = pack_bf16x2(v[0], v[1]), h1 = pack_bf16x2(v[2], v[3]);
                hi[0] = __uint_as_float(h0 << 16), hi[1] = __uint_as_float(h0 & 0xffff0000u);
                hi[2] = __uint_as_float(h1 << 16), hi[3] = __uint_as_float(h1 & 0xffff0000u);
                u32x2 pkh = {h0, h1};
                u32x2 pkl = {pack_bf16x2(v[0] - hi[0], v[1] - hi[1]), pack_bf16x2(v[2] - hi[2], v[3] - hi[3])};
                *(u32x2*)(orow + dt * dstep + 8 * g) = pkh;
                *(u32x2*)(orow + lo_off + dt * dstep + 8 * g) = pkl;
            }
    }
}

}  // namespace

// q / k / v: first element of head 0 of batch item 0 (hi plane); the lo planes follow q_plane / in_plane / out_plane elements behind.
// The host has checked shapes, alignment and that one batch item's K / V rows fit a 32-bit byte offset.
int ufm_launch_attn_x3_pw(const uint16_t* q, int ldq, long long q_plane, const uint16_t* k, const uint16_t* v, int ldkv, long long in_plane,
                          uint16_t* out, int ldo, long long out_plane, int B, int Nq, int Nk, int H, float c, hipStream_t stream, int waves, int out_il, int fixref) {
    const int nw = waves == 8 ? 8 : 4;
    const dim3 grid(((Nq + nw * 32 - 1) / (nw * 32)) * H * B), block(nw * 64);
    if (nw == 8) hipLaunchKernelGGL(attn_x3_pw_kernel<8>, grid, block, 0, stream, q, ldq, q_plane, k, v, ldkv, in_plane, out, ldo, out_plane, Nq, Nk, H, c, out_il);
    else if (fixref && c == 0.0f) hipLaunchKernelGGL((attn_x3_pw_kernel<4, true, true>), grid, block, 0, stream, q, ldq, q_plane, k, v, ldkv, in_plane, out, ldo, out_plane, Nq, Nk, H, 1.0f, out_il);  // q pre-scaled
    else if (fixref) hipLaunchKernelGGL((attn_x3_pw_kernel<4, true>), grid, block, 0, stream, q, ldq, q_plane, k, v, ldkv, in_plane, out, ldo, out_plane, Nq, Nk, H, c, out_il);
    else hipLaunchKernelGGL(attn_x3_pw_kernel<4>, grid, block, 0, stream, q, ldq, q_plane, k, v, ldkv, in_plane, out, ldo, out_plane, Nq, Nk, H, c, out_il);
    return 0;
}
