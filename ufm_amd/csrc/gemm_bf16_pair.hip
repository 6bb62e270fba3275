// 256x128x64 "pair" bf16 MFMA GEMM: the 8-phase wave schedule of gemm_bf16_8ph.hip in a 4-wave workgroup that leaves room for a
// SECOND resident workgroup on the CU (72 KiB of LDS, 256 registers per wave), so that one workgroup's epilogue -- the fp32
// read-modify-write of the residual stream, 17-43 % of a proj / fc2 launch (profiles/r04/gemm_epilogue_contention.log) -- runs under
// its neighbour's K loop instead of under nothing.  C[M,N] = A[M,K] . W[N,K]^T with the fused epilogues of gemm_common.h.
//
// Structure (4 waves = 2 (M) x 2 (N), 128x64 output per wave: the per-wave fragment traffic of the 256x256 kernel):
//   * A K-tile (64 deep) is staged as four half-tiles in the order the wave's four 64x32 quadrants consume them:
//       kind 0 = W-lo (columns nh = 0 of both wave columns, 64 rows x 128 B = 8 KiB), 1 = X-lo (rows mh = 0 of both wave rows,
//       128 x 128 B = 16 KiB), 2 = W-hi, 3 = X-hi.  Half-tile s = 4 tile + kind is read in phase s - 1 only.
//   * The X and the W half-tiles each live in a ring of THREE buffers (48 + 24 KiB): half-tile s + 6 takes the buffer of s.
//   * One phase = one quadrant over the K-tile:
//       ds_read_b128 the operand sub-tile the next quadrant needs (8 / 4 / 8 / 4),  issue the DMA of half-tile phase + 6 (4 or 2
//       global_load_lds_dwordx4 per wave),  s_waitcnt vmcnt(12) -> half-tile phase + 2 has landed, four half-tiles (one of each kind:
//       4 + 2 + 4 + 2 instructions) stay in flight,  s_waitcnt lgkmcnt(0),  s_barrier,  16 MFMA 16x16x32 under s_setprio(1).
//     ONE barrier per phase: the four waves sit on four different SIMDs, so inside a workgroup nothing overlaps a wave's LDS / DMA
//     slot -- the co-resident workgroup's wave on the same SIMD does (its MFMA slot, or its whole epilogue).
//   * Ordering.  RAW: half-tile s is waited for (counted vmcnt, each wave for its own pieces) in phase s - 2 in front of that
//     phase's barrier; it is read in phase s - 1.  WAR: the DMA of s + 6 into the buffer of s is issued in phase s, behind the barrier
//     of phase s - 1, in front of which every wave retired (lgkmcnt(0)) its reads of s.  No vmcnt(0), no __syncthreads() in the loop.
//   * LDS image, XCD chunking, grouped rasterization, LDS-staged epilogue: as the 256x256 kernel.  Every accumulator sees the same
//     MFMA sequence in the same K order as in the 128x128 and the 256x256 kernels: results are bit-identical (tested).
//   * First round only: the second resident workgroup of a CU (non-zero LDS base) starts `stagger` sleeps late, so the two do not
//     reach their epilogues together (co-resident workgroups otherwise start within 40 ns and stay in lockstep,
//     profiles/r03/block_placement.log); while it sleeps its neighbour has the CU's matrix pipes to itself, so the delay is not lost.
#include "gemm_common.h"

namespace {

constexpr int XH = 128 * 128;        // X half-tile, 16 KiB
constexpr int WH = 64 * 128;         // W half-tile, 8 KiB
constexpr int W_RING = 3 * XH;       // byte offset of the W ring
constexpr int PAIR_LDS = 3 * XH + 3 * WH;  // 72 KiB

template <int K>
using IC = std::integral_constant<int, K>;

// NF = 16-row fragments per wave row (5..8): the tile is 32 NF rows high, as in gemm_bf16_8ph.hip (the second 64-row half of a wave's rows
// has NF - 4 fragments; staging, schedule and every element's accumulation order are those of NF = 8, so all heights are bitwise identical).
template <int OUT_BF16, int NF, int EPI, bool STAMP = false>
__device__ __forceinline__ void gemm_bf16_pair_body(const GemmArgs& p, char* smem) {
    static_assert(NF >= 5 && NF <= 8, "NF");
    constexpr int RW = 16 * NF, BMT = 2 * RW, BNT = 128;
    GemmStamps stamps;
    if constexpr (STAMP) stamps.entry();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int ntn = p.N >> 7, ntm = (p.M - p.m_begin + BMT - 1) / BMT;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GM = 8;
    const int per_group = GM * ntn;
    const int grp = bid / per_group, in_g = bid - grp * per_group;
    const int gm = min(GM, ntm - grp * GM);
    const int m0 = p.m_begin + (grp * GM + in_g % gm) * BMT, n0 = (in_g / gm) * BNT;
    const int nt = p.K >> 6;

    // first-round stagger of the second resident workgroup (debug bits 16..22: units of s_sleep(64) ~ 4096 cycles)
    {
        const int units = lab_get(p.debug, gemm_lab::STAGGER);
        if (units && blockIdx.x < 2 * 256) {
            unsigned lds_alloc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(lds_alloc));
            if (lds_alloc & 0x1ff)
                for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(64);
        }
    }

    // ---- DMA source offsets (bytes).  Piece = 8 rows x 128 B; wave w issues pieces w, 4 + w, ... of every half-tile ----
    const int srow = lane >> 3, slot = lane & 7;
    const int chunk = (slot ^ srow) * 8;
    unsigned xsrc[2][4], wsrc[2][2];  // [half][piece]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int lr = (i * 4 + wave) * 8 + srow;                 // local row of the X half-tile, lr & 7 == srow
            const int brow = (lr >> 6) * RW + h * 64 + (lr & 63);     // X half h: rows mh = h of both wave rows (rows past RW: unused)
            xsrc[h][i] = 2u * ((unsigned)min(m0 + brow, p.M - 1) * (unsigned)p.lda + chunk);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = (i * 4 + wave) * 8 + srow;                 // local row of the W half-tile
            const int bcol = (lr >> 5) * 64 + h * 32 + (lr & 31);     // W half h: columns nh = h of both wave columns
            wsrc[h][i] = 2u * ((unsigned)(n0 + bcol) * (unsigned)p.ldw + chunk);
        }
    }
    auto stage = [&](auto kind, auto buf, int tile) {  // kind: 0 W-lo, 1 X-lo, 2 W-hi, 3 X-hi; buf: ring slot
        constexpr int KIND = decltype(kind)::value, BUF = decltype(buf)::value;
        if constexpr (KIND & 1) {
            char* dst = smem + BUF * XH + wave * 1024;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR((const char*)p.A + xsrc[KIND >> 1][i] + (unsigned)tile * 128u), LDS_PTR(dst + i * 4096), 16, 0, 0);
        } else {
            char* dst = smem + W_RING + BUF * WH + wave * 1024;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR((const char*)p.W + wsrc[KIND >> 1][i] + (unsigned)tile * 128u), LDS_PTR(dst + i * 4096), 16, 0, 0);
        }
    };

    // ---- fragment read offsets ----
    const int fr = lane & 15, fq = lane >> 4;
    const int sw = fr & 7;
    const int ck0 = ((fq ^ sw) << 4), ck1 = (((4 + fq) ^ sw) << 4);
    const int x_base = (wr * 64 + fr) * 128;  // + i * 2048
    const int w_base = (wc * 32 + fr) * 128;  // + j * 2048

    f32x4 acc[2][4][4];  // [mh][n][m]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[h][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 xf[4][2], wa[2][2], wb[2][2];

    auto read_x = [&](auto buf, auto mh_) {
        const char* s = smem + decltype(buf)::value * XH + x_base;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (decltype(mh_)::value == 1 && i >= NF - 4) continue;
            xf[i][0] = *(const bf16x8*)(s + i * 2048 + ck0);
            xf[i][1] = *(const bf16x8*)(s + i * 2048 + ck1);
        }
    };
    auto read_w = [&](bf16x8 (&w)[2][2], auto buf) {
        const char* s = smem + W_RING + decltype(buf)::value * WH + w_base;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            w[j][0] = *(const bf16x8*)(s + j * 2048 + ck0);
            w[j][1] = *(const bf16x8*)(s + j * 2048 + ck1);
        }
    };
    auto mma = [&](auto mh_, auto nh_, bf16x8 (&w)[2][2]) {
        constexpr int MH = decltype(mh_)::value, NH = decltype(nh_)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < (MH == 0 ? 4 : NF - 4); ++i)
                    acc[MH][NH * 2 + j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][kk], xf[i][kk], acc[MH][NH * 2 + j][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // end of the load slot of phase ph = 4 tile + I: issue half-tile ph + 6 into the ring slot of half-tile ph, wait for half-tile
    // ph + 2, retire this phase's fragment reads, barrier.  B2 = ring slot of the tile's lo halves = (2 tile) % 3.
    const int nhalf = 4 * nt;
    auto l_end = [&](int tile, auto i_, auto b2_) {
        constexpr int I = decltype(i_)::value, B2 = decltype(b2_)::value;
        const int ph = 4 * tile + I;
        if (ph + 6 < nhalf) {
            stage(IC<(I + 2) & 3>{}, IC<(I < 2) ? B2 : (B2 + 1) % 3>{}, tile + (I + 6) / 4);
            wait_vmcnt<12>();
        } else {
            // the tail: the half-tiles younger than ph + 2 are the last nhalf - ph - 3 of the K range, kinds ... 1, 2, 3 = 4, 2, 4 pieces
            const int inflight = nhalf - ph - 3;
            if (inflight >= 3) wait_vmcnt<10>();
            else if (inflight == 2) wait_vmcnt<6>();
            else if (inflight == 1) wait_vmcnt<4>();
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this phase's fragment reads are retired in front of the barrier (WAR above)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // ring slots of tile t: X-lo / W-lo in B2, X-hi / W-hi in B2 + 1 (mod 3); the next tile's lo halves in B2 + 2
    auto tile_body = [&](int t, auto b2_, bf16x8 (&wcur)[2][2], bf16x8 (&wnxt)[2][2]) {  // wcur holds W-lo(t) on entry
        constexpr int B2 = decltype(b2_)::value, B2H = (B2 + 1) % 3, B2N = (B2 + 2) % 3;
        read_x(IC<B2>{}, IC<0>{});
        l_end(t, IC<0>{}, b2_);
        mma(IC<0>{}, IC<0>{}, wcur);
        read_w(wnxt, IC<B2H>{});
        l_end(t, IC<1>{}, b2_);
        mma(IC<0>{}, IC<1>{}, wnxt);
        read_x(IC<B2H>{}, IC<1>{});
        l_end(t, IC<2>{}, b2_);
        mma(IC<1>{}, IC<1>{}, wnxt);
        if (t + 1 < nt) read_w(wnxt, IC<B2N>{});  // W-lo of the next K-tile into the set W-hi(t) just vacated
        l_end(t, IC<3>{}, b2_);
        mma(IC<1>{}, IC<0>{}, wcur);
    };

    // ---- prologue: half-tiles 0..5 (host guarantees nt >= 2): tile 0 in slots 0 / 1, the lo halves of tile 1 in slot 2 ----
    stage(IC<0>{}, IC<0>{}, 0);
    stage(IC<1>{}, IC<0>{}, 0);
    stage(IC<2>{}, IC<1>{}, 0);
    stage(IC<3>{}, IC<1>{}, 0);
    stage(IC<0>{}, IC<2>{}, 1);
    stage(IC<1>{}, IC<2>{}, 1);
    wait_vmcnt<12>();  // half-tiles 0 (W-lo) and 1 (X-lo) of tile 0 have landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_w(wb, IC<0>{});
    // phase 0 re-fills W slot 0 (half-tile 6): every wave's read of half-tile 0 is retired in front of a barrier first
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    if constexpr (STAMP) stamps.t_prologue = gemm_stamp();
    // (2 t) % 3 has period 3, the W register sets swap every tile: six tile bodies per trip
    for (int t = 0;;) {
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
    if constexpr (STAMP) stamps.t_loop = gemm_stamp();
    // Every wave retired its last ds_read in front of the last phase's barrier and every DMA has landed (the tail waits end at
    // vmcnt(0)): the rings are free for the epilogue staging, 16 KiB per wave, two 64x64 passes.
    if (lab_get(p.debug, gemm_lab::NO_EPILOGUE)) {  // ablation (tools/): no epilogue traffic; keep the accumulators live
        float keep = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) keep += acc[h][n][m][0] + acc[h][n][m][1] + acc[h][n][m][2] + acc[h][n][m][3];
        if (keep == 123.456f) ((float*)p.out)[0] = keep;
        return;
    }
    epilogue_two_slices<OUT_BF16, RW - 64, EPI>(p, acc[0], acc[1], smem + wave * 16384, m0 + wr * RW, n0 + wc * 64, lane);
    if constexpr (STAMP) stamps.finish(p.stamps, p.stamp_rows);
}

template <int OUT_BF16, int NF, int EPI, bool STAMP = false>
__global__ __launch_bounds__(256, 2) void gemm_bf16_pair_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[PAIR_LDS];
    gemm_bf16_pair_body<OUT_BF16, NF, EPI, STAMP>(p, smem);
}

}  // namespace

int ufm_launch_gemm_pair(const GemmArgs& p, int out_dtype, hipStream_t stream, int nf, int epi) {
    const int bmt = 32 * nf;
    const int ntm = (p.M - p.m_begin + bmt - 1) / bmt, ntn = p.N / 128;
    dim3 grid(ntm * ntn), block(256);
    if (p.stamps && (nf == 6 || nf == 8) && ((out_dtype == UFM_BF16 && epi == 1) || (out_dtype == UFM_F32 && epi == 3))) {  // diagnostic build
        if (nf == 6 && epi == 1) hipLaunchKernelGGL((gemm_bf16_pair_kernel<1, 6, 1, true>), grid, block, 0, stream, p);
        else if (nf == 6) hipLaunchKernelGGL((gemm_bf16_pair_kernel<0, 6, 3, true>), grid, block, 0, stream, p);
        else if (epi == 1) hipLaunchKernelGGL((gemm_bf16_pair_kernel<1, 8, 1, true>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((gemm_bf16_pair_kernel<0, 8, 3, true>), grid, block, 0, stream, p);
        return 0;
    }
#define UFM_LPE(NF_, OUT_, EPI_) hipLaunchKernelGGL((gemm_bf16_pair_kernel<OUT_, NF_, EPI_>), grid, block, 0, stream, p)
#define UFM_LP(NF_)                                                                                   \
    case NF_:                                                                                         \
        if (out_dtype == UFM_BF16) {                                                                  \
            if (epi == 1) UFM_LPE(NF_, 1, 1); else if (epi == 2) UFM_LPE(NF_, 1, 2); else UFM_LPE(NF_, 1, 0); \
        } else {                                                                                      \
            if (epi == 3) UFM_LPE(NF_, 0, 3); else if (epi == 4) UFM_LPE(NF_, 0, 4); else UFM_LPE(NF_, 0, 0); \
        }                                                                                             \
        break;
    switch (nf) {
        UFM_LP(5) UFM_LP(6) UFM_LP(7) UFM_LP(8)
        default: return 1;
    }
#undef UFM_LP
#undef UFM_LPE
    return 0;
}
