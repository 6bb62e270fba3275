// 256x256x64 "8-phase" bf16 MFMA GEMM (cdna_hip_programming.md section 5, "The 256^2 8-phase template"),
// written for the large UFM linears: C[M,N] = A[M,K] . W[N,K]^T with the same fused epilogue as gemm_bf16.hip.
//
// Structure (one block per CU: 128 KiB LDS, 8 waves = 2 (M) x 4 (N), 128x64 output per wave):
//   * A K-tile (64 deep) is staged as FOUR half-tiles of 128 rows x 128 B (16 KiB each), in the order the MFMA
//     quadrants consume them:  kind 0 = W-lo (columns nh=0 of every wave), 1 = X-lo (rows mh=0), 2 = W-hi,
//     3 = X-hi.  Half-tile index s = 4*tile + kind is first read in phase s-1 and last read there too.
//   * One phase = one 64x32 quadrant of the wave's output over the whole K-tile (16 MFMA 16x16x32):
//       L slot:  ds_read_b128 the operand sub-tile this or the next quadrant needs (8 / 4 / 8 / 4 reads),
//                issue the DMA of half-tile s = phase+6 (2 global_load_lds_dwordx4 per wave),
//                s_waitcnt vmcnt(8)  -> half-tile phase+2 has landed, 4 half-tiles stay in flight,  s_barrier
//       M slot:  s_waitcnt lgkmcnt(0), 16 MFMA under s_setprio(1), s_barrier
//     The two wave groups (wr = 0 / 1, one wave of each per SIMD) run ONE slot apart (wr = 1 takes one extra
//     barrier up front), so one group's MFMA slot overlaps the other's LDS-read/DMA-issue slot.
//   * W-lo of tile t+1 is pre-read in phase 3 of tile t into the register set W-hi(t) just vacated, which makes
//     the reads per phase 8/4/8/4 and swaps the two W register sets every tile -> the loop is unrolled over two
//     K-tiles (8 phases).
//   * Ordering.  RAW: half-tile s is waited for (counted vmcnt, each wave for its own DMA pieces) in the L slot of
//     phase s-2 by both groups, i.e. before the barrier that ends slot 2(s-2)+1; its first reader is group 0 in
//     slot 2(s-1).  WAR: half-tile s+8 reuses the buffer of s (two K-tile buffers), whose last read (group 1,
//     slot 2(s-1)+1) is retired by the lgkmcnt(0) that follows; the DMA into it is issued in phase s+2, slot
//     >= 2(s+2).  No vmcnt(0) and no __syncthreads() inside the loop.
//   * LDS image: 128-B rows, 16-B chunk index XOR (row & 7), applied to the DMA source address and to the
//     ds_read_b128 address (the DMA writes LDS lane-linearly).
#include "gemm_common.h"

namespace {

constexpr int HALF = 128 * 64 * 2;  // 16 KiB half-tile

template <int K>
using IC = std::integral_constant<int, K>;

// NF = 16-row MFMA fragments per wave (5..8): the tile is 32 * NF rows high (wave group wr owns rows [wr * 16 NF, (wr + 1) * 16 NF)
// of it; its second 64-row half simply has NF - 4 fragments).  Lower tiles make ceil(M / height) * (N / 256) fit whole rounds of
// the chip where 256-row tiles would strand CUs (M = 10 952, N = 1024: 172 tiles on 256 CUs -> 232 tiles of 192 rows).  The
// staging, the phase schedule and every output element's accumulation order are those of NF = 8.
template <int OUT_BF16, int NF, int EPI, bool STAMP = false, bool SK = false>
__device__ __forceinline__ void gemm_bf16_8ph_body(const GemmArgs& p, char* smem) {
    static_assert(NF >= 5 && NF <= 8, "NF");
    GemmStamps stamps;
    if constexpr (STAMP) stamps.entry();
    constexpr int RW = 16 * NF, BMT = 2 * RW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- tile of this block: XCD chunking + grouped rasterization (as gemm_bf16.hip) ----
    const int ntn = p.N >> 8, ntm = (p.M - p.m_begin + BMT - 1) / BMT;
    // SK: workgroups 2 v and 2 v + 1 (neighbours in launch order: they run side by side) are the two K halves of output tile v
    const int half_k = SK ? (int)(blockIdx.x & 1) : 0;
    const int bid = SK ? xcd_remap(blockIdx.x >> 1, gridDim.x >> 1) : xcd_remap(blockIdx.x, gridDim.x);
    // grouped rasterization height (tools/lab/gemm_gm_probe.py: flag bits 8..15); 8 vs 4: QKV -8 %, others +-1 %.  (Until late in round 5 this read ALL bits above 8: every A/B arm
    // that set one of the later lab flags -- bits 16..28 -- also ran this kernel column-major, GM = 2^k; the affected logs say so.)
    const int GM = lab_get(p.debug, gemm_lab::RASTER_GROUP) ? lab_get(p.debug, gemm_lab::RASTER_GROUP) : 8;
    const int per_group = GM * ntn;
    const int grp = bid / per_group, in_g = bid - grp * per_group;
    const int gm = min(GM, ntm - grp * GM);
    const int m0 = p.m_begin + (grp * GM + in_g % gm) * BMT, n0 = (in_g / gm) << 8;
    const int nt_all = p.K >> 6;
    const int kt0 = SK ? half_k * (nt_all >> 1) : 0;                       // first K-tile of this workgroup
    const int nt = SK ? (half_k ? nt_all - (nt_all >> 1) : (nt_all >> 1)) : nt_all;
    if (lab_get(p.debug, gemm_lab::STAGGER) & 7) {  // lab (tools/lab/epi_contention.py): first-round blocks start (block / 8) % 4 x units x ~1 us apart
        if (blockIdx.x < 256) {
            const int units = (lab_get(p.debug, gemm_lab::STAGGER) & 7) * (int)((blockIdx.x >> 3) & 3);
            for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(32);
        }
    }

    // ---- DMA source offsets (elements).  Wave w issues pieces w and 8+w of every half-tile; piece = 8 rows ----
    const int srow = lane >> 3, slot = lane & 7;
    unsigned xsrc[2][2], wsrc[2][2];  // [half][piece]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lr = (i * 8 + wave) * 8 + srow;  // local row of the half-tile, lr & 7 == srow
        const int chunk = (slot ^ srow) * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int brow = (lr >> 6) * RW + h * 64 + (lr & 63);   // X half h: rows mh = h of both wave groups (rows past RW: unused)
            const int bcol = (lr >> 5) * 64 + h * 32 + (lr & 31);   // W half h: columns nh = h of the four wave columns
            xsrc[h][i] = 2u * ((unsigned)min(m0 + brow, p.M - 1) * (unsigned)p.lda + chunk) + (unsigned)kt0 * 128u;  // BYTE offsets: the DMA
            wsrc[h][i] = 2u * ((unsigned)(n0 + bcol) * (unsigned)p.ldw + chunk) + (unsigned)kt0 * 128u;               // address is sgpr base + vgpr32
        }
    }
    auto stage = [&](auto kind, int tile) {  // kind: 0 W-lo, 1 X-lo, 2 W-hi, 3 X-hi
        constexpr int KIND = decltype(kind)::value;
        char* dst = smem + ((tile & 1) * 4 + KIND) * HALF + wave * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* base = (KIND & 1) ? (const char*)p.A : (const char*)p.W;
            const unsigned off = ((KIND & 1) ? xsrc[KIND >> 1][i] : wsrc[KIND >> 1][i]) + (unsigned)tile * 128u;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(base + off), LDS_PTR(dst + i * 8192), 16, 0, 0);
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

    auto read_x = [&](int tile, int mh) {
        const char* s = smem + ((tile & 1) * 4 + 1 + 2 * mh) * HALF + x_base;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (mh == 1 && i >= NF - 4) continue;
            xf[i][0] = *(const bf16x8*)(s + i * 2048 + ck0);
            xf[i][1] = *(const bf16x8*)(s + i * 2048 + ck1);
        }
    };
    auto read_w = [&](bf16x8 (&w)[2][2], int tile, int nh) {
        const char* s = smem + ((tile & 1) * 4 + 2 * nh) * HALF + w_base;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            w[j][0] = *(const bf16x8*)(s + j * 2048 + ck0);
            w[j][1] = *(const bf16x8*)(s + j * 2048 + ck1);
        }
    };
    auto mma = [&](auto mh_, auto nh_, bf16x8 (&w)[2][2]) {
        constexpr int MH = decltype(mh_)::value, NH = decltype(nh_)::value;
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
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
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // end of the L slot of phase ph (= 4 * tile + i): issue half-tile ph + 6, wait for half-tile ph + 2, barrier
    const int nhalf = 4 * nt;
    auto l_end = [&](int tile, auto i_) {
        constexpr int I = decltype(i_)::value;
        const int ph = 4 * tile + I;
        if (ph + 6 < nhalf) {
            stage(IC<(I + 2) & 3>{}, tile + (I + 6) / 4);
            wait_vmcnt<8>();
        } else {
            const int inflight = nhalf - ph - 3;  // half-tiles issued after half-tile ph + 2
            if (inflight >= 3) wait_vmcnt<6>();
            else if (inflight == 2) wait_vmcnt<4>();
            else if (inflight == 1) wait_vmcnt<2>();
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto tile_body = [&](int t, bf16x8 (&wcur)[2][2], bf16x8 (&wnxt)[2][2]) {  // wcur holds W-lo(t) on entry
        read_x(t, 0);
        l_end(t, IC<0>{});
        mma(IC<0>{}, IC<0>{}, wcur);
        read_w(wnxt, t, 1);
        l_end(t, IC<1>{});
        mma(IC<0>{}, IC<1>{}, wnxt);
        read_x(t, 1);
        l_end(t, IC<2>{});
        mma(IC<1>{}, IC<1>{}, wnxt);
        if (t + 1 < nt) read_w(wnxt, t + 1, 0);  // W-lo of the next K-tile into the set W-hi(t) just vacated
        l_end(t, IC<3>{});
        mma(IC<1>{}, IC<0>{}, wcur);
    };

    // ---- prologue: half-tiles 0..5 (host guarantees nt >= 2) ----
    stage(IC<0>{}, 0);
    stage(IC<1>{}, 0);
    stage(IC<2>{}, 0);
    stage(IC<3>{}, 0);
    stage(IC<0>{}, 1);
    stage(IC<1>{}, 1);
    wait_vmcnt<8>();  // half-tiles 0 (W-lo) and 1 (X-lo) of tile 0 have landed
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_w(wb, 0, 0);
    if constexpr (STAMP) stamps.t_prologue = gemm_stamp();
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
    // every wave has passed its last ds_read and every DMA has landed (the tail waits end at vmcnt(0)):
    // the staging buffers are free for the epilogue, 16 KiB per wave, two 64x64 passes
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
    if constexpr (SK) {
        // ---- 2-way split-K combine (cdna_hip_programming.md section 5, "In-launch split-K reduction"; the machinery of conv_bf16x3.hip):
        // plain 16-byte slab stores in fragment order -> every wave drains -> barrier -> one lane: agent release, drain, ticket ----
        constexpr int TILE_F = 256 * 256;
        float* const tile_slab = p.slab + (size_t)bid * 2 * TILE_F;
        float* const mine = tile_slab + (size_t)half_k * TILE_F + (wave * 32) * 256 + lane * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) *(f32x4*)(mine + ((h * 4 + n) * 4 + m) * 256) = acc[h][n][m];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* const flag = (unsigned*)smem;  // the staging buffers are idle: word 0 carries the ticket to all waves
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            *flag = __hip_atomic_fetch_add(p.counters + bid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const unsigned ticket = *flag;
        if (ticket != 1u) return;  // the first arriver of this tile is done
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(p.counters + bid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        }
        __syncthreads();
        // the two partial tiles added in HALF order (this workgroup's own one re-read like the other): the same bits whoever came last
        const float* src = tile_slab + (wave * 32) * 256 + lane * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const f32x4 a = *(const f32x4*)(src + ((h * 4 + n) * 4 + m) * 256);
                    const f32x4 b = *(const f32x4*)(src + TILE_F + ((h * 4 + n) * 4 + m) * 256);
                    acc[h][n][m] = a + b;
                }
        __syncthreads();  // `flag` (smem word 0) has been read by every wave before the epilogue re-uses the buffer
    }
    epilogue_two_slices<OUT_BF16, RW - 64, EPI>(p, acc[0], acc[1], smem + wave * 16384, m0 + wr * RW, n0 + wc * 64, lane);
    if constexpr (STAMP) stamps.finish(p.stamps, p.stamp_rows);
}

template <int OUT_BF16, int NF, int EPI, bool STAMP = false, bool SK = false>
__global__ __launch_bounds__(512, 1) void gemm_bf16_8ph_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[8 * HALF];  // [tile & 1][kind]
    gemm_bf16_8ph_body<OUT_BF16, NF, EPI, STAMP, SK>(p, smem);
}

}  // namespace

int ufm_launch_gemm_8ph(const GemmArgs& p, int out_dtype, hipStream_t stream, int nf, int epi) {
    const int bmt = 32 * nf;
    const int ntm = (p.M - p.m_begin + bmt - 1) / bmt, ntn = p.N / 256;
    dim3 grid(ntm * ntn), block(512);
    if (p.splitk == 2) {  // lab (ufm_debug_set_gemm_splitk): the read-modify-write form only, full-height tiles, K >= 256 (each half: >= 2 K-tiles)
        if (out_dtype != UFM_F32 || epi != 3 || nf != 8 || (p.K >> 6) < 4) return 1;
        hipLaunchKernelGGL((gemm_bf16_8ph_kernel<0, 8, 3, false, true>), dim3(2 * ntm * ntn), block, 0, stream, p);
        return 0;
    }
    if (p.stamps && (nf == 6 || nf == 8) && ((out_dtype == UFM_BF16 && epi == 1) || (out_dtype == UFM_F32 && epi == 3))) {  // diagnostic build
        if (nf == 6 && epi == 1) hipLaunchKernelGGL((gemm_bf16_8ph_kernel<1, 6, 1, true>), grid, block, 0, stream, p);
        else if (nf == 6) hipLaunchKernelGGL((gemm_bf16_8ph_kernel<0, 6, 3, true>), grid, block, 0, stream, p);
        else if (epi == 1) hipLaunchKernelGGL((gemm_bf16_8ph_kernel<1, 8, 1, true>), grid, block, 0, stream, p);
        else hipLaunchKernelGGL((gemm_bf16_8ph_kernel<0, 8, 3, true>), grid, block, 0, stream, p);
        return 0;
    }
    // bf16 output: EPI 0 / 1 / 2; fp32 output: EPI 0 / 3 / 4 (gemm_common.h EpiTraits)
#define UFM_L8E(NF_, OUT_, EPI_) hipLaunchKernelGGL((gemm_bf16_8ph_kernel<OUT_, NF_, EPI_>), grid, block, 0, stream, p)
#define UFM_L8(NF_)                                                                                   \
    case NF_:                                                                                         \
        if (out_dtype == UFM_BF16) {                                                                  \
            if (epi == 1) UFM_L8E(NF_, 1, 1); else if (epi == 2) UFM_L8E(NF_, 1, 2); else UFM_L8E(NF_, 1, 0); \
        } else {                                                                                      \
            if (epi == 3) UFM_L8E(NF_, 0, 3); else if (epi == 4) UFM_L8E(NF_, 0, 4); else UFM_L8E(NF_, 0, 0); \
        }                                                                                             \
        break;
    switch (nf) {
        UFM_L8(5) UFM_L8(6) UFM_L8(7) UFM_L8(8)
        default: return 1;
    }
#undef UFM_L8
#undef UFM_L8E
    return 0;
}
