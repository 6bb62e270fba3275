// Persistent form of the 256x256x64 8-phase bf16 GEMM (gemm_bf16_8ph.hip) for the bf16-output Linear layers (QKV: bias + per-column scale,
// fc1: bias + GELU) on launches that are WHOLE rounds of the chip: one workgroup per CU walks tiles v, v + grid, ... and the six half-tiles
// of the NEXT tile's prologue are requested BEFORE the current tile's epilogue, so that the prologue's DMA latency (3.4 k cycles per tile,
// stamps of round 5) and the dispatch gap between two workgroups on a CU (1.3 k) run under the epilogue's 10 k cycles.
//
// Round 1 built a persistent form that issued the prologue BEHIND the epilogue's stores and gained 1-2 %: gfx950 counts stores in vmcnt in
// issue order, so the next tile's first counted wait also waited for them.  Here the DMAs are OLDER than the stores, and every counted wait
// of the loop stays valid as it is: vmcnt(N) admits the N YOUNGEST operations, so with stores younger than the half-tile a wait is for,
// the wait only becomes stricter (it also retires some stores) -- never weaker.  The epilogue cannot stage through the ring any more (the
// ring is receiving the next tile): it goes through a 4-KiB slice per wave behind the ring (160 KiB of LDS in all), 16 rows at a time.
//
// All LDS-DMAs of this kernel are inline asm (M0 written in the statement that uses it; tools/check_attn_x3_isa.py audits that hipcc neither
// touches M0 nor adds an LDS-DMA or a vmcnt wait of its own inside an MFMA block): with the builtin, hipcc would put its own vmcnt(0) in front
// of the epilogue's LDS accesses while the prefetch is in flight.  K loop, fragment layouts and every accumulator's MFMA order are those of
// gemm_bf16_8ph.hip; the epilogue performs its operations per element in the same order: results are BIT-IDENTICAL (tests).
#include "gemm_common.h"

namespace {

constexpr int HALF = 128 * 64 * 2;      // 16 KiB half-tile
constexpr int RING = 8 * HALF;          // [tile & 1][kind]
constexpr int STG = 4096;               // epilogue staging per wave: 16 rows x 64 fp32
constexpr int PERSIST_LDS = RING + 8 * STG;  // 160 KiB

template <int K>
using IC = std::integral_constant<int, K>;

__device__ __forceinline__ void glds16(const char* gbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(gbase) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const char* p) { return (unsigned)(uintptr_t)LDS_PTR(p); }

template <int EPI>  // 1: bias, GELU -> bf16 (fc1)     2: bias, per-column scale -> bf16 (QKV with the Q pre-scale)
__global__ __launch_bounds__(512, 1) void gemm_bf16_8ph_persist_kernel(GemmArgs p) {
    static_assert(EPI == 1 || EPI == 2, "bf16-output epilogues");
    __shared__ __attribute__((aligned(16))) char smem[PERSIST_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int ntn = p.N >> 8, ntm = (p.M - p.m_begin + 255) / 256;
    const int ntiles = ntn * ntm;
    const int nt = p.K >> 6;
    constexpr int GM = 8;
    auto tile_origin = [&](int v, int& m0, int& n0) {  // XCD chunking + grouped rasterization (as gemm_bf16_8ph.hip), over virtual workgroup ids
        const int bid = xcd_remap(v, ntiles);
        const int per_group = GM * ntn;
        const int grp = bid / per_group, in_g = bid - grp * per_group;
        const int gm = min(GM, ntm - grp * GM);
        m0 = p.m_begin + (grp * GM + in_g % gm) * 256;
        n0 = (in_g / gm) << 8;
    };

    // ---- DMA source offsets (bytes) of the current tile.  Wave w issues pieces w and 8 + w of every half-tile; piece = 8 rows ----
    unsigned xsrc[2][2], wsrc[2][2];  // [half][piece]
    auto set_src = [&](int m0, int n0) {
        // an opaque copy of the lane id: hipcc otherwise keeps this function's lane constants live through the K loop and spills them (the
        // spills reload through vmcnt in the epilogue); recomputing them per tile costs a dozen VALU instructions
        int l_ = lane;
        asm volatile("" : "+v"(l_));
        const int srow = l_ >> 3, slot = l_ & 7;
        const int chunk = (slot ^ srow) * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = (i * 8 + wave) * 8 + srow;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int brow = (lr >> 6) * 128 + h * 64 + (lr & 63);
                const int bcol = (lr >> 5) * 64 + h * 32 + (lr & 31);
                xsrc[h][i] = 2u * ((unsigned)min(m0 + brow, p.M - 1) * (unsigned)p.lda + chunk);
                wsrc[h][i] = 2u * ((unsigned)(n0 + bcol) * (unsigned)p.ldw + chunk);
            }
        }
    };
    auto stage = [&](auto kind, int tile) {  // kind: 0 W-lo, 1 X-lo, 2 W-hi, 3 X-hi
        constexpr int KIND = decltype(kind)::value;
        const char* dst = smem + ((tile & 1) * 4 + KIND) * HALF + wave * 1024;
        const char* base = (KIND & 1) ? (const char*)p.A : (const char*)p.W;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned off = ((KIND & 1) ? xsrc[KIND >> 1][i] : wsrc[KIND >> 1][i]) + (unsigned)tile * 128u;
            glds16(base, off, lds_addr_of(dst + i * 8192));
        }
    };
    auto prologue = [&]() {  // half-tiles 0..5 (host guarantees nt >= 2)
        stage(IC<0>{}, 0);
        stage(IC<1>{}, 0);
        stage(IC<2>{}, 0);
        stage(IC<3>{}, 0);
        stage(IC<0>{}, 1);
        stage(IC<1>{}, 1);
    };

    // ---- fragment read offsets ----
    const int kfr = lane & 15, kfq = lane >> 4;
    const int sw = kfr & 7;
    const int ck0 = ((kfq ^ sw) << 4), ck1 = (((4 + kfq) ^ sw) << 4);
    const int x_base = (wr * 64 + kfr) * 128;  // + i * 2048
    const int w_base = (wc * 32 + kfr) * 128;  // + j * 2048

    f32x4 acc[2][4][4];  // [mh][n][m]
    bf16x8 xf[4][2], wa[2][2], wb[2][2];
    auto read_x = [&](int tile, int mh) {
        const char* s = smem + ((tile & 1) * 4 + 1 + 2 * mh) * HALF + x_base;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
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
                for (int i = 0; i < 4; ++i)
                    acc[MH][NH * 2 + j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][kk], xf[i][kk], acc[MH][NH * 2 + j][i], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // end of the L slot of phase ph (= 4 * tile + i): issue half-tile ph + 6, wait for half-tile ph + 2, barrier.  The counts are those of
    // gemm_bf16_8ph.hip: stores of the previous output tile's epilogue that are still in flight are YOUNGER than every half-tile a wait of the
    // first phases is for and OLDER than the DMAs issued here, so they can only make a wait stricter.
    const int nhalf = 4 * nt;
    bool stores_behind = false;  // wave-uniform: the previous output tile's 16 stores may be in flight (always exactly 16 per wave: no ragged rows)
    auto l_end = [&](int tile, auto i_) {
        constexpr int I = decltype(i_)::value;
        const int ph = 4 * tile + I;
        if (ph + 6 < nhalf) {
            stage(IC<(I + 2) & 3>{}, tile + (I + 6) / 4);
            // K-tile 0 of an output tile that follows another one: this wave's 16 row stores of the previous epilogue sit, in issue order, between the
            // prologue's half-tiles and the DMAs issued from here on.  vmcnt(8) would also wait for all but 8 - 2 (I + 1) of those stores -- the drain the
            // prefetch was meant to overlap.  vmcnt(8 + 16) asks for exactly the half-tile this phase needs: with every store still in flight the 24
            // youngest operations are the 8 - 2 (I + 1) remaining prologue pieces, the 16 stores and the 2 (I + 1) pieces issued since; with some stores
            // retired (in issue order: everything older has retired with them) the wait can only be stricter than needed.
            if (tile == 0 && stores_behind) wait_vmcnt<24>();
            else wait_vmcnt<8>();
        } else {
            const int inflight = nhalf - ph - 3;
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
        if (t + 1 < nt) read_w(wnxt, t + 1, 0);
        l_end(t, IC<3>{});
        mma(IC<1>{}, IC<0>{}, wcur);
    };

    int v = blockIdx.x, m0, n0;
    tile_origin(v, m0, n0);
    set_src(m0, n0);
    prologue();
    char* const ws = smem + RING + wave * STG;
    for (;;) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[h][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        // half-tiles 0 (W-lo) and 1 (X-lo) of this tile's K-tile 0 have landed; behind an epilogue the 16 stores are younger than all 12 prologue pieces
        if (stores_behind) wait_vmcnt<24>();
        else wait_vmcnt<8>();
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
        // every wave has passed its last ds_read of the ring and every DMA has landed (the tail waits end at vmcnt(0))

        int le_ = lane;  // (opaque copy, as in set_src: the epilogue's lane constants are recomputed per tile)
        asm volatile("" : "+v"(le_));
        const int fr = le_ & 15, fq = le_ >> 4, r8 = le_ >> 3, c8 = le_ & 7;
        // ---- this tile's bias (and scale) first: their waits are hipcc's, which does not see the asm DMAs -- nothing is in flight here ----
        const int row0 = m0 + wr * 128, col0 = n0 + wc * 64;
        f32x4 bv[4], gv[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            bv[n] = *(const f32x4*)(p.bias + col0 + n * 16 + fq * 4);
            if (EPI == 2) gv[n] = *(const f32x4*)(p.gamma + col0 + n * 16 + fq * 4);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            asm volatile("" : "+v"(bv[n]));  // loaded HERE, in front of the prefetch
            if (EPI == 2) asm volatile("" : "+v"(gv[n]));
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- the next tile's prologue, in front of this tile's stores ----
        const int vn = v + (int)gridDim.x;
        const bool more = vn < ntiles;
        if (more) {
            tile_origin(vn, m0, n0);
            set_src(m0, n0);
            prologue();
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- epilogue: 16 rows x 64 columns at a time through the wave's own 4 KiB (same operations per element as epilogue_lds) ----
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    f32x4 x = acc[h][n][m] + bv[n];
                    if (EPI == 1) x = gelu_bf16_x4(x);
                    else x *= gv[n];
                    *(f32x4*)(ws + fr * 256 + (((n * 4 + fq) ^ fr) << 4)) = x;
                }
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    const int r = pass * 8 + r8;
                    const f32x4 v0 = *(const f32x4*)(ws + r * 256 + (((2 * c8) ^ r) << 4));
                    const f32x4 v1 = *(const f32x4*)(ws + r * 256 + (((2 * c8 + 1) ^ r) << 4));
                    const u32x4 pk = {pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])};
                    *(u32x4*)((uint16_t*)p.out + (size_t)(row0 + h * 64 + m * 16 + r) * p.ldo + col0 + c8 * 8) = pk;
                }
                __builtin_amdgcn_sched_barrier(0);  // one 16-row step at a time: interleaved steps cost registers the kernel does not have
            }
        if (!more) break;
        v = vn;
        stores_behind = true;
    }
}

}  // namespace

// The host has checked: bf16 output with the compile-time epilogue `epi` (1 or 2), every tile whole (M - m_begin a multiple of 256),
// N % 256 == 0, K >= 128, 32-bit operand offsets.  grid = min(tiles, CUs): a launch of whole rounds keeps every workgroup equally busy.
int ufm_launch_gemm_8ph_persist(const GemmArgs& p, hipStream_t stream, int epi, int ncu) {
    const int tiles = ((p.M - p.m_begin) / 256) * (p.N / 256);
    dim3 grid(tiles < ncu ? tiles : ncu), block(512);
    if (epi == 1) hipLaunchKernelGGL((gemm_bf16_8ph_persist_kernel<1>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((gemm_bf16_8ph_persist_kernel<2>), grid, block, 0, stream, p);
    return 0;
}
