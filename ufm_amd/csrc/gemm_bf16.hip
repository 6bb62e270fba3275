// bf16 MFMA GEMM with fused epilogue: C[M,N] = A[M,K] . W[N,K]^T  (both operands K-contiguous).
//
// gfx950 design (cdna_hip_programming.md section 5):
//   * 128x128x64 block tile, 4 waves (2x2), 64x64 per wave = 4x4 v_mfma_f32_16x16x32_bf16 tiles.
//   * operands go global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction),
//     two LDS stages, one barrier per K-step; the next K-tile's DMA overlaps this tile's MFMAs.
//   * LDS image: 128-byte rows (64 bf16 of K), 16-byte chunk index XOR (row & 7).  The DMA writes
//     LDS linearly, so the swizzle is applied to the per-lane SOURCE address and again on the
//     ds_read_b128 fragment reads (rule 21) -> conflict-free b128 reads.
//   * MFMA is issued as D = Wfrag . Afrag so a lane's 4 accumulator registers are 4 CONSECUTIVE
//     output columns of one row: the epilogue's bias/gamma/residual accesses are 16-byte.
//   * XCD-aware tile order (T1): consecutive logical tiles share an A row-panel and sit in one L2.
#include "gemm_common.h"

namespace {

template <int OUT_BF16>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntn = p.N / BN, ntm = (p.M - p.m_begin + BM - 1) / BM;
    // XCD chunking + grouped rasterization: 64 consecutive logical tiles (what one XCD runs at a time) cover
    // 8 M-panels x 8 N-panels, so each A and W panel is re-used 8x out of that XCD's L2 (N-fastest order
    // streamed W from beyond L2 once per M-panel: 622 MB fetched for 51 MB of operands, PMC FETCH_SIZE)
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    constexpr int GM = 8;
    const int per_group = GM * ntn;
    const int grp = bid / per_group, in_g = bid - grp * per_group;
    const int gm = min(GM, ntm - grp * GM);
    const int tm = grp * GM + in_g % gm, tn = in_g / gm;
    const int m0 = p.m_begin + tm * BM, n0 = tn * BN;
    const int nk = p.K / BK;

    // ---- staging addresses: wave w issues DMA pieces 4w..4w+3 of each operand tile; piece = 8 rows ----
    const int srow = lane >> 3, slot = lane & 7;
    const uint16_t* ga[4];
    const uint16_t* gw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = (wave * 4 + i) * 8 + srow;     // tile row 0..127
        int chunk = slot ^ (r & 7);            // source chunk that must land in this lane's LDS slot
        int ar = min(m0 + r, p.M - 1);         // clamp: tail rows re-read a valid row, stores are masked
        ga[i] = p.A + (size_t)ar * p.lda + chunk * 8;
        gw[i] = p.W + (size_t)(n0 + r) * p.ldw + chunk * 8;
    }
    auto stage = [&](int buf, int kt) {
        if (p.debug & 2) return;  // ablation (tools/): no DMA
        char* sa = smem + buf * STAGE_BYTES + wave * 4096;
        char* sb = sa + BM * BK * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(ga[i] + kt * BK), LDS_PTR(sa + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gw[i] + kt * BK), LDS_PTR(sb + i * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read addresses ----
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    int a_off[4], b_off[4];  // byte offset of (row, chunk 0) ; chunk XOR applied per kk
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_off[i] = (wr * 64 + i * 16 + fr) * 128;
        b_off[i] = BM * BK * 2 + (wc * 64 + i * 16 + fr) * 128;
    }
    const int sw = fr & 7;  // (row & 7): tile row bases are multiples of 16

    f32x4 acc[4][4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        const char* s = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int coff = ((kk * 4 + fq) ^ sw) * 16;
            bf16x8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *(const bf16x8*)(s + a_off[i] + coff);
                b[i] = *(const bf16x8*)(s + b_off[i] + coff);
            }
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[n], a[m], acc[n][m], 0, 0, 0);
        }
    }

    if (p.debug & 4) {  // ablation (tools/): no epilogue traffic; keep the accumulators live
        float keep = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) keep += acc[n][m][0] + acc[n][m][1] + acc[n][m][2] + acc[n][m][3];
        if (keep == 123.456f) ((float*)p.out)[0] = keep;
        return;
    }
    if (p.debug & 8) {  // A/B hook: the direct (unstaged) epilogue
        epilogue<OUT_BF16>(p, acc, m0 + wr * 64, n0 + wc * 64, fr, fq);
        return;
    }
    __syncthreads();  // every wave is done reading the staging buffers
    epilogue_lds<OUT_BF16>(p, acc, smem + wave * 16384, m0 + wr * 64, n0 + wc * 64, lane);
}

// ---------------------------------------------------------------------------------------------
// Persistent large-GEMM kernel: 256 x BN x 32 tiles (BN = 256 or 128), 8 waves, one block per CU
// that walks its share of the tiles with the DMA ring running ACROSS tile boundaries.
//
// Why (measured on MI355X, tools/ksweep.py + PMC): with one tile per block the K-independent cost
// (first-load latency of every tile, epilogue with the memory system idle during the main loop and
// saturated at its end, all CUs in lockstep) is 25-50 us per GEMM -- as much as the K=1024 main
// loop itself; and waves sat 40-47 % of their time on the DMA wait.  Here:
//   * K-step 32, NS-stage LDS ring (4 x 32 KiB for BN=256, 6 x 24 KiB for BN=128); the fragments of
//     K-tile g+1 are read into a second register set while the MFMAs of K-tile g run, so a ring
//     stage is free as soon as its fragments are in registers and the DMA runs NS K-tiles ahead;
//   * the flat index g runs over (tile, k) pairs of this block: when the issue cursor reaches the end
//     of a tile it moves to the block's next tile, so the next tile's first K-tiles are already in
//     LDS when the epilogue of the current tile finishes (stores are asynchronous);
//   * one counted s_waitcnt vmcnt + one raw s_barrier per K-tile.  RAW: tile g+1 is read after every
//     wave's vmcnt wait for it and the barrier.  WAR: stage g%NS is restaged (DMA of g+NS) only after
//     the barrier of iteration g, which every wave reaches after its lgkmcnt(0) for the fragments of
//     tile g (read during iteration g-1).  The vmcnt immediate counts only DMA pieces; epilogue
//     loads/stores that slip into the in-order queue make the wait stricter, never weaker.
//   * LDS rows are 64 B, chunk XOR g[(row>>2)&3] (conflict-free b128 reads); XCD-aware, grouped tile
//     order so concurrently running blocks share A and W panels in their XCD's L2.
// ---------------------------------------------------------------------------------------------
constexpr int PBM = 256, PBK = 32;
constexpr int GROUP_M = 4;

__device__ __forceinline__ int swz64(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }

template <int BN, int OUT_BF16>
__global__ __launch_bounds__(512, 2) void gemm_bf16_persistent(GemmArgs p) {
    constexpr int NS = BN == 256 ? 4 : 6;            // ring stages
    constexpr int STAGE = (PBM + BN) * PBK * 2;      // 32 / 24 KiB
    constexpr int WN = BN / 64, WM = 8 / WN;         // 2x4 or 4x2 waves
    constexpr int TM = (PBM / WM) / 16;              // 8 or 4 row tiles per wave
    constexpr int PA = 2, PW = BN / 128;             // DMA pieces (16 rows x 64 B) per wave per K-tile
    constexpr int P = PA + PW;
    __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntn = p.N / BN, ntm = (p.M + PBM - 1) / PBM, ntiles = ntm * ntn;
    const int nk = p.K / PBK;
    const int nblk = gridDim.x;
    const int bslot = xcd_remap(blockIdx.x, nblk);
    const int my_tiles = (ntiles - bslot + nblk - 1) / nblk;  // tiles bslot, bslot+nblk, ...
    const int G = my_tiles * nk;                              // flat (tile, k) iterations of this block

    auto tile_origin = [&](int j, int& m0, int& n0) {  // j-th tile of this block -> grouped rasterization
        const int lid = bslot + j * nblk;
        const int per_group = GROUP_M * ntn;
        const int g = lid / per_group, in_g = lid - g * per_group;
        const int gm = min(GROUP_M, ntm - g * GROUP_M);
        m0 = (p.debug == 1) ? 0 : (g * GROUP_M + in_g % gm) * PBM;
        n0 = (p.debug == 1) ? 0 : (in_g / gm) * BN;
    };

    // ---- DMA issue cursor ----
    const int srow = lane >> 2, slot = lane & 3;
    int ia_off[PA], iw_off[PW];  // element offsets of this lane's source rows for the tile being issued
    int i_tile = 0, i_kt = 0;
    auto set_issue_tile = [&](int j) {
        int m0, n0;
        tile_origin(j, m0, n0);
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            const int r = wave * 32 + i * 16 + srow;
            ia_off[i] = min(m0 + r, p.M - 1) * p.lda + (slot ^ swz64(r)) * 8;
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int r = wave * (16 * PW) + i * 16 + srow;
            iw_off[i] = (n0 + r) * p.ldw + (slot ^ swz64(r)) * 8;
        }
    };
    auto issue = [&](int g) {  // DMA of flat iteration g (always == the cursor position)
        if (p.debug != 2) {
            char* sa = smem + (g % NS) * STAGE + wave * 2048;
            char* sb = smem + (g % NS) * STAGE + PBM * PBK * 2 + wave * (1024 * PW);
#pragma unroll
            for (int i = 0; i < PA; ++i)
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(p.A + ia_off[i] + i_kt * PBK), LDS_PTR(sa + i * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < PW; ++i)
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(p.W + iw_off[i] + i_kt * PBK), LDS_PTR(sb + i * 1024), 16, 0, 0);
        }
        if (++i_kt == nk) {
            i_kt = 0;
            if (++i_tile < my_tiles) set_issue_tile(i_tile);
        }
    };

    // ---- fragment offsets ----
    const int wr = wave / WN, wc = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    int a_off[TM], b_off[4];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wr * (TM * 16) + i * 16 + fr;
        a_off[i] = r * 64 + ((fq ^ swz64(r)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wc * 64 + i * 16 + fr;
        b_off[i] = PBM * PBK * 2 + r * 64 + ((fq ^ swz64(r)) << 4);
    }

    f32x4 acc[4][TM];
    auto zero_acc = [&]() {
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int m = 0; m < TM; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    bf16x8 a0[TM], b0[4], a1[TM], b1[4];
    auto read_frags = [&](bf16x8 (&a)[TM], bf16x8 (&b)[4], int g) {
        const char* s = smem + (g % NS) * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = *(const bf16x8*)(s + b_off[i]);
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *(const bf16x8*)(s + a_off[i]);
    };
    auto mfma_all = [&](bf16x8 (&a)[TM], bf16x8 (&b)[4]) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[n], a[m], acc[n][m], 0, 0, 0);
    };
    // wait until this wave's DMA pieces of flat iteration x have landed; `issued` = highest index issued so far
    auto wait_landed = [&](int x, int issued) {
        const int younger = issued - x;  // 0 .. NS-1 tiles issued after x
        if (younger >= NS - 1) wait_vmcnt<P*(NS - 1)>();
        else if (younger == NS - 2) wait_vmcnt<P*(NS - 2)>();
        else if (younger == 2 && NS > 4) wait_vmcnt<P * 2>();
        else if (younger == 3 && NS > 5) wait_vmcnt<P * 3>();
        else if (younger == 1) wait_vmcnt<P>();
        else wait_vmcnt<0>();
    };

    int c_tile = 0, c_kt = 0;  // compute cursor
    auto finish_kt = [&]() {   // after the MFMAs of one K-tile: epilogue at the end of a tile
        if (++c_kt == nk) {
            int m0, n0;
            tile_origin(c_tile, m0, n0);
            if (p.debug == 1) tile_origin(c_tile, m0, n0);
#pragma unroll
            for (int h = 0; h < TM / 4; ++h) {
                f32x4 part[4][4];
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int m = 0; m < 4; ++m) part[n][m] = acc[n][h * 4 + m];
                epilogue<OUT_BF16>(p, part, m0 + wr * (TM * 16) + h * 64, n0 + wc * 64, fr, fq);
            }
            zero_acc();
            c_kt = 0;
            ++c_tile;
        }
    };

    if (G == 0) return;
    zero_acc();
    set_issue_tile(0);
    int issued = -1;
    constexpr bool DB = (BN == 128);  // second fragment register set only where 128 accumulators leave room
    if constexpr (DB) {
        for (int g = 0; g < NS && g < G; ++g) {
            issue(g);
            issued = g;
        }
        wait_landed(0, issued);
        __builtin_amdgcn_s_barrier();
        read_frags(a0, b0, 0);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        for (int g = 0; g < G; g += 2) {
            // ---- even: MFMA(g) from set 0 ; prefetch set 1 <- g+1 ----
            if (g + 1 < G) {
                wait_landed(g + 1, issued);
                __builtin_amdgcn_s_barrier();
                if (issued + 1 < G) issue(++issued);  // into stage g % NS (fragments of g are in registers)
                read_frags(a1, b1, g + 1);
            }
            mfma_all(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            finish_kt();
            if (g + 1 >= G) break;
            // ---- odd: MFMA(g+1) from set 1 ; prefetch set 0 <- g+2 ----
            if (g + 2 < G) {
                wait_landed(g + 2, issued);
                __builtin_amdgcn_s_barrier();
                if (issued + 1 < G) issue(++issued);
                read_frags(a0, b0, g + 2);
            }
            mfma_all(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            finish_kt();
        }
    } else {
        // single fragment set: stage (g-1)%NS is free after the barrier of iteration g (its fragments were
        // waited for before the MFMAs of iteration g-1), so the DMA runs NS-1 K-tiles ahead.
        for (int g = 0; g < NS - 1 && g < G; ++g) {
            issue(g);
            issued = g;
        }
        for (int g = 0; g < G; ++g) {
            wait_landed(g, issued);
            __builtin_amdgcn_s_barrier();
            if (issued + 1 < G) issue(++issued);
            read_frags(a0, b0, g);
            mfma_all(a0, b0);
            finish_kt();
        }
    }
}


}  // namespace

static int g_force_small = 0;
// test/tuning hook: 0 = auto, 1 = 128x128 kernel, 2 = persistent 256x128, 3 = persistent 256x256, 4 = 256x256 8-phase; 1x/2x = diagnostics
extern "C" int ufm_debug_set_gemm_variant(int force_small) {
    g_force_small = force_small;
    return UFM_OK;
}

extern "C" int ufm_gemm_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, int M, int N, int K,
                             const float* bias, int act, const float* gamma, const float* res, int ldres,
                             int res_row_mod, void* out, int out_dtype, int ldo, int out_row_group,
                             void* stream) {
    UFM_REQUIRE(A && W && out, "ufm_gemm_bf16: null pointer");
    UFM_REQUIRE(M > 0 && N > 0 && K > 0, "ufm_gemm_bf16: empty problem M=%d N=%d K=%d", M, N, K);
    UFM_REQUIRE(K % BK == 0, "ufm_gemm_bf16: K=%d must be a multiple of %d", K, BK);
    UFM_REQUIRE(N % BN == 0, "ufm_gemm_bf16: N=%d must be a multiple of %d", N, BN);
    UFM_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K, "ufm_gemm_bf16: bad lda/ldw %d/%d", lda, ldw);
    UFM_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, "ufm_gemm_bf16: A/W must be 16-byte aligned");
    UFM_REQUIRE(ldo % 4 == 0 && ldo >= N && (!res || (ldres % 4 == 0 && ldres >= N)), "ufm_gemm_bf16: bad ldo/ldres");
    UFM_REQUIRE(out_dtype == UFM_F32 || out_dtype == UFM_BF16, "ufm_gemm_bf16: bad out_dtype %d", out_dtype);
    UFM_REQUIRE((size_t)M * (size_t)(lda > ldo ? lda : ldo) < (1ull << 40), "ufm_gemm_bf16: problem too large");
    GemmArgs p{A, W, bias, gamma, res, out, lda, ldw, M, N, K, act, ldres, res_row_mod, ldo, out_row_group,
               g_force_small >= 40 ? g_force_small - 40 : g_force_small >= 20 ? g_force_small - 20 : (g_force_small >= 10 ? g_force_small - 10 : 0), 0};
    constexpr int NCU = 256;
    const int ntm256 = (M + PBM - 1) / PBM;
    const int t256 = (N % 256 == 0) ? ntm256 * (N / 256) : 0;
    const int t128 = ntm256 * (N / 128);
    // variant: 0 auto, 1 = 128x128, 2 = persistent 256x128, 3 = persistent 256x256, 4 = 256x256 8-phase,
    //          5 = hybrid (8-phase on the leading rows that fill whole rounds of 256 CUs, 128x128 on the rest)
    int variant = g_force_small >= 40 ? 4 : g_force_small >= 20 ? 1 : (g_force_small >= 10 ? 3 : g_force_small);
    const bool fits32 = (long long)M * lda < (1ll << 31) && (long long)N * ldw < (1ll << 31);
    const bool ok8 = t256 > 0 && K >= 128 && fits32;
    int m_split = 0;  // rows [0, m_split) -> 8-phase kernel, [m_split, M) -> 128x128 kernel
    if (variant == 0 || variant == 5) {
        // Cost model in units of one 8-phase tile (T4): a round of the 128x128 kernel (512 co-resident tiles of a
        // quarter of the work) measured 0.62 T4, a second launch ~0.1 T4 (tools/gemm_ab.py, profiles/r01/gemm_ab_*.log).
        // The 8-phase kernel only where it measured faster at all: wide N, K >= 512.
        const int ntn = N / 256;
        auto rounds1 = [&](int rows) { return rows <= 0 ? 0.0 : 0.62 * (double)(((long long)((rows + 127) / 128) * (N / 128) + 511) / 512); };
        double best = rounds1(M);
        int best_variant = 1;
        if (ok8 && ((N >= 1024 && K >= 512) || (N >= 768 && K >= 2048) || variant == 5)) {
            const double c4 = (double)((t256 + NCU - 1) / NCU);
            if (c4 < best) best = c4, best_variant = 4;
            const int full = t256 / NCU;                   // whole rounds of the 8-phase kernel
            const int rows_main = full * NCU / ntn * 256;  // leading rows whose tiles fit in them
            if (full > 0 && rows_main < M) {
                const double c5 = full + rounds1(M - rows_main) + 0.1;
                if (c5 < best || variant == 5) best = c5, best_variant = 5, m_split = rows_main;
            }
        }
        variant = (variant == 5 && best_variant != 5) ? (ok8 ? 4 : 1) : best_variant;
    }
    if ((variant == 4 || variant == 5) && !ok8) variant = 1;
    if (variant == 3 && t256 == 0) variant = 2;
    if (!fits32) variant = 1;
    auto launch128 = [&](const GemmArgs& q) {
        const int ntm = (q.M - q.m_begin + BM - 1) / BM, ntn = N / BN;
        dim3 grid(ntm * ntn), block(256);
        if (out_dtype == UFM_BF16)
            hipLaunchKernelGGL(gemm_bf16_kernel<1>, grid, block, 0, (hipStream_t)stream, q);
        else
            hipLaunchKernelGGL(gemm_bf16_kernel<0>, grid, block, 0, (hipStream_t)stream, q);
    };
    if (variant == 5) {
        GemmArgs lead = p, rest = p;
        lead.M = m_split;
        rest.m_begin = m_split;
        ufm_launch_gemm_8ph(lead, out_dtype, (hipStream_t)stream);
        launch128(rest);
    } else if (variant == 4) {
        ufm_launch_gemm_8ph(p, out_dtype, (hipStream_t)stream);
    } else if (variant == 3) {
        dim3 grid(t256 < NCU ? t256 : NCU), block(512);
        if (out_dtype == UFM_BF16)
            hipLaunchKernelGGL((gemm_bf16_persistent<256, 1>), grid, block, 0, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL((gemm_bf16_persistent<256, 0>), grid, block, 0, (hipStream_t)stream, p);
    } else if (variant == 2) {
        dim3 grid(t128 < NCU ? t128 : NCU), block(512);
        if (out_dtype == UFM_BF16)
            hipLaunchKernelGGL((gemm_bf16_persistent<128, 1>), grid, block, 0, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL((gemm_bf16_persistent<128, 0>), grid, block, 0, (hipStream_t)stream, p);
    } else {
        launch128(p);
    }
    UFM_CHECK_LAUNCH("ufm_gemm_bf16");
    return UFM_OK;
}
