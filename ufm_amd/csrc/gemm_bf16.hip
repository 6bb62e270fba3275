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
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB

struct GemmArgs {
    const uint16_t* A;
    const uint16_t* W;
    const float* bias;
    const float* gamma;
    const float* res;
    void* out;
    int lda, ldw, M, N, K;
    int act, ldres, res_row_mod, ldo, out_row_group;
    int debug;  // diagnostics only (tools/): 1 = every block reads tile (0,0), 2 = no DMA
};

// lane holds out[row][nb..nb+3] for each (n_rep, m_rep) tile of its wave's 64x64 block
template <int OUT_BF16>
__device__ __forceinline__ void epilogue(const GemmArgs& p, f32x4 (&acc)[4][4], int row0, int col0, int fr, int fq) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int row = row0 + m * 16 + fr;
        if (row >= p.M) continue;
        const int rrow = (p.res_row_mod > 0) ? (row % p.res_row_mod) : row;
        const int orow = (p.out_row_group > 0)
                             ? (row / p.out_row_group) * (p.out_row_group + 1) + 1 + row % p.out_row_group
                             : row;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int nb = col0 + n * 16 + fq * 4;
            f32x4 v = acc[n][m];
            if (p.bias) {
                const f32x4 bv = *(const f32x4*)(p.bias + nb);
                v += bv;
            }
            if (p.act == UFM_ACT_GELU && OUT_BF16) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = gelu_erf_fast(v[j]);
            } else if (p.act != UFM_ACT_NONE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], p.act);
            }
            if (p.gamma) {
                const f32x4 gv = *(const f32x4*)(p.gamma + nb);
                v *= gv;
            }
            if (p.res) {
                const f32x4 rv = *(const f32x4*)(p.res + (size_t)rrow * p.ldres + nb);
                v += rv;
            }
            if (OUT_BF16) {
                u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                *(u32x2*)((uint16_t*)p.out + (size_t)orow * p.ldo + nb) = pk;
            } else {
                *(f32x4*)((float*)p.out + (size_t)orow * p.ldo + nb) = v;
            }
        }
    }
}


template <int OUT_BF16>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntn = p.N / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / ntn, tn = bid - tm * ntn;
    const int m0 = tm * BM, n0 = tn * BN;
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

    epilogue<OUT_BF16>(p, acc, m0 + wr * 64, n0 + wc * 64, fr, fq);
}

// ---------------------------------------------------------------------------------------------
// Large-problem kernel: 256x128x64 tile, 8 waves (4 along M x 2 along N, 64x64 each, 2 waves per
// SIMD), 3-stage LDS ring (144 KiB, one block per CU).  The DMA for K-tile t+2 is issued at the top
// of iteration t, so two tiles of loads are in flight while tile t is consumed; the only wait in
// the loop is a COUNTED s_waitcnt vmcnt(6) (= this wave's 6 DMA pieces of tile t+1 may stay in
// flight) followed by one raw s_barrier per K-tile -- never vmcnt(0), never __syncthreads()
// (cdna_hip_programming.md "Pipelining across barriers").  RAW: a wave reads stage t%3 only after
// every wave's vmcnt wait for tile t and the barrier; WAR: stage (t+2)%3 == (t-1)%3 is restaged
// only after the barrier that follows every wave's last read of tile t-1.
// ---------------------------------------------------------------------------------------------
constexpr int LBM = 256, LBN = 128;
constexpr int LSTAGE = (LBM + LBN) * BK * 2;  // 48 KiB

template <int OUT_BF16>
__global__ __launch_bounds__(512, 2) void gemm_bf16_kernel_256x128(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[3 * LSTAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntn = p.N / LBN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = bid / ntn, tn = bid - tm * ntn;
    const int m0 = tm * LBM, n0 = tn * LBN;
    const int nk = p.K / BK;

    const int srow = lane >> 3, slot = lane & 7;
    const uint16_t* ga[4];
    const uint16_t* gw[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + srow;  // A tile row 0..255
        const int ar = min(m0 + r, p.M - 1);
        ga[i] = p.A + (size_t)ar * p.lda + (slot ^ (r & 7)) * 8;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (wave * 2 + i) * 8 + srow;  // W tile row 0..127
        gw[i] = p.W + (size_t)(n0 + r) * p.ldw + (slot ^ (r & 7)) * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* sa = smem + buf * LSTAGE + wave * 4096;
        char* sb = smem + buf * LSTAGE + LBM * BK * 2 + wave * 2048;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(ga[i] + kt * BK), LDS_PTR(sa + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gw[i] + kt * BK), LDS_PTR(sb + i * 1024), 16, 0, 0);
    };

    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    int a_off[4], b_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a_off[i] = (wr * 64 + i * 16 + fr) * 128;
        b_off[i] = LBM * BK * 2 + (wc * 64 + i * 16 + fr) * 128;
    }
    const int sw = fr & 7;

    f32x4 acc[4][4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    if (nk > 1) stage(1, 1);
    int cur = 0;  // t % 3
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk)
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) stage(cur == 0 ? 2 : cur - 1, kt + 2);  // (kt+2)%3
        const char* s = smem + cur * LSTAGE;
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
        cur = cur == 2 ? 0 : cur + 1;
    }
    epilogue<OUT_BF16>(p, acc, m0 + wr * 64, n0 + wc * 64, fr, fq);
}


// ---------------------------------------------------------------------------------------------
// 256x256x32 kernel for the large GEMMs.  PMC profiling of the kernels above showed waves parked
// 47 % of the time on the DMA wait: the global->LDS fill rate (~30-70 GB/s per CU), not the matrix
// pipe, bounds a tile whose operands are streamed through LDS.  A 256x256 tile needs half the fill
// bytes per FLOP of 256x128 and a quarter of 128x128.
//   * 8 waves as 2 (M) x 4 (N): 128x64 per wave = 8x4 MFMA tiles, 128 accumulator VGPRs.
//   * K-step 32 (64-byte LDS rows, chunk XOR g[(row>>2)&3], g={0,2,3,1}: conflict-free b128 reads),
//     4-stage ring of 32 KiB: tiles t+1..t+3 are in flight while tile t is consumed; the loop waits
//     with a counted vmcnt(8) (this wave's 4+4 DMA pieces of the next two tiles stay in flight) and
//     one raw s_barrier per K-step.
//   * grouped rasterization: 32 consecutive logical tiles (one XCD's worth) cover 4 M-panels x 8
//     N-panels, so every A panel is reused 8x and every W panel 4x out of that XCD's L2.
// ---------------------------------------------------------------------------------------------
constexpr int HBM_ = 256, HBN = 256, HBK = 32;
constexpr int HSTAGE = (HBM_ + HBN) * HBK * 2;  // 32 KiB
constexpr int GROUP_M = 4;

__device__ __forceinline__ int swz64(int row) { return (0x78 >> (((row >> 2) & 3) << 1)) & 3; }

template <int OUT_BF16>
__global__ __launch_bounds__(512, 2) void gemm_bf16_kernel_256x256(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * HSTAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntn = p.N / HBN, ntm = (p.M + HBM_ - 1) / HBM_;
    // grouped rasterization on top of the XCD chunking
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int per_group = GROUP_M * ntn;
    const int g = lid / per_group, in_g = lid - g * per_group;
    const int gm = min(GROUP_M, ntm - g * GROUP_M);  // rows in this (possibly last, shorter) group
    const int tm = g * GROUP_M + in_g % gm, tn = in_g / gm;
    const int m0 = (p.debug == 1) ? 0 : tm * HBM_, n0 = (p.debug == 1) ? 0 : tn * HBN;
    const int nk = p.K / HBK;

    // DMA pieces: 1 KiB = 16 rows of 64 B.  wave w: A rows [32w, 32w+32), W rows [32w, 32w+32).
    const int srow = lane >> 2, slot = lane & 3;
    const uint16_t* ga[2];
    const uint16_t* gw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = wave * 32 + i * 16 + srow;
        const int ar = min(m0 + r, p.M - 1);
        ga[i] = p.A + (size_t)ar * p.lda + (slot ^ swz64(r)) * 8;
        gw[i] = p.W + (size_t)(n0 + r) * p.ldw + (slot ^ swz64(r)) * 8;
    }
    auto stage = [&](int buf, int kt) {
        if (p.debug == 2) return;
        char* sa = smem + buf * HSTAGE + wave * 2048;
        char* sb = sa + HBM_ * HBK * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(ga[i] + kt * HBK), LDS_PTR(sa + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(gw[i] + kt * HBK), LDS_PTR(sb + i * 1024), 16, 0, 0);
        }
    };

    const int wr = wave >> 2, wc = wave & 3;  // 2 x 4
    const int fr = lane & 15, fq = lane >> 4;
    int a_off[8], b_off[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = wr * 128 + i * 16 + fr;
        a_off[i] = r * 64 + ((fq ^ swz64(r)) << 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wc * 64 + i * 16 + fr;
        b_off[i] = HBM_ * HBK * 2 + r * 64 + ((fq ^ swz64(r)) << 4);
    }

    f32x4 acc[4][8];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Software pipeline inside each wave: the fragments of K-tile t+1 are read from LDS while the
    // MFMAs of K-tile t run (two named register sets, statically indexed -- rule 20), so LDS latency
    // never sits between a barrier and the first MFMA.  Iteration kt:
    //   wait(tile kt+1 landed) ; barrier ; DMA(tile kt+4) ; ds_read(tile kt+1) || MFMA(tile kt)
    // Ring of 4 stages: the barrier of iteration kt follows every wave's ds_reads of tile kt (issued in
    // iteration kt-1 and consumed by its own MFMAs before it reaches this barrier?  no -- they are only
    // ISSUED; a wave passes lgkmcnt(0) for them before its MFMAs of iteration kt, i.e. before the
    // barrier of iteration kt+1), so the stage of tile kt is restaged at iteration kt+1 at the earliest:
    // DMA(tile kt+4) at iteration kt targets stage (kt+4)&3 == kt&3 ... which tile kt's reads may still
    // be using.  Therefore the DMA look-ahead is 3 tiles (stage (kt+3)&3 == (kt-1)&3, whose reads were
    // waited for before the MFMAs of iteration kt-1, before this barrier).
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    if (nk > 2) stage(2, 2);
    bf16x8 a0[8], b0[4], a1[8], b1[4];
    auto read_frags = [&](bf16x8 (&a)[8], bf16x8 (&b)[4], int kt) {
        const char* s = smem + (kt & 3) * HSTAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = *(const bf16x8*)(s + b_off[i]);
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = *(const bf16x8*)(s + a_off[i]);
    };
    auto wait_tile = [&](int kt) {  // this wave's DMA pieces of tile kt have landed (later tiles may be in flight)
        const int younger = min(nk - 1, kt + 2) - kt;  // tiles issued after kt so far: kt+1, kt+2 (kt+3 is issued after the barrier)
        if (younger >= 2)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto mfma_all = [&](bf16x8 (&a)[8], bf16x8 (&b)[4]) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[n], a[m], acc[n][m], 0, 0, 0);
    };
    wait_tile(0);
    __builtin_amdgcn_s_barrier();
    read_frags(a0, b0, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int kt = 0; kt < nk; kt += 2) {
        // ---- even half: MFMA(kt) from set 0, prefetch set 1 <- tile kt+1 ----
        if (kt + 1 < nk) {
            wait_tile(kt + 1);
            __builtin_amdgcn_s_barrier();
            if (kt + 3 < nk) stage((kt + 3) & 3, kt + 3);
            read_frags(a1, b1, kt + 1);
        }
        mfma_all(a0, b0);
        __builtin_amdgcn_sched_barrier(0);   // keep the wait BELOW the MFMAs (the scheduler otherwise hoists it)
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): set 1 landed long ago; compiler-visible so no wait is re-inserted at the loop head
        if (kt + 1 >= nk) break;
        // ---- odd half: MFMA(kt+1) from set 1, prefetch set 0 <- tile kt+2 ----
        if (kt + 2 < nk) {
            wait_tile(kt + 2);
            __builtin_amdgcn_s_barrier();
            if (kt + 4 < nk) stage((kt + 4) & 3, kt + 4);
            read_frags(a0, b0, kt + 2);
        }
        mfma_all(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);
    }
    // epilogue in two 64-row halves (reuses the 64x64 epilogue)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 part[4][4];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) part[n][m] = acc[n][h * 4 + m];
        epilogue<OUT_BF16>(p, part, m0 + wr * 128 + h * 64, n0 + wc * 64, fr, fq);
    }
}

}  // namespace

static int g_force_small = 0;
// test/tuning hook: 0 = auto, 1 = 128x128 kernel, 2 = 256x128 kernel, 3 = 256x256 kernel (A/B timing in one process)
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
    GemmArgs p{A, W, bias, gamma, res, out, lda, ldw, M, N, K, act, ldres, res_row_mod, ldo, out_row_group, g_force_small >= 10 ? g_force_small - 10 : 0};
    // big tile when it fills the chip (>= 256 tiles of 256x128), else the 128x128 kernel (2 blocks/CU)
    const int big_tiles = ((M + LBM - 1) / LBM) * (N / LBN);
    const int huge_tiles = (N % HBN == 0) ? ((M + HBM_ - 1) / HBM_) * (N / HBN) : 0;
    // 256x256 pays off only when its wave quantization is good: tiles / (rounds * 256 CUs) >= 0.88
    const bool huge_ok = huge_tiles >= 192 && huge_tiles * 100 >= 88 * (((huge_tiles + 255) / 256) * 256);
    if ((huge_ok && g_force_small == 0) || (huge_tiles > 0 && (g_force_small == 3 || g_force_small >= 10))) {
        dim3 grid(huge_tiles), block(512);
        if (out_dtype == UFM_BF16)
            hipLaunchKernelGGL(gemm_bf16_kernel_256x256<1>, grid, block, 0, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL(gemm_bf16_kernel_256x256<0>, grid, block, 0, (hipStream_t)stream, p);
    } else if (big_tiles >= 256 && g_force_small != 1) {
        dim3 grid(big_tiles), block(512);
        if (out_dtype == UFM_BF16)
            hipLaunchKernelGGL(gemm_bf16_kernel_256x128<1>, grid, block, 0, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL(gemm_bf16_kernel_256x128<0>, grid, block, 0, (hipStream_t)stream, p);
    } else {
        const int ntm = (M + BM - 1) / BM, ntn = N / BN;
        dim3 grid(ntm * ntn), block(256);
        if (out_dtype == UFM_BF16)
            hipLaunchKernelGGL(gemm_bf16_kernel<1>, grid, block, 0, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL(gemm_bf16_kernel<0>, grid, block, 0, (hipStream_t)stream, p);
    }
    UFM_CHECK_LAUNCH("ufm_gemm_bf16");
    return UFM_OK;
}
