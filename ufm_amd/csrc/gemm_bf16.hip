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

// KS = K-tiles per barrier.  KS = 1: two 32-KiB stages, two co-resident blocks per CU (they hide each other's DMA waits).
// KS = 2 (grids of at most one block per CU -- the one-pair shapes): a lone block spends a K-step on its own serial chain
// {vmcnt(0), barrier, LDS reads, 32 MFMAs} (~0.75 us against 0.24 us of MFMA issue), so it takes two K-tiles per barrier:
// half the barriers, twice the MFMAs behind each (128 KiB of LDS).  Every accumulator sees the same MFMA sequence in the
// same K order as with KS = 1: results are bit-identical.
template <int OUT_BF16, int KS = 1>
__global__ __launch_bounds__(256, KS == 1 ? 2 : 1) void gemm_bf16_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * KS * STAGE_BYTES];
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
        if (lab_get(p.debug, gemm_lab::NO_DMA)) return;  // ablation (tools/): no DMA
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

    // buffer of K-tile kt: group (kt / KS) & 1, slot kt % KS
    auto buf_of = [](int kt) { return ((kt / KS) & 1) * KS + kt % KS; };
#pragma unroll
    for (int j = 0; j < KS; ++j)
        if (j < nk) stage(buf_of(j), j);
    for (int kt = 0; kt < nk; kt += KS) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KS; ++j)
            if (kt + KS + j < nk) stage(buf_of(kt + KS + j), kt + KS + j);
        auto load_frags = [&](bf16x8 (&a)[4], bf16x8 (&b)[4], int j, int kk) {
            const char* s = smem + buf_of(kt + j) * STAGE_BYTES;
            const int coff = ((kk * 4 + fq) ^ sw) * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *(const bf16x8*)(s + a_off[i] + coff);
                b[i] = *(const bf16x8*)(s + b_off[i] + coff);
            }
        };
        auto mma16 = [&](const bf16x8 (&a)[4], const bf16x8 (&b)[4]) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[n], a[m], acc[n][m], 0, 0, 0);
        };
        if constexpr (KS == 2) {
            // The lone block (at most one per CU: nothing else covers its LDS latency) reads the fragments of k-substep s + 1 while
            // the 16 MFMAs of substep s issue: two register sets, 512 registers per wave at one block per CU.  The MFMA sequence of
            // every accumulator is unchanged (k order 0, 1, 2, ...): bit-identical to KS = 1.
            bf16x8 a0[4], b0[4], a1[4], b1[4];
            const bool two = kt + 1 < nk;
            load_frags(a0, b0, 0, 0);
            load_frags(a1, b1, 0, 1);
            mma16(a0, b0);
            if (two) load_frags(a0, b0, 1, 0);
            mma16(a1, b1);
            if (two) {
                load_frags(a1, b1, 1, 1);
                mma16(a0, b0);
                mma16(a1, b1);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 a[4], b[4];
                load_frags(a, b, 0, kk);
                mma16(a, b);
            }
        }
    }

    if (lab_get(p.debug, gemm_lab::NO_EPILOGUE)) {  // ablation (tools/): no epilogue traffic; keep the accumulators live
        float keep = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) keep += acc[n][m][0] + acc[n][m][1] + acc[n][m][2] + acc[n][m][3];
        if (keep == 123.456f) ((float*)p.out)[0] = keep;
        return;
    }
    if (lab_get(p.debug, gemm_lab::DIRECT_EPILOGUE)) {  // A/B hook: the direct (unstaged) epilogue
        epilogue<OUT_BF16>(p, acc, m0 + wr * 64, n0 + wc * 64, fr, fq);
        return;
    }
    __syncthreads();  // every wave is done reading the staging buffers
    if constexpr (!OUT_BF16) {
        // the transformer's read-modify-write form on a whole slice: residual loads out of the store chain (gemm_common.h, round 5)
        if (p.res && p.bias && p.act == UFM_ACT_NONE && p.res_row_mod == 0 && p.out_row_group == 0 && m0 + wr * 64 + 64 <= p.M && !lab_get(p.debug, gemm_lab::SERIAL_RMW)) {
            epilogue_lds_rmw2<0, 5>(p, acc, acc, smem + wave * 16384, m0 + wr * 64, n0 + wc * 64, lane);
            return;
        }
    }
    epilogue_lds<OUT_BF16>(p, acc, smem + wave * 16384, m0 + wr * 64, n0 + wc * 64, lane);
}

}  // namespace

static int g_gemm_variant = 0;  // 0 auto, 1 = 128x128 kernel, 4 = 256x256 8-phase kernel, 5 = hybrid (8-phase on whole rounds + 128x128 on the rest)
static int g_gemm_flags = 0;    // diagnostics (tools/): 2 = no DMA, 4 = no epilogue traffic, 8 = direct (un-staged) epilogue of the 128x128 kernel
extern "C" int ufm_debug_set_gemm_variant(int variant) {
    if (variant != 0 && variant != 1 && variant != 4 && variant != 5 && variant != 6 && variant != 7) {
        ufm_set_error("ufm_debug_set_gemm_variant: %d is not one of 0 (auto), 1 (128x128), 4 (8-phase), 5 (hybrid), 6 (256x128 pair), 7 (persistent 8-phase)", variant);
        return UFM_ERR_ARG;
    }
    g_gemm_variant = variant;
    return UFM_OK;
}
static int g_gemm_tile_rows = 0;  // 0 = cost model; 160 / 192 / 224 / 256 pins the 8-phase tile height (tests, tools/lab)
extern "C" int ufm_debug_set_gemm_tile_rows(int rows) {
    if (rows != 0 && rows != 160 && rows != 192 && rows != 224 && rows != 256) {
        ufm_set_error("ufm_debug_set_gemm_tile_rows: %d is not one of 0 (auto), 160, 192, 224, 256", rows);
        return UFM_ERR_ARG;
    }
    g_gemm_tile_rows = rows;
    return UFM_OK;
}
// Diagnostics (tools/lab/gemm_stamps.py): while a buffer is set, the 8-phase / pair launches that have a stamped instantiation
// write one 8 x uint64 row per workgroup (gemm_common.h GemmStamps) for workgroups [0, rows); nullptr = off (the default).
static unsigned long long* g_gemm_stamps = nullptr;
static int g_gemm_stamp_rows = 0;
extern "C" int ufm_debug_set_gemm_stamps(unsigned long long* buf, int rows) {
    UFM_REQUIRE((buf == nullptr) == (rows == 0) && rows >= 0, "ufm_debug_set_gemm_stamps: buffer and row count must be given together");
    g_gemm_stamps = buf;
    g_gemm_stamp_rows = rows;
    return UFM_OK;
}
// Lab (round 6): per-stream workspaces for the deterministic 2-way split-K of the read-modify-write launches (proj / fc2).
// While a stream has a workspace, its fp32-residual launches with K >= min_k whose dispatch is full-height 8-phase tiles run the split
// form (gemm_bf16_8ph.hip, SK).  ws = NULL removes the stream's entry.  One workspace per CONCURRENT stream: launches on one stream are
// ordered, so they can share it.  ws layout: [16384 x u32 counters, zero][tiles x 2 x 256 KiB partial tiles].
struct SplitKWs {
    void* stream;
    char* ws;
    long long bytes;
    int min_k;
};
static SplitKWs g_splitk_ws[8] = {};
extern "C" int ufm_debug_set_gemm_splitk(void* stream, void* ws, long long bytes, int min_k) {
    UFM_REQUIRE(!ws || (bytes >= (1ll << 20) && ((uintptr_t)ws % 16) == 0 && min_k >= 256), "ufm_debug_set_gemm_splitk: a workspace of >= 1 MiB, 16-byte aligned, min_k >= 256");
    for (auto& e : g_splitk_ws)
        if (e.ws && e.stream == stream) e = SplitKWs{};
    if (!ws) return UFM_OK;
    for (auto& e : g_splitk_ws)
        if (!e.ws) {
            e = SplitKWs{stream, (char*)ws, bytes, min_k};
            return UFM_OK;
        }
    ufm_set_error("ufm_debug_set_gemm_splitk: more than 8 streams with a workspace");
    return UFM_ERR_ARG;
}
extern "C" int ufm_debug_set_gemm_flags(int flags) {
    // fields: lab_flags.h gemm_lab::ALL (one table; disjoint at compile time); a bit outside the table is refused, not dropped
    const unsigned unknown = (unsigned)flags & ~lab_known(gemm_lab::ALL);
    UFM_REQUIRE(unknown == 0, "ufm_debug_set_gemm_flags: bits 0x%x belong to no field of the lab flag table (lab_flags.h)", unknown);
    g_gemm_flags = flags;
    return UFM_OK;
}

extern "C" int ufm_gemm_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, int M, int N, int K,
                             const float* bias, int act, const float* gamma, const float* res, int ldres,
                             int res_row_mod, void* out, int out_dtype, int ldo, int out_row_group,
                             void* stream) {
    return ufm_gemm_bf16_rope(A, lda, W, ldw, M, N, K, bias, act, gamma, res, ldres, res_row_mod, out, out_dtype, ldo, out_row_group,
                              nullptr, nullptr, 0, 0, stream);
}

// The same GEMM with RoPE-2D fused into the epilogue (the north star's "fused QKV+RoPE"): columns [0, rope_cols) of the bf16
// output -- the q and k heads of a QKV / Q / KV projection -- are rotated on the fp32 accumulator (after bias and gamma,
// before the bf16 rounding) with the tables [rope_mod][64] of rope.hip, row % rope_mod = the token's index in its image.
extern "C" int ufm_gemm_bf16_rope(const uint16_t* A, int lda, const uint16_t* W, int ldw, int M, int N, int K,
                                  const float* bias, int act, const float* gamma, const float* res, int ldres,
                                  int res_row_mod, void* out, int out_dtype, int ldo, int out_row_group,
                                  const float* rope_cos, const float* rope_sin, int rope_mod, int rope_cols, void* stream) {
    UFM_REQUIRE(A && W && out, "ufm_gemm_bf16: null pointer");
    if (rope_cos || rope_sin) {
        UFM_REQUIRE(rope_cos && rope_sin && rope_mod > 0 && rope_cols > 0 && rope_cols % 64 == 0 && rope_cols <= N, "ufm_gemm_bf16_rope: bad RoPE arguments (mod=%d cols=%d)", rope_mod, rope_cols);
        UFM_REQUIRE(out_dtype == UFM_BF16 && !res && out_row_group == 0 && act == UFM_ACT_NONE && ldo % 8 == 0 && ((uintptr_t)out % 16) == 0,
                    "ufm_gemm_bf16_rope: the fused rotation needs a plain bf16 output (no residual, activation or row re-mapping)");
    }
    UFM_REQUIRE(M > 0 && N > 0 && K > 0, "ufm_gemm_bf16: empty problem M=%d N=%d K=%d", M, N, K);
    UFM_REQUIRE(K % BK == 0, "ufm_gemm_bf16: K=%d must be a multiple of %d", K, BK);
    UFM_REQUIRE(N % BN == 0, "ufm_gemm_bf16: N=%d must be a multiple of %d", N, BN);
    UFM_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K, "ufm_gemm_bf16: bad lda/ldw %d/%d", lda, ldw);
    UFM_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, "ufm_gemm_bf16: A/W must be 16-byte aligned");
    UFM_REQUIRE(ldo % 4 == 0 && ldo >= N && (!res || (ldres % 4 == 0 && ldres >= N)), "ufm_gemm_bf16: bad ldo/ldres");
    UFM_REQUIRE(out_dtype == UFM_F32 || out_dtype == UFM_BF16, "ufm_gemm_bf16: bad out_dtype %d", out_dtype);
    UFM_REQUIRE((size_t)M * (size_t)(lda > ldo ? lda : ldo) < (1ull << 40), "ufm_gemm_bf16: problem too large");
    GemmArgs p{A, W, bias, gamma, res, out, lda, ldw, M, N, K, act, ldres, res_row_mod, ldo, out_row_group, g_gemm_flags, 0, rope_cos, rope_sin, rope_mod, rope_cols, g_gemm_stamps, g_gemm_stamp_rows};
    if (lab_get(g_gemm_flags, gemm_lab::LDA0)) p.lda = 0;
    if (lab_get(g_gemm_flags, gemm_lab::LDW0)) p.ldw = 0;
    const int NCU = ufm_device_cu_count();  // whole rounds of the one-block-per-CU 8-phase kernel
    const int ntn = N / 256;
    // variant: 0 auto, 1 = 128x128, 4 = 8-phase on all rows, 5 = hybrid (256-row 8-phase tiles on the leading rows that fill
    //          whole rounds of the chip's CUs, the rest by whichever is cheaper: lower 8-phase tiles or the 128x128 kernel)
    int variant = g_gemm_variant;
    const bool fits32 = (long long)M * lda < (1ll << 31) && (long long)N * ldw < (1ll << 31);
    const bool ok8 = N % 256 == 0 && K >= 128 && fits32;
    // Cost model in units of one 256-row 8-phase tile (T4), fitted to tools/lab/gemm_tile_rows.py on the UFM shapes
    // (profiles/r02/gemm_tile_rows.log): a full round of the 128x128 kernel (512 co-resident tiles of a quarter of the work)
    // = 0.62 T4 (0.85 at K = 4096), a partly filled last round 0.8..1 of that, a second launch 0.3 T4.  An 8-phase tile of 32 nf rows costs
    // C8_FIX + (1 - C8_FIX) nf / 8: the phase's L slot (LDS reads, DMA issue, two barriers) does not shrink with the MFMA
    // count, the epilogue does -- so the fixed share depends on the epilogue (GELU + bf16 store: 0.55, fp32 read-modify-write:
    // 0.7, plain bf16 store: 0.9).
    const double C8_FIX = (act == UFM_ACT_GELU && out_dtype == UFM_BF16) ? 0.55 : (out_dtype == UFM_F32 ? 0.7 : 0.9);
    auto rounds1 = [&](int rows) {
        if (rows <= 0) return 0.0;
        const long long t = (long long)((rows + 127) / 128) * (N / 128), per = 2 * NCU;
        const long long full = t / per, rem = t % per;
        const double w1 = 0.62 + 0.23 * (K <= 1024 ? 0.0 : K >= 4096 ? 1.0 : (K - 1024) / 3072.0);  // its main loop is the slower one: long K costs more
        return w1 * ((double)full + (rem ? 0.8 + 0.2 * (double)rem / (double)per : 0.0));
    };
    auto cost8 = [&](int rows, int nf) {
        const int h = 32 * nf;
        const long long tiles = (long long)((rows + h - 1) / h) * ntn;
        return 0.08 + 0.92 * (double)((tiles + NCU - 1) / NCU) * (C8_FIX + (1.0 - C8_FIX) * nf / 8.0);  // rounds after the first pipeline a little
    };
    auto best8 = [&](int rows, int& nf_out) {  // cheapest tile height for `rows` rows
        double best = 1e30;
        // The cost model prices ONE launch alone on the chip (its latency).  On a stream the caller has flagged as one of several concurrent ones
        // (ufm_hint_concurrent_stream: the engine's micro-batch streams) what counts is the CU time a launch consumes: 232 tiles of 192 rows finish
        // a lone proj 7 % sooner than 172 tiles of 256 rows, but occupy 25 % more CU time that the other stream's kernels could have used.  There,
        // from 8192 rows on, only full-height tiles are considered: +1.0...+2.4 % pairs/s in the pipeline, -0.8 % on a single stream -- hence the
        // hint (profiles/r05/gemm_tile_policy_pipeline.log).  Flag bits 29 / 30 (lab): the latency / the CU-time objective on every stream.
        const bool throughput = lab_get(g_gemm_flags, gemm_lab::CU_TIME) || (ufm_stream_is_concurrent(stream) && !lab_get(g_gemm_flags, gemm_lab::LATENCY));
        const bool only8 = rows >= 8192 && throughput;
        for (int nf = 8; nf >= 5; --nf) {
            if (g_gemm_tile_rows && 32 * nf != g_gemm_tile_rows) continue;
            if (only8 && !g_gemm_tile_rows && nf != 8) continue;
            const double c = cost8(rows, nf);
            if (c < best - 1e-9) best = c, nf_out = nf;
        }
        return best;
    };
    int m_split = 0, nf_lead = 8, nf_rest = 0;  // rows [0, m_split): 8-phase at nf_lead; [m_split, M): 8-phase at nf_rest, or 128x128 if 0
    if (variant == 0 || variant == 5) {
        double best = rounds1(M);
        int best_variant = 1;
        // the 8-phase kernel only where it measured faster at all: wide N, K >= 512
        if (ok8 && ((N >= 1024 && K >= 512) || (N >= 768 && K >= 768) || variant == 5)) {
            int nf = 8;
            const double c4 = best8(M, nf);
            if (c4 < best) best = c4, best_variant = 4, nf_lead = nf;
            const int full = (int)(((long long)((M + 255) / 256) * ntn) / NCU);  // whole rounds of 256-row tiles
            const int rows_main = full * NCU / ntn * 256;                         // leading rows whose tiles fit in them
            if (full > 0 && rows_main < M && !g_gemm_tile_rows) {
                int nfr = 8;
                const double r8 = best8(M - rows_main, nfr), r1 = rounds1(M - rows_main);
                const double c5 = 0.08 + 0.92 * full + (r8 < r1 ? r8 : r1) + 0.3;
                if (c5 < best || variant == 5) best = c5, best_variant = 5, m_split = rows_main, nf_lead = 8, nf_rest = r8 < r1 ? nfr : 0;
            }
        }
        variant = (variant == 5 && best_variant != 5) ? (ok8 ? 4 : 1) : best_variant;
    } else if (variant == 4 && g_gemm_tile_rows) {
        nf_lead = g_gemm_tile_rows / 32;
    }
    if ((variant == 4 || variant == 5) && !ok8) variant = 1;
    // The 256x128 two-resident-workgroups kernel (gemm_bf16_pair.hip) on the shapes where it measured faster than the choice above, isolated behind a cold
    // cache (tools/lab/gemm_pair_ab.py, profiles/r05/gemm_pair_ab.log): the info-sharing widths (D = 768: three or nine 256-column tiles quantise badly on
    // 256 CUs) and the encoder's QKV at micro-batch row counts.  In the two-stream pipeline the three rules together are worth +1.7 % pairs/s (35.49 vs
    // 36.11 ms, 12 interleaved rounds; profiles/r05/gemm_pair_pipeline_ab.log -- the first pipeline A/B of round 5 called them neutral because its flagged
    // arm also ran the 8-phase kernel column-major: the grouped-rasterization height read every flag bit above 8).  Flag bits 24..27 flip the rules (A/B).
    int nf_pair = g_gemm_tile_rows ? g_gemm_tile_rows / 32 : 8;
    if (g_gemm_variant == 0 && K >= 128 && fits32 && !g_gemm_tile_rows) {
        constexpr int PAIR_DEFAULT = 7;
        const int pol = PAIR_DEFAULT ^ lab_get(g_gemm_flags, gemm_lab::PAIR_FLIP);
        const bool small = M < 16000;
        int nfp = 0;
        if (out_dtype == UFM_BF16) {
            if ((pol & 1) && K == 768 && (N == 2304 || (N == 3072 && small))) nfp = small ? 7 : 8;
            if ((pol & 4) && K == 1024 && N == 3072 && small) nfp = 7;
        } else if (res) {
            if ((pol & 2) && N == 768 && (K == 768 || !small)) nfp = 6;
        }
        if ((pol & 8) && !nfp) nfp = 8;
        if (nfp) variant = 6, nf_pair = nfp;
    }
    if (variant == 6 && !(K >= 128 && fits32)) variant = 1;
    auto launch128 = [&](const GemmArgs& q) {
        const int ntm = (q.M - q.m_begin + BM - 1) / BM, ntn128 = N / BN;
        dim3 grid(ntm * ntn128), block(256);
        // at most one block per CU anyway: two K-tiles per barrier (bit-identical; g_gemm_flags & 128 = never, for A/B)
        const bool lone = (int)grid.x <= NCU && K >= 4 * BK && !lab_get(g_gemm_flags, gemm_lab::NO_TWO_KTILES);
        if (out_dtype == UFM_BF16) {
            if (lone) hipLaunchKernelGGL((gemm_bf16_kernel<1, 2>), grid, block, 0, (hipStream_t)stream, q);
            else hipLaunchKernelGGL((gemm_bf16_kernel<1, 1>), grid, block, 0, (hipStream_t)stream, q);
        } else {
            if (lone) hipLaunchKernelGGL((gemm_bf16_kernel<0, 2>), grid, block, 0, (hipStream_t)stream, q);
            else hipLaunchKernelGGL((gemm_bf16_kernel<0, 1>), grid, block, 0, (hipStream_t)stream, q);
        }
    };
    // the transformer's hot Linear forms get the epilogue specialised at compile time (gemm_common.h EpiTraits); flag 64 = generic
    int epi = 0;
    const bool plain_rows = res_row_mod == 0 && out_row_group == 0 && !rope_cos && !lab_get(g_gemm_flags, gemm_lab::GENERIC_EPILOGUE);
    if (plain_rows && bias && out_dtype == UFM_BF16 && !res && (ldo & 7) == 0 && ((uintptr_t)out & 15) == 0) {
        if (act == UFM_ACT_GELU && !gamma) epi = 1;
        else if (act == UFM_ACT_NONE && gamma) epi = 2;
    } else if (plain_rows && bias && out_dtype == UFM_F32 && res && act == UFM_ACT_NONE) {
        epi = gamma ? 3 : 4;
    }
    // Persistent 8-phase form (round 5, gemm_bf16_8ph_persist.hip): bf16-output launches of WHOLE 256-row tiles that make whole rounds of the
    // chip -- the lead part of the hybrid split, or a whole GEMM whose tile count is a multiple of the CU count.  Flag bit 28 = never (A/B).
    auto persist_able = [&](const GemmArgs& q) {  // what the kernel needs: a bf16-output compile-time epilogue, whole 256-row tiles
        const long long rows = q.M - q.m_begin;
        return (epi == 1 || epi == 2) && out_dtype == UFM_BF16 && ok8 && rows > 0 && rows % 256 == 0 && !q.stamps && !(g_gemm_flags & (lab_mask(gemm_lab::NO_DMA) | lab_mask(gemm_lab::NO_EPILOGUE) | lab_mask(gemm_lab::LDA0) | lab_mask(gemm_lab::LDW0) | lab_mask(gemm_lab::RASTER_GROUP) | lab_mask(gemm_lab::STAGGER)));
    };
    auto persist_ok = [&](const GemmArgs& q) {    // where it is dispatched: at least two rounds of the chip, the last one (all but) full
        const long long tiles = ((q.M - q.m_begin) / 256) * ntn;
        const long long short_of = (NCU - tiles % NCU) % NCU;  // the hybrid split's lead part is whole rounds less at most ntn - 1 tiles (whole tile ROWS)
        return persist_able(q) && tiles + short_of >= 2 * NCU && short_of < ntn && !lab_get(g_gemm_flags, gemm_lab::NO_PERSIST);
    };
    // lab split-K (ufm_debug_set_gemm_splitk): this stream has a workspace -> its read-modify-write launches of K >= min_k run as two K halves
    // per full-height 8-phase tile, whatever the cost model above chose (an A/B arm, not a dispatch rule)
    if (epi == 3 && ok8 && !p.stamps && K % 128 == 0) {
        for (const auto& e : g_splitk_ws) {
            const long long tiles = (long long)((M + 255) / 256) * (N / 256);
            if (e.ws && e.stream == stream && K >= e.min_k && tiles <= 16384 && 65536 + tiles * 2 * 262144 <= e.bytes) {
                GemmArgs q = p;
                q.splitk = 2, q.counters = (unsigned*)e.ws, q.slab = (float*)(e.ws + 65536);
                if (ufm_launch_gemm_8ph(q, out_dtype, (hipStream_t)stream, 8, epi) == 0) {
                    UFM_CHECK_LAUNCH("ufm_gemm_bf16");
                    return UFM_OK;
                }
            }
        }
    }
    if (variant == 7) {  // tests / tools: the persistent kernel wherever it can run at all (any tile count), else the 8-phase kernel
        if (persist_able(p)) {
            ufm_launch_gemm_8ph_persist(p, (hipStream_t)stream, epi, NCU);
            UFM_CHECK_LAUNCH("ufm_gemm_bf16");
            return UFM_OK;
        }
        variant = ok8 ? 4 : 1;
    }
    if (variant == 5) {
        GemmArgs lead = p, rest = p;
        lead.M = m_split;
        rest.m_begin = m_split;
        if (persist_ok(lead)) ufm_launch_gemm_8ph_persist(lead, (hipStream_t)stream, epi, NCU);
        else
        ufm_launch_gemm_8ph(lead, out_dtype, (hipStream_t)stream, 8, epi);
        if (nf_rest) ufm_launch_gemm_8ph(rest, out_dtype, (hipStream_t)stream, nf_rest, epi);
        else launch128(rest);
    } else if (variant == 4) {
        if (nf_lead == 8 && g_gemm_variant == 0 && persist_ok(p)) ufm_launch_gemm_8ph_persist(p, (hipStream_t)stream, epi, NCU);
        else
        ufm_launch_gemm_8ph(p, out_dtype, (hipStream_t)stream, nf_lead, epi);
    } else if (variant == 6) {
        ufm_launch_gemm_pair(p, out_dtype, (hipStream_t)stream, nf_pair, epi);
    } else {
        launch128(p);
    }
    UFM_CHECK_LAUNCH("ufm_gemm_bf16");
    return UFM_OK;
}
