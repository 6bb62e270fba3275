// Implicit-GEMM convolution on NHWC activations with fp32-class accuracy on the bf16 matrix cores:
// every value x is carried as a (hi, lo) pair of bf16 with hi = bf16(x), lo = bf16(x - hi), and each
// product is evaluated as hi*hi + hi*lo + lo*hi (3 x v_mfma_f32_16x16x32_bf16, fp32 accumulate);
// the dropped lo*lo term is 2^-16 relative, so a dot product carries ~2^-17 relative error --
// between TF32 (2^-11, what cuDNN gives the reference's "fp32 island" on NVIDIA by default) and
// exact fp32, at 5.3x the throughput of the exact-fp32 MFMA (3 x 32 cyc per 32 Kflop vs 512 cyc).
//
// The split is done ONCE by the producer: activations live in HBM as two bf16 planes
// [2][rows][C] (same bytes as fp32), weights are pre-split at pack time, so the main loop is a
// pure DMA -> LDS -> MFMA pipeline with no conversion VALU work.
//   * 128 x BN x 32 tile (BN 128/64/32), 4 waves, 16x16x32 MFMA tiles, 2 LDS stages, 2 blocks/CU.
//   * LDS rows are 64 B (32 channels of one tap); 16-byte chunk index XOR g[(row>>2)&3],
//     g = {0,2,3,1}: conflict-free ds_read_b128 for the 16-row operand pattern; applied to the DMA
//     source address (gather + zero halo via per-lane source, as in conv_f32.hip) and to the reads.
//   * ReLU-on-input uses the sign of hi for both halves (packed 16-bit ops on the fragments).
//   * epilogue: bias, act, two split residuals, split (hi, lo) store, pixel-shuffle store.
#include <mutex>

#include "conv_x3_common.h"

namespace {

constexpr int BK = 32;

// BM x BN tile: 128x{128,64,32} for the big maps; 64x64 for the small grids (19^2 / 37^2: with 128-row tiles only 46-172
// blocks exist, each a 72-216 step latency-bound K-loop -- 4x more, smaller blocks co-reside and overlap their DMA waits)
// NS = LDS ring depth.  NS = 2: one vmcnt(0) + __syncthreads() per K-step, two co-resident blocks per CU hide each
// other's DMA latency (big grids).  NS = 4 (BN >= 64): counted s_waitcnt vmcnt + raw s_barrier, two K-tiles stay in flight
// across every barrier -- used for the 64x64 tile on small grids with long K loops (the 19^2 layers: 46-184 blocks, K up
// to 6912), where nothing else on the CU covers the ~0.5 us of each K-step's DMA.  Same arithmetic order: results are bit-identical.
// PASSES = 3: hi*hi + hi*lo + lo*hi.  PASSES = 1: the hi planes only -- exactly a bf16 convolution with fp32 accumulation
// (operands rounded to bf16, what torch.autocast(bfloat16) gives the reference's UNet, ufm.py:915-917): the lo planes are
// neither staged nor read, one MFMA per fragment pair and K-tile.  Output format and epilogue are the same.
template <int BM, int BN, int NS, int PASSES = 3>
__global__ __launch_bounds__(256, NS == 2 ? 2 : 1) void conv_x3_kernel(ConvX3Args p) {
    static_assert(PASSES == 3 || PASSES == 1, "PASSES");
    constexpr int WN = BN >= 64 ? 2 : 1, WM = 4 / WN;
    constexpr int A_RPW = BM / 4, A_PIECES = A_RPW / 16;  // A rows / DMA pieces per wave and plane
    constexpr int TM = (BM / WM) / 16, TN = (BN / WN) / 16;
    constexpr int A_PLANE = BM * BK * 2, W_PLANE = BN * BK * 2;         // bytes per plane per stage
    constexpr int STAGE_BYTES = 2 * (A_PLANE + W_PLANE);
    constexpr int W_ROWS_PER_WAVE = BN / 4;                             // 32 / 16 / 8
    constexpr int W_PIECES = W_ROWS_PER_WAVE >= 16 ? W_ROWS_PER_WAVE / 16 : 1;
    static_assert(NS == 2 || BN >= 64, "the deep ring needs the same DMA piece count in every wave");
    constexpr int PIECES = (PASSES == 3 ? 2 : 1) * (A_PIECES + W_PIECES);  // global_load_lds per wave and K-tile (BN >= 64)
    __shared__ __attribute__((aligned(16))) char smem[NS * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntn = p.Cout / BN;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    int slice = 0;
    if (p.splitk > 1) {  // a tile's slices are consecutive logical ids: concurrent on one XCD, the reducer reads them from its L2
        slice = bid % p.splitk;
        bid /= p.splitk;
    }
    const int tmi_all = bid / ntn, tni = bid - tmi_all * ntn;
    int grp, tmi;
    conv_x3_group_of(p, tmi_all, (p.M - p.m_begin + BM - 1) / BM, grp, tmi);
    const uint16_t* const in_g = p.in + (size_t)grp * p.in_group;
    const uint16_t* const w_g = p.w + (size_t)grp * p.w_group;
    const int m0 = p.m_begin + tmi * BM, n0 = tni * BN;
    const int cpt = p.Cin / BK;
    const int ntaps = p.KH * p.KW;
    const int nk = ntaps * cpt;
    const size_t ktot = (size_t)p.KH * p.KW * p.Cin;

    // ---- staging: wave w owns tile rows [A_RPW*w, A_RPW*(w+1)) of A (pieces of 16 rows), both planes ----
    const int srow = lane >> 2, slot = lane & 3;
    int a_iy0[A_PIECES], a_ix0[A_PIECES], a_chunk[A_PIECES];
    size_t a_img[A_PIECES];
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
        const int r = wave * A_RPW + i * 16 + srow;
        a_chunk[i] = (slot ^ swz(r)) * 8;
        const int m = min(m0 + r, p.M - 1);
        const int ox = m % p.Wo, t = m / p.Wo, oy = t % p.Ho, b = t / p.Ho;
        a_iy0[i] = oy * p.stride - p.pad;
        a_ix0[i] = ox * p.stride - p.pad;
        a_img[i] = (size_t)b * p.H * p.W;
    }
    // weights: BN=128: wave owns rows [32w,32w+32) (2 pieces); BN=64: [16w,16w+16) (1 piece); BN=32: waves 0,1 own 16 rows each
    const bool w_active = (BN >= 64) || (wave < 2);
    const uint16_t* gw[W_PIECES];
#pragma unroll
    for (int i = 0; i < W_PIECES; ++i) {
        const int r = (BN >= 64 ? wave * W_ROWS_PER_WAVE : wave * 16) + i * 16 + srow;
        gw[i] = w_g + (size_t)(n0 + min(r, BN - 1)) * ktot + (slot ^ swz(r)) * 8;
    }
    const int w_lds_row0 = (BN >= 64 ? wave * W_ROWS_PER_WAVE : wave * 16);

    auto stage = [&](int buf, int kt) {
        // channel-chunk outer, filter tap inner: the KH*KW taps of one 32-channel chunk are consecutive K-steps,
        // so the +-1-pixel-shifted re-reads of the same input lines hit L1/L2 (tap-major order re-fetched the
        // whole input 9x from beyond L2: 6.5 GB for the 296^2 layer, PMC FETCH_SIZE, profiles/r01)
        const int chunk = kt / ntaps, tap = kt - chunk * ntaps, c0 = chunk * BK;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const size_t koff = (size_t)tap * p.Cin + c0;  // weight layout stays [Cout][KH][KW][Cin]
        char* sa = smem + buf * STAGE_BYTES + wave * A_RPW * 64;  // A hi plane, this wave's rows
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) {
            int iy = a_iy0[i] + kh, ix = a_ix0[i] + kw;
            if (p.replicate) iy = min(max(iy, 0), p.H - 1), ix = min(max(ix, 0), p.W - 1);
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const uint16_t* src = ok ? in_g + (a_img[i] + (size_t)iy * p.W + ix) * p.Cin + c0 + a_chunk[i] : p.zero + a_chunk[i];
            const uint16_t* src_lo = ok ? src + p.in_plane : src;
            __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(sa + i * 1024), 16, 0, 0);
            if (PASSES == 3) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src_lo), LDS_PTR(sa + A_PLANE + i * 1024), 16, 0, 0);
        }
        if (w_active) {
            char* sb = smem + buf * STAGE_BYTES + 2 * A_PLANE + w_lds_row0 * 64;
#pragma unroll
            for (int i = 0; i < W_PIECES; ++i) {
                const uint16_t* src = gw[i] + koff;
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(sb + i * 1024), 16, 0, 0);
                if (PASSES == 3) __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src + p.w_plane), LDS_PTR(sb + W_PLANE + i * 1024), 16, 0, 0);
            }
        }
    };

    // ---- fragment offsets (16x16x32: lane (fr, fq) reads row fr, chunk fq) ----
    const int wm = wave / WN, wn = wave % WN;
    const int fr = lane & 15, fq = lane >> 4;
    int a_off[TM], w_off[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * (TM * 16) + i * 16 + fr;
        a_off[i] = r * 64 + ((fq ^ swz(r)) << 4);
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int r = wn * (TN * 16) + i * 16 + fr;
        w_off[i] = 2 * A_PLANE + r * 64 + ((fq ^ swz(r)) << 4);
    }

    f32x4 acc[TN][TM];
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int m = 0; m < TM; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // K-tiles of this workgroup: all of them, or slice `slice` of `splitk` consecutive ranges (balanced, in K order)
    const int kt0 = p.splitk > 1 ? (int)((long long)nk * slice / p.splitk) : 0;
    const int kt1 = p.splitk > 1 ? (int)((long long)nk * (slice + 1) / p.splitk) : nk;
    const int nkl = kt1 - kt0;
    if (NS == 2) {
        stage(0, kt0);
    } else {
#pragma unroll
        for (int i = 0; i < NS - 1; ++i)
            if (i < nkl) stage(i, kt0 + i);
    }
    for (int it = 0; it < nkl; ++it) {
        if (NS == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (it + 1 < nkl) stage((it + 1) & 1, kt0 + it + 1);
        } else {
            // K-tile `it` has landed once at most the DMA pieces of the younger tiles (<= NS - 2 of them) are outstanding.
            // RAW: every wave waits for its own pieces, then the barrier.  WAR: the slot restaged below was last read in
            // iteration it - 1, whose fragment reads every wave has retired (lgkmcnt(0) before its MFMAs) before this barrier.
            const int younger = min(NS - 2, nkl - 1 - it);
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (it + NS - 1 < nkl) stage((it + NS - 1) % NS, kt0 + it + NS - 1);
        }
        const char* s = smem + (it % NS) * STAGE_BYTES;
        bf16x8 ah[TM], al[TM], wh[TN], wl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            ah[i] = *(const bf16x8*)(s + a_off[i]);
            if (PASSES == 3) al[i] = *(const bf16x8*)(s + A_PLANE + a_off[i]);
            if (p.relu_in) {
                const bf16x8 neg = ah[i] >> 15;  // 0xFFFF where hi < 0 (sign of hi decides for both halves)
                ah[i] &= ~neg;
                if (PASSES == 3) al[i] &= ~neg;
            }
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            wh[i] = *(const bf16x8*)(s + w_off[i]);
            if (PASSES == 3) wl[i] = *(const bf16x8*)(s + W_PLANE + w_off[i]);
        }
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                if (PASSES == 3) {
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[n], ah[m], acc[n][m], 0, 0, 0);
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[n], al[m], acc[n][m], 0, 0, 0);
                }
                acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[n], ah[m], acc[n][m], 0, 0, 0);
            }
    }

    // ---- epilogue, staged through LDS (conv_x3_common.h): each wave parks its tile in its own slice of the idle staging buffer ----
    __syncthreads();  // all waves are done with the operand stages
    if (p.splitk > 1) {
        // ---- split-K combine (cdna_hip_programming.md section 5, "In-launch split-K reduction"): plain 16-byte slab stores in
        // fragment order (1 KiB per wave instruction) -> every wave drains -> barrier -> one lane: agent release, drain, ticket ----
        constexpr int FR = TN * TM;                       // f32x4 fragments per lane
        constexpr int TILE_F = 4 * FR * 256;              // floats of one partial tile (4 waves x FR x 64 lanes x 4)
        float* const tile_slab = p.slab + (size_t)bid * p.splitk * TILE_F;
        float* const mine = tile_slab + (size_t)slice * TILE_F + (wave * FR) * 256 + lane * 4;
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int m = 0; m < TM; ++m) *(f32x4*)(mine + (n * TM + m) * 256) = acc[n][m];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* const flag = (unsigned*)smem;           // the staging buffers are idle: word 0 carries the ticket to all waves
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (always, and AFTER the fence: guideline 16 pitfall 12)
            *flag = __hip_atomic_fetch_add(p.counters + bid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const unsigned ticket = *flag;
        if (ticket != (unsigned)(p.splitk - 1)) return;   // not the last arriver of this tile
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(p.counters + bid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
        }
        __syncthreads();
        // partial tiles added in SLICE order (this workgroup's own one re-read like the others): the same bits whoever came last
        const float* src = tile_slab + (wave * FR) * 256 + lane * 4;
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int m = 0; m < TM; ++m) acc[n][m] = *(const f32x4*)(src + (n * TM + m) * 256);
        for (int sl = 1; sl < p.splitk; ++sl) {
            src += TILE_F;
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int m = 0; m < TM; ++m) acc[n][m] += *(const f32x4*)(src + (n * TM + m) * 256);
        }
        __syncthreads();  // `flag` (smem word 0) has been read by every wave before the epilogue re-uses the buffer
    }
    conv_x3_epilogue<TM, TN>(p, acc, smem + wave * (TM * 16 * TN * 64), m0 + wm * (TM * 16), n0 + wn * (TN * 16), lane, grp);
}

}  // namespace

// Diagnostics (bench.py's clock leg, tools/lab): while a buffer is set, launches of the 8-phase convolution kernel run its stamped
// instantiation and write one 8 x uint64 row per workgroup < rows (common.h GemmStamps); nullptr = off (the default).
static unsigned long long* g_conv_stamps = nullptr;
static int g_conv_stamp_rows = 0;
extern "C" int ufm_debug_set_conv_stamps(unsigned long long* buf, int rows) {
    UFM_REQUIRE((buf == nullptr) == (rows == 0) && rows >= 0, "ufm_debug_set_conv_stamps: buffer and row count must be given together");
    g_conv_stamps = buf;
    g_conv_stamp_rows = rows;
    return UFM_OK;
}
static int g_conv_variant_all = 0;
// test/tuning hook: 0 = auto, 1 = 128-row kernels only, 2 = 8-phase kernel wherever it is applicable,
// 3 = 128-row kernels only and never the deep (NS = 4) ring; + 16 = the serial (per-pass) residual read-out of rounds 1-4;
// bits 8..11 = pinned 8-phase tile height; bits 12..18 (with stamps set) = timing ablations of the 8-phase loop (ConvX3Args::ablate); bits 19 / 20 = the latency / the CU-time objective of the tile-height choice on every stream (default: by ufm_hint_concurrent_stream)
// Interleaved copies of convolution weights (round 6): a caller that holds, beside the planar split weights [2][G Cout][KH KW Cin] it passes to
// ufm_conv2d_nhwc_bf16x3(_grouped), the same values interleaved per 32-channel chunk -- [G Cout][KH KW Cin / 32][hi 32 | lo 32] -- registers the
// pair here; launches whose `weight` pointer is registered and that run on the halo kernel then stage W from the interleaved copy (whole 128-byte
// DMA rows).  The caller owns both buffers and removes the entry (il = NULL) before freeing either.  Bitwise the same results.
struct WilEntry {
    const void* planar;
    const uint16_t* il;
};
static WilEntry g_wil[512] = {};
static std::mutex g_wil_mutex;
extern "C" int ufm_conv_x3_register_interleaved_weights(const void* planar, const void* il) {
    UFM_REQUIRE(planar && ((uintptr_t)il % 16) == 0, "ufm_conv_x3_register_interleaved_weights: null / misaligned pointer");
    std::lock_guard<std::mutex> lock(g_wil_mutex);
    for (auto& e : g_wil)
        if (e.planar == planar) e = WilEntry{};
    if (!il) return UFM_OK;
    for (auto& e : g_wil)
        if (!e.planar) {
            e = WilEntry{planar, (const uint16_t*)il};
            return UFM_OK;
        }
    ufm_set_error("ufm_conv_x3_register_interleaved_weights: table full (512 entries)");
    return UFM_ERR_ARG;
}
static const uint16_t* conv_x3_wil_of(const void* planar) {
    for (const auto& e : g_wil)  // (read without the lock: entries are written whole, by callers that are not launching on these weights)
        if (e.planar == planar) return e.il;
    return nullptr;
}
extern "C" int ufm_debug_set_conv_variant(int v) {
    // fields: lab_flags.h conv_lab::ALL (one table; disjoint at compile time); a bit outside the table is refused
    const unsigned unknown = (unsigned)v & ~lab_known(conv_lab::ALL);
    UFM_REQUIRE(unknown == 0, "ufm_debug_set_conv_variant: bits 0x%x belong to no field of the lab flag table (lab_flags.h)", unknown);
    UFM_REQUIRE(lab_get(v, conv_lab::KERNEL) <= 4, "ufm_debug_set_conv_variant: kernel %d is not in 0..4", lab_get(v, conv_lab::KERNEL));
    const int nfp = lab_get(v, conv_lab::NF_PIN);
    UFM_REQUIRE(nfp == 0 || (nfp >= 5 && nfp <= 8), "ufm_debug_set_conv_variant: pinned tile height nf = %d is not 0 or 5..8", nfp);
    g_conv_variant_all = v;
    return UFM_OK;
}

// Split-K factor of a layer: a function of its GEOMETRY ONLY (K-tiles and output pixels per image), never of the batch or the
// group count, so that every pixel is summed in the same order whatever batch it is computed in.  Small maps with long K loops
// (the 19^2 / 37^2 layers of the DPT heads: 72-216 K-tiles on grids of 6-172 tiles) are latency-bound chains of K-steps; cutting
// the chain 2-6 ways multiplies the workgroups that share a CU.
constexpr long long SPLITK_COUNTER_BYTES = 64 * 1024;  // 16384 tile counters in front of the slabs
static int conv_x3_splitk_factor(int Ho, int Wo, int Cin, int KH, int KW, int passes) {
    const int nk = KH * KW * (Cin / BK);
    if (passes != 3 || (long long)Ho * Wo > 1600 || nk < 48) return 1;
    const int s = nk / 24;
    return s < 1 ? 1 : s > 6 ? 6 : s;
}
// rows = output pixels of ALL groups; bytes of counters + partial tiles for any tile shape the launcher may pick
static long long conv_x3_splitk_bytes(long long rows, int Cout, int S) {
    if (S <= 1) return 0;
    const long long rpad = (rows + 127) / 128 * 128 + 128LL * 64, cpad = (Cout + 127) / 128 * 128;  // (+ one ragged tile per group, <= 64 groups)
    return SPLITK_COUNTER_BYTES + rpad * cpad * 4 * S;
}
extern "C" long long ufm_conv_x3_splitk_ws_bytes(int groups, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    if (groups < 1 || B < 1 || KH < 1 || KW < 1 || stride < 1 || Cin % BK) return 0;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return 0;
    return conv_x3_splitk_bytes((long long)groups * B * Ho * Wo, Cout, conv_x3_splitk_factor(Ho, Wo, Cin, KH, KW, 3));
}

// Kernel choice for one problem (shared by the convolution and the Linear entry points).
static void launch_conv_x3(const ConvX3Args& p_in, int passes, hipStream_t stream) {
    ConvX3Args p = p_in;
    p.serial_epilogue = lab_get(g_conv_variant_all, conv_lab::SERIAL_EPILOGUE);
    p.stamps = g_conv_stamps, p.stamp_rows = g_conv_stamp_rows;
    p.ablate = g_conv_stamps ? lab_get(g_conv_variant_all, conv_lab::ABLATE) : 0;  // (only the stamped instantiation reads it)
    const int nf_pin = lab_get(g_conv_variant_all, conv_lab::NF_PIN);  // tools / tests: pin the 8-phase tile height (5..8 fragments per wave row)
    const int g_conv_variant = lab_get(g_conv_variant_all, conv_lab::KERNEL);
    const int Cout = p.Cout, KH = p.KH, KW = p.KW, Cin = p.Cin;
    const long long M = p.M;
    const int S = p.splitk;
    if (p.il) {  // interleaved operands: the 8-phase Linear form only; tile height by the cost model below (no hybrid: the 128-row kernels read planes)
        const long long ncu = ufm_device_cu_count();
        const bool throughput = lab_get(g_conv_variant_all, conv_lab::CU_TIME) || (ufm_stream_is_concurrent(stream) && !lab_get(g_conv_variant_all, conv_lab::LATENCY));
        const bool only8 = M >= 8192 && throughput;
        double best = 1e30;
        int best_nf = 8;
        for (int nf = 8; nf >= (only8 ? 8 : 5); --nf) {
            const long long t = ((M + 32 * nf - 1) / (32 * nf)) * (Cout / 256);
            const double c = (double)((t + ncu - 1) / ncu) * (0.65 + 0.35 * nf / 8.0);
            if (c < best - 1e-9) best = c, best_nf = nf;
        }
        if (nf_pin >= 5 && nf_pin <= 8) best_nf = nf_pin;
        ufm_launch_gemm_x3_il_8ph(p, stream, best_nf);
        return;
    }
    // Kernel choice.  Cout % 256 == 0 and a grid that fills the chip: the 256x256 8-phase kernel on the leading pixels
    // that make whole rounds of 256 CUs, the 128-row kernel on the rest (g_conv_variant: 0 auto, 1 = 128-row kernels
    // only, 2 = 8-phase on everything it accepts -- tests/tools).
    const int G = p.groups;
    auto launch128 = [&](const ConvX3Args& q) {
        const long long Mq = q.M - q.m_begin;
        const int ntm = (int)((Mq + 127) / 128) * G;  // row tiles of all groups (128-row kernels; the 64-row grids below scale the same way)
        const long long blocks128 = (long long)ntm * (Cout % 128 == 0 ? Cout / 128 : Cout % 64 == 0 ? Cout / 64 : Cout / 32) * S;
        // small grid: 64x64 tiles, several co-resident blocks per CU.  Threshold from an end-to-end sweep (400 / 200 / 100:
        // 189.6 / 196.6 / 197.2 pairs/s with two micro-batches): at 128..400 blocks the 128-row tiles win
        if (passes == 1) {  // plain bf16: the two-stage kernels only
            if (Cout % 64 == 0 && blocks128 < 128)
                hipLaunchKernelGGL((conv_x3_kernel<64, 64, 2, 1>), dim3((unsigned)(((Mq + 63) / 64) * G * (Cout / 64))), dim3(256), 0, stream, q);
            else if (Cout % 128 == 0)
                hipLaunchKernelGGL((conv_x3_kernel<128, 128, 2, 1>), dim3(ntm * (Cout / 128)), dim3(256), 0, stream, q);
            else if (Cout % 64 == 0)
                hipLaunchKernelGGL((conv_x3_kernel<128, 64, 2, 1>), dim3(ntm * (Cout / 64)), dim3(256), 0, stream, q);
            else
                hipLaunchKernelGGL((conv_x3_kernel<128, 32, 2, 1>), dim3(ntm * (Cout / 32)), dim3(256), 0, stream, q);
            return;
        }
        if (Cout % 64 == 0 && blocks128 < 128) {
            const unsigned grid = (unsigned)(((Mq + 63) / 64) * G * (Cout / 64)) * S;
            // measured per layer (tools/conv_breakdown.py): the deep ring pays on long K loops only (19^2 768->256: 138 -> 102 us;
            // short loops lose 2-3 us to its prologue), and not at all on the 128x128 tile (37^2 RCU: 68 -> 83 us)
            if (grid <= 512 && KH * KW * (Cin / 32) / S >= 64 && g_conv_variant != 3)
                hipLaunchKernelGGL((conv_x3_kernel<64, 64, 4>), dim3(grid), dim3(256), 0, stream, q);
            else
                hipLaunchKernelGGL((conv_x3_kernel<64, 64, 2>), dim3(grid), dim3(256), 0, stream, q);
        } else if (Cout % 128 == 0) {
            hipLaunchKernelGGL((conv_x3_kernel<128, 128, 2>), dim3(ntm * (Cout / 128) * S), dim3(256), 0, stream, q);
        } else if (Cout % 64 == 0) {
            hipLaunchKernelGGL((conv_x3_kernel<128, 64, 2>), dim3(ntm * (Cout / 64) * S), dim3(256), 0, stream, q);
        } else {
            hipLaunchKernelGGL((conv_x3_kernel<128, 32, 2>), dim3(ntm * (Cout / 32) * S), dim3(256), 0, stream, q);
        }
    };
    // 8-phase tile: 256 px x 256 cout (Cout % 256 == 0)
    const int tile_n = 256, tile_m = 256;
    const bool ok8 = Cout % 256 == 0 && KH * KW * (Cin / 32) >= 2 && p.in_plane < (1ll << 31) && p.w_plane < (1ll << 31);
    const long long t8 = ((M + tile_m - 1) / tile_m) * (Cout / tile_n) * G;  // 8-phase tiles of all groups
    // Round 6: wherever the 8-phase kernel is chosen, a 3x3 / stride 1 / pad 1 / zero-padding layer runs on its row-window halo form
    // (conv_bf16x3_halo.hip: the input staged once per filter row, not once per tap; bit-identical).  Lab field HALO = 1: never (A/B, tests).
    const bool halo = KH == 3 && KW == 3 && p.stride == 1 && p.pad == 1 && !p.replicate && !p.relu_in && p.Ho == p.H && p.Wo == p.W && passes == 3 &&
                      p.W >= 32 && (long long)(p.H - 2) * p.W >= 272 &&   // one column border per 16-pixel fragment; image boundaries further apart than a window + two rows
                      p.in_plane < (1ll << 29) && p.w_plane < (1ll << 29) &&  // 32-bit buffer offsets of both planes below 2^31 bytes
                      lab_get(g_conv_variant_all, conv_lab::HALO) != 1 && !p.ablate;
    auto launch8 = [&](const ConvX3Args& q_, int nf) {
        if (halo) {
            ConvX3Args q = q_;
            q.w_il = lab_get(g_conv_variant_all, conv_lab::HALO) == 2 ? nullptr : conv_x3_wil_of(q.w);  // HALO = 2: the planar W staging (A/B)
            ufm_launch_conv_x3_halo(q, stream, nf);
        } else {
            ufm_launch_conv_x3_8ph(q_, stream, nf);
        }
    };
    // 256 px x 128 cout pair tile (round 5): Cout a multiple of 128 but not of 256 (the heads' p_conv1, 296^2 x 256 -> 128) on grids of at
    // least one workgroup per CU.  g_conv_variant 4 = wherever it applies (tests), 1 / 3 = never.
    const bool okp = Cout % 128 == 0 && KH * KW * (Cin / 32) >= 2 && p.in_plane < (1ll << 31) && p.w_plane < (1ll << 31) && passes == 3 && S == 1;
    const long long tpair = ((M + 255) / 256) * (Cout / 128) * G;
    if (okp && (g_conv_variant == 4 || (g_conv_variant == 0 && Cout % 256 != 0 && tpair >= ufm_device_cu_count() && KH * KW * (Cin / 32) >= 16))) {
        ufm_launch_conv_x3_pair(p, stream);
    } else if (passes == 1 || S > 1) {  // (split-K lives in the 128- / 64-row kernels: its layers are the small maps the 8-phase tile never fits)
        launch128(p);
    } else if (ok8 && g_conv_variant == 2) {
        launch8(p, nf_pin >= 5 && nf_pin <= 8 ? nf_pin : 8);
    } else if (ok8 && g_conv_variant == 0 && t8 >= ufm_device_cu_count() / 2 && KH * KW * (Cin / 32) >= 16) {  // (variants 1 and 3 never take this branch)
        // Measured per shape (tools/lab/conv_rounds.py, profiles/r03/conv_rounds.log): a last partial round of at least half
        // the chip is cheaper on the 8-phase kernel than on the 128-row kernels (148^2 RCU, 684 tiles: 485 vs 518 us; 74^2 RCU,
        // 171 tiles: 154 vs 178 us), a smaller one on the 128-row kernels (148^2 at 4 images, 342 tiles: 270 vs 301 us); with
        // fewer than 16 K-tiles (1x1 layers) the 8-phase prologue and drain cost more than its loop gains (110 vs 122-130 us).
        const long long ncu = ufm_device_cu_count();
        const long long full = t8 / ncu;                                 // whole rounds of the 8-phase kernel
        const long long m_main = full * ncu / ((long long)(Cout / tile_n) * G) * tile_m;  // leading pixels (of every group) whose tiles fit in them
        // Round 5: tiles of 160 / 192 / 224 rows as well (32 nf rows; conv_bf16x3_8ph.hip).  Cost in units of one round of 256-row tiles,
        // fitted to tools/lab/gemm_x3_rows.py (profiles/r05/gemm_x3_rows.log): a tile of 32 nf rows costs FIX + (1 - FIX) nf / 8 (prologue,
        // drain and the barriers of a phase do not shrink with its MFMA count), a launch is whole rounds of the chip; the hybrid form
        // (256-row tiles on the whole rounds + the 128-row kernels on a last partial round below half the chip) costs its whole rounds
        // + 0.6 .. 0.95 for the second launch.  M = 10 960 x N = 1024: 172 tiles of 256 rows (two thirds of a round) -> 232 of 192 rows,
        // 71.8 -> 67.5 us; x 768: 129 tiles -> 207 of 160 rows, 54.4 -> 48.0 us; the N >= 3072 shapes keep the hybrid form.
        // That cost model prices one launch ALONE on the chip.  On a stream flagged by ufm_hint_concurrent_stream (the engine's micro-batch streams)
        // what counts is the CU time a launch takes from the other stream's kernels: lower tiles and the 128-row rest launch finish a lone launch
        // sooner and cost more CU time.  There, from 8192 rows on: full-height tiles, no hybrid split -- numerics "precise" +2.2...+2.8 % pairs/s,
        // "fast" +-0 (profiles/r05/gemm_tile_policy_pipeline.log).  Variant bits 19 / 20 (lab): the latency / the CU-time objective on every stream.
        const bool throughput = lab_get(g_conv_variant_all, conv_lab::CU_TIME) || (ufm_stream_is_concurrent(stream) && !lab_get(g_conv_variant_all, conv_lab::LATENCY));
        const bool only8 = M >= 8192 && throughput;
        constexpr double FIX = 0.65;
        double best = 1e30;
        int best_nf = 8;
        for (int nf = 8; nf >= (only8 ? 8 : 5); --nf) {
            const long long t = ((M + 32 * nf - 1) / (32 * nf)) * (Cout / tile_n) * G;
            const double c = (double)((t + ncu - 1) / ncu) * (FIX + (1.0 - FIX) * nf / 8.0);
            if (c < best - 1e-9) best = c, best_nf = nf;
        }
        bool hybrid = false;
        if (nf_pin >= 5 && nf_pin <= 8) {
            best_nf = nf_pin;
        } else if (!only8 && full >= 1 && m_main < M && t8 - full * ncu < ncu / 2) {
            const long long blocks128 = ((M - m_main + 127) / 128) * (Cout / 128) * G;
            const double frac128 = (double)blocks128 / (2.0 * (double)ncu);
            hybrid = (double)full + 0.6 + 0.35 * (frac128 < 1.0 ? frac128 : 1.0) < best;
        }
        if (!hybrid) {
            launch8(p, best_nf);
        } else {
            ConvX3Args lead = p, rest = p;
            lead.M = (int)m_main;
            rest.m_begin = (int)m_main;
            launch8(lead, 8);
            launch128(rest);
        }
    } else {
        launch128(p);
    }
}

extern "C" int ufm_conv2d_nhwc_bf16x3(const uint16_t* in, int B, int H, int W, int Cin, const uint16_t* weight, int Cout,
                                      int KH, int KW, int stride, int pad, int relu_in, const float* bias, int act,
                                      const uint16_t* res1, const uint16_t* res2, int shuffle, uint16_t* out,
                                      uint16_t* out_relu, const uint16_t* zero_page, int passes, void* stream) {
    return ufm_conv2d_nhwc_bf16x3_grouped(in, 1, 0, B, H, W, Cin, weight, Cout, KH, KW, stride, pad, relu_in, bias, act, res1, res2, shuffle, out, out_relu,
                                          zero_page, passes, nullptr, 0, stream);
}

// `groups` convolutions of identical geometry in one launch (conv_x3_common.h ConvX3Args): the two DPT heads.
extern "C" int ufm_conv2d_nhwc_bf16x3_grouped(const uint16_t* in, int groups, int in_shared, int B, int H, int W, int Cin, const uint16_t* weight,
                                              int Cout, int KH, int KW, int stride, int pad, int relu_in, const float* bias, int act,
                                              const uint16_t* res1, const uint16_t* res2, int shuffle, uint16_t* out,
                                              uint16_t* out_relu, const uint16_t* zero_page, int passes, void* splitk_ws, long long splitk_ws_bytes,
                                              void* stream) {
    UFM_REQUIRE(in && weight && out && zero_page, "ufm_conv2d_nhwc_bf16x3: null pointer");
    UFM_REQUIRE(groups >= 1 && groups <= 64, "ufm_conv2d_nhwc_bf16x3: groups=%d out of range", groups);
    UFM_REQUIRE(passes == 3 || passes == 1, "ufm_conv2d_nhwc_bf16x3: passes=%d must be 3 (bf16x3) or 1 (plain bf16)", passes);
    UFM_REQUIRE(B > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0, "ufm_conv2d_nhwc_bf16x3: bad geometry");
    UFM_REQUIRE(Cin % BK == 0 && Cin > 0, "ufm_conv2d_nhwc_bf16x3: Cin=%d must be a multiple of %d", Cin, BK);
    UFM_REQUIRE(Cout % 32 == 0 && Cout > 0, "ufm_conv2d_nhwc_bf16x3: Cout=%d must be a multiple of 32", Cout);
    UFM_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)weight % 16) == 0 && ((uintptr_t)out % 8) == 0, "ufm_conv2d_nhwc_bf16x3: misaligned pointer");
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    UFM_REQUIRE(Ho > 0 && Wo > 0, "ufm_conv2d_nhwc_bf16x3: empty output");
    int Co = Cout;
    if (shuffle) {
        UFM_REQUIRE(KH == 1 && KW == 1 && stride == 1 && pad == 0, "ufm_conv2d_nhwc_bf16x3: shuffle mode needs a 1x1 geometry");
        UFM_REQUIRE(Cout % (shuffle * shuffle) == 0, "ufm_conv2d_nhwc_bf16x3: Cout not divisible by shuffle^2");
        Co = Cout / (shuffle * shuffle);
        UFM_REQUIRE(Co % 4 == 0, "ufm_conv2d_nhwc_bf16x3: Co=%d must be a multiple of 4 in shuffle mode", Co);
        UFM_REQUIRE(!res1 && !res2 && !out_relu && act == UFM_ACT_NONE, "ufm_conv2d_nhwc_bf16x3: shuffle mode supports bias only");
    }
    const long long M = (long long)B * Ho * Wo;  // output pixels PER GROUP
    UFM_REQUIRE(M * groups < (1ll << 31), "ufm_conv2d_nhwc_bf16x3: problem too large");
    const long long Mout = (shuffle ? M * shuffle * shuffle : M) * groups;
    const long long in_imgs = in_shared ? B : (long long)B * groups;
    ConvX3Args p{in, weight, bias, res1, res2, zero_page, out, out_relu,
                 in_imgs * H * W * Cin, (long long)groups * Cout * KH * KW * Cin, Mout * Co,
                 B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, (int)M, relu_in & 1, act, shuffle, Co, 0};
    p.replicate = (relu_in >> 1) & 1;
    p.groups = groups, p.Mg = (int)M;
    p.in_group = in_shared ? 0 : (long long)B * H * W * Cin;
    p.w_group = (long long)Cout * KH * KW * Cin;
    p.splitk = 1;
    if (splitk_ws && !shuffle) {  // a workspace is offered: split the K loop of the layers conv_x3_splitk_factor names
        const int S = conv_x3_splitk_factor(Ho, Wo, Cin, KH, KW, passes);
        if (S > 1) {
            const long long need = conv_x3_splitk_bytes(M * groups, Cout, S);
            UFM_REQUIRE(((uintptr_t)splitk_ws % 16) == 0 && splitk_ws_bytes >= need, "ufm_conv2d_nhwc_bf16x3: split-K workspace of %lld bytes, this layer needs %lld (ufm_conv_x3_splitk_ws_bytes)", splitk_ws_bytes, need);
            // one counter per tile of ANY shape the launcher may pick: the smallest is 64 rows x 32 columns, rows are tiled per group
            UFM_REQUIRE(((M + 63) / 64) * (long long)groups * ((Cout + 31) / 32) <= SPLITK_COUNTER_BYTES / 4, "ufm_conv2d_nhwc_bf16x3: too many tiles for the split-K counters");
            p.splitk = S;
            p.counters = (unsigned*)splitk_ws;
            p.slab = (float*)((char*)splitk_ws + SPLITK_COUNTER_BYTES);
        }
    }
    launch_conv_x3(p, passes, (hipStream_t)stream);
    UFM_CHECK_LAUNCH("ufm_conv2d_nhwc_bf16x3");
    return UFM_OK;
}

// Linear layer on the split format: out[M][N] = epilogue(A[M][K] . W[N][K]^T), every product hi*hi + hi*lo + lo*hi.
// The 1x1 convolution over M "pixels" it is, with the transformer's Linear epilogue (bias, activation, per-column
// scale, fp32 residual in place) -- numerics mode "precise".
extern "C" int ufm_gemm_bf16x3(const uint16_t* A, const uint16_t* W, int M, int N, int K, const float* bias, int act,
                               const float* gamma, const float* res, void* out, int out_dtype, const uint16_t* zero_page,
                               void* stream) {
    UFM_REQUIRE(A && W && out && zero_page, "ufm_gemm_bf16x3: null pointer");
    UFM_REQUIRE(M > 0 && N > 0 && K > 0, "ufm_gemm_bf16x3: bad shape M=%d N=%d K=%d", M, N, K);
    UFM_REQUIRE(K % BK == 0, "ufm_gemm_bf16x3: K=%d must be a multiple of %d", K, BK);
    UFM_REQUIRE(N % 32 == 0, "ufm_gemm_bf16x3: N=%d must be a multiple of 32", N);
    UFM_REQUIRE(out_dtype == UFM_F32 || out_dtype == UFM_BF16X2, "ufm_gemm_bf16x3: out_dtype must be UFM_F32 or UFM_BF16X2");
    UFM_REQUIRE(out_dtype == UFM_F32 || !res, "ufm_gemm_bf16x3: the fp32 residual needs an fp32 output");
    UFM_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0, "ufm_gemm_bf16x3: misaligned pointer");
    UFM_REQUIRE((long long)M * K < (1ll << 31) && (long long)N * K < (1ll << 31) && (long long)M * N < (1ll << 31), "ufm_gemm_bf16x3: problem too large");
    ConvX3Args p{A, W, bias, nullptr, nullptr, zero_page, out_dtype == UFM_BF16X2 ? (uint16_t*)out : nullptr, nullptr,
                 (long long)M * K, (long long)N * K, (long long)M * N,
                 1, 1, M, K, N, 1, 1, 1, 0, 1, M, M, 0, act, 0, N, 0};
    p.gamma = gamma;
    p.res_f32 = res;
    p.out_f32 = out_dtype == UFM_F32 ? (float*)out : nullptr;
    p.groups = 1, p.Mg = M, p.splitk = 1;
    launch_conv_x3(p, 3, (hipStream_t)stream);
    UFM_CHECK_LAUNCH("ufm_gemm_bf16x3");
    return UFM_OK;
}

// The same Linear layer on INTERLEAVED split operands (round 6): A [M][K / 32][hi 32 | lo 32], W [N][K / 32][hi 32 | lo 32] -- every LDS-DMA row of
// the 8-phase loop is then one whole 128-byte line instead of two 64-byte halves a plane apart.  Written by ufm_layernorm (out_dtype
// UFM_BF16X2_IL); weights interleaved at pack time.  Outputs as ufm_gemm_bf16x3 (planar split or fp32).  N % 256 == 0, K % 32 == 0, K >= 64.
extern "C" int ufm_gemm_bf16x3_il(const uint16_t* A, const uint16_t* W, int M, int N, int K, const float* bias, int act,
                                  const float* gamma, const float* res, void* out, int out_dtype, const uint16_t* zero_page,
                                  void* stream) {
    UFM_REQUIRE(A && W && out && zero_page, "ufm_gemm_bf16x3_il: null pointer");
    UFM_REQUIRE(M > 0 && N > 0 && K > 0, "ufm_gemm_bf16x3_il: bad shape M=%d N=%d K=%d", M, N, K);
    UFM_REQUIRE(K % BK == 0 && K >= 2 * BK, "ufm_gemm_bf16x3_il: K=%d must be a multiple of %d and at least %d", K, BK, 2 * BK);
    UFM_REQUIRE(N % 256 == 0, "ufm_gemm_bf16x3_il: N=%d must be a multiple of 256 (the 8-phase tile)", N);
    UFM_REQUIRE(out_dtype == UFM_F32 || out_dtype == UFM_BF16X2 || out_dtype == UFM_BF16X2_IL, "ufm_gemm_bf16x3_il: out_dtype must be UFM_F32, UFM_BF16X2 or UFM_BF16X2_IL");
    UFM_REQUIRE(out_dtype == UFM_F32 || !res, "ufm_gemm_bf16x3_il: the fp32 residual needs an fp32 output");
    UFM_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0, "ufm_gemm_bf16x3_il: misaligned pointer");
    UFM_REQUIRE((long long)M * K < (1ll << 30) && (long long)N * K < (1ll << 30) && (long long)M * N < (1ll << 31), "ufm_gemm_bf16x3_il: problem too large");
    ConvX3Args p{A, W, bias, nullptr, nullptr, zero_page, out_dtype != UFM_F32 ? (uint16_t*)out : nullptr, nullptr,
                 (long long)M * K, (long long)N * K, (long long)M * N,
                 1, 1, M, K, N, 1, 1, 1, 0, 1, M, M, 0, act, 0, N, 0};
    p.gamma = gamma;
    p.res_f32 = res;
    p.out_f32 = out_dtype == UFM_F32 ? (float*)out : nullptr;
    p.groups = 1, p.Mg = M, p.splitk = 1;
    p.il = 1;
    p.out_il = out_dtype == UFM_BF16X2_IL;
    launch_conv_x3(p, 3, (hipStream_t)stream);
    UFM_CHECK_LAUNCH("ufm_gemm_bf16x3_il");
    return UFM_OK;
}
