// Shared pieces of the bf16 GEMM kernels (gemm_bf16.hip, gemm_bf16_8ph.hip): argument block, fused epilogues.
#pragma once
#include "common.h"
#include "lab_flags.h"

struct GemmArgs {
    const uint16_t* A;
    const uint16_t* W;
    const float* bias;
    const float* gamma;
    const float* res;
    void* out;
    int lda, ldw, M, N, K;
    int act, ldres, res_row_mod, ldo, out_row_group;
    int debug;    // the lab flag word (ufm_debug_set_gemm_flags); fields: lab_flags.h gemm_lab::*, read through lab_get() only
    int m_begin;  // the launch covers rows [m_begin, M) (hybrid 256x256 + 128x128 split of one GEMM); row indices stay absolute
    // fused RoPE-2D (ufm_gemm_bf16_rope, rope.hip): columns [0, rope_cols) of the bf16 output are rotated per 64-wide head with
    // the fp32 tables [rope_mod][64] indexed by row % rope_mod; null tables = no rotation
    const float* rope_cos;
    const float* rope_sin;
    int rope_mod, rope_cols;
    // diagnostic builds only (ufm_debug_set_gemm_stamps; the STAMP = true instantiations): 8 x uint64 per workgroup, see gemm_stamp_row
    unsigned long long* stamps;
    int stamp_rows;
    // round 6 (lab: ufm_debug_set_gemm_splitk): deterministic 2-way split of the K loop of a read-modify-write launch -- workgroups 2 v and
    // 2 v + 1 compute the two halves of output tile v, each parks its fp32 partial tile in `slab` (fragment order), the second to arrive (an
    // agent-scope arrival counter per tile) adds the two in HALF order and runs the epilogue.  The cut is at K / 2: it depends on the layer
    // alone, so a row's bits do not depend on its batch neighbours.  0 / 1 = off.
    int splitk;
    float* slab;          // [tiles][2][256 * 256] partial tiles
    unsigned* counters;   // [tiles], zero between launches (the last arriver resets its word)
};

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB


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
                v = gelu_bf16_x4(v);
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


// LDS-staged epilogue for the 128x128 kernel.  Ablation (tools/ksweep.py) showed the direct epilogue --
// a wave store touching 16 rows x 32 B -- costs ~50 us of a 190 us QKV GEMM.  Each wave parks its 64x64
// fp32 tile in its own 16 KiB of the (now idle) staging buffer, 16-byte chunk index XOR (row & 15)
// (conflict-free for both the accumulator-shaped writes and the row-shaped reads), then every wave
// instruction moves 4 whole rows: 256-B fp32 / 128-B bf16 contiguous per row, for the residual read
// (in-place update) and the store alike.  Within a wave LDS ops are in order: no barrier after the writes.
// ROWS (a multiple of 16, <= 64): only the first ROWS rows of the slice exist (8-phase tiles lower than 256 rows).
// EPI: the epilogue's run-time switches as compile-time facts for the transformer's four hot Linear forms (0 = generic, every
// switch read from the argument block).  With the switches known the unrolled fragment loop has no branches in it, so the
// compiler interleaves the independent GELU polynomial chains of neighbouring fragments instead of running 32 dependent
// chains one after the other behind scalar branches and per-column-group `s_waitcnt vmcnt(0)`s:
//   1 = bias, GELU, bf16 out (fc1)          2 = bias, gamma, bf16 out (QKV with the Q pre-scale)
//   3 = bias, gamma, fp32 residual in place (proj / fc2 with LayerScale)      4 = bias, fp32 residual in place (no LayerScale)
template <int EPI>
struct EpiTraits {
    static constexpr bool known = EPI != 0;
    static constexpr bool gelu = EPI == 1, gamma = EPI == 2 || EPI == 3, res = EPI == 3 || EPI == 4;
};

template <int OUT_BF16, int ROWS = 64, int EPI = 0>
__device__ __forceinline__ void epilogue_lds(const GemmArgs& p, f32x4 (&acc)[4][4], char* wave_lds, int row0, int col0,
                                             int lane) {
    static_assert(ROWS % 16 == 0 && ROWS > 0 && ROWS <= 64, "ROWS");
    using E = EpiTraits<EPI>;
    const bool has_bias = E::known ? true : p.bias != nullptr;
    const bool has_gamma = E::known ? E::gamma : p.gamma != nullptr;
    const bool has_res = E::known ? E::res : p.res != nullptr;
    const int act = E::known ? (E::gelu ? UFM_ACT_GELU : UFM_ACT_NONE) : p.act;
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int nb = col0 + n * 16 + fq * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f}, gv = {1.f, 1.f, 1.f, 1.f};
        if (has_bias) bv = *(const f32x4*)(p.bias + nb);
        if (has_gamma) gv = *(const f32x4*)(p.gamma + nb);
#pragma unroll
        for (int m = 0; m < ROWS / 16; ++m) {
            f32x4 v = acc[n][m] + bv;
            if (act == UFM_ACT_GELU && OUT_BF16) {
                v = gelu_bf16_x4(v);
            } else if (act != UFM_ACT_NONE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = apply_act(v[j], act);
            }
            if (has_gamma) v *= gv;
            const int r = m * 16 + fr;
            *(f32x4*)(wave_lds + r * 256 + (((n * 4 + fq) ^ (r & 15)) << 4)) = v;
        }
    }
    // Read-out.  PLAIN = the whole 64-row slice is inside M and no row re-mapping is in use (every tile but the ragged last
    // row-panel of every GEMM but patch-embed): a straight-line, branch-free body, so the compiler issues all LDS reads
    // and residual loads up front instead of one dependent {ds_read, wait, branch, convert, store} round trip per pass
    // (that serialisation was ~12 us of fixed cost per 5-round GEMM: epilogue time = 11.9 us + bytes / 7.05 TB/s).
    auto readout = [&](auto plain_c) {
        constexpr bool PLAIN = decltype(plain_c)::value;
        if (OUT_BF16 && !has_res && (E::known || ((p.ldo & 7) == 0 && ((uintptr_t)p.out & 15) == 0))) {
            // bf16 output without residual: a lane converts 8 consecutive columns (two staged chunks) and stores 16 B,
            // a wave instruction covers 8 whole 128-B row segments
            const int r8 = lane >> 3, c8 = lane & 7;
            const bool rope = !E::known && p.rope_cos != nullptr && col0 < p.rope_cols;  // wave-uniform: the wave's 64 columns are one head
#pragma unroll
            for (int pass = 0; pass < ROWS / 8; ++pass) {
                const int r = pass * 8 + r8;
                f32x4 v0 = *(const f32x4*)(wave_lds + r * 256 + (((2 * c8) ^ (r & 15)) << 4));
                f32x4 v1 = *(const f32x4*)(wave_lds + r * 256 + (((2 * c8 + 1) ^ (r & 15)) << 4));
                if (rope) {  // out[j] = v[j] cos[t][j] + v[j ^ 16] sin[t][j]: the partner columns are 4 chunks away in the same staged row
                    const f32x4 q0 = *(const f32x4*)(wave_lds + r * 256 + ((((2 * c8) ^ 4) ^ (r & 15)) << 4));
                    const f32x4 q1 = *(const f32x4*)(wave_lds + r * 256 + ((((2 * c8 + 1) ^ 4) ^ (r & 15)) << 4));
                    const float* ct = p.rope_cos + (size_t)(min(row0 + r, p.M - 1) % p.rope_mod) * 64 + c8 * 8;
                    const float* st = p.rope_sin + (size_t)(min(row0 + r, p.M - 1) % p.rope_mod) * 64 + c8 * 8;
                    v0 = v0 * *(const f32x4*)ct + q0 * *(const f32x4*)st;
                    v1 = v1 * *(const f32x4*)(ct + 4) + q1 * *(const f32x4*)(st + 4);
                }
                const int row = row0 + r;
                int orow = row;
                if (!PLAIN) {
                    if (row >= p.M) continue;
                    if (p.out_row_group > 0) orow = (row / p.out_row_group) * (p.out_row_group + 1) + 1 + row % p.out_row_group;
                }
                u32x4 pk = {pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[0], v1[1]), pack_bf16x2(v1[2], v1[3])};
                *(u32x4*)((uint16_t*)p.out + (size_t)orow * p.ldo + col0 + c8 * 8) = pk;
            }
            return;
        }
        const int rr = lane >> 4, c = lane & 15;
        const int nb = col0 + c * 4;
#pragma unroll
        for (int pass = 0; pass < ROWS / 4; ++pass) {
            const int r = pass * 4 + rr;
            f32x4 v = *(const f32x4*)(wave_lds + r * 256 + ((c ^ (r & 15)) << 4));
            const int row = row0 + r;
            int rrow = row, orow = row;
            if (!PLAIN) {
                if (row >= p.M) continue;
                if (p.res_row_mod > 0) rrow = row % p.res_row_mod;
                if (p.out_row_group > 0) orow = (row / p.out_row_group) * (p.out_row_group + 1) + 1 + row % p.out_row_group;
            }
            if (has_res) v += *(const f32x4*)(p.res + (size_t)rrow * p.ldres + nb);
            if (OUT_BF16) {
                u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                *(u32x2*)((uint16_t*)p.out + (size_t)orow * p.ldo + nb) = pk;
            } else {
                *(f32x4*)((float*)p.out + (size_t)orow * p.ldo + nb) = v;
            }
        }
    };
    if (row0 + ROWS <= p.M && (E::known || (p.res_row_mod == 0 && p.out_row_group == 0)))
        readout(std::integral_constant<bool, true>{});
    else
        readout(std::integral_constant<bool, false>{});
}


// The read-modify-write epilogue (EPI 3 / 4: bias [, LayerScale], fp32 residual updated in place) of a wave's TWO 64-column slices
// (rows [row0, row0 + 64) from acc0, [row0 + 64, row0 + 64 + ROWS1) from acc1) with the residual loads taken out of the store chain.
// epilogue_lds' read-out compiles to {global_load, s_waitcnt vmcnt(0), add, global_store} x 16 per slice: `res` and `out` are the
// same buffer, so hipcc may not move load i + 1 above store i, and vmcnt(0) in front of every add also waits for the previous
// STORE (stores count in vmcnt): 32+ serial memory round trips per wave -- the 28-41 k cycles per tile the stamps show
// (tools/lab/gemm_stamps.py; round 5).  Every lane reads and writes only its own elements, so here all 16 residual loads of a slice
// are issued first (the second slice's as soon as acc0's registers are free, i.e. under the first slice's read-out), and the
// stores follow; the compiler's counted vmcnt waits do the rest.  Only for whole slices (no ragged rows): the caller falls back
// to epilogue_lds otherwise.  Same arithmetic in the same order as epilogue_lds: bit-identical.
template <int ROWS1, int EPI>
__device__ __forceinline__ void epilogue_lds_rmw2(const GemmArgs& p, f32x4 (&acc0)[4][4], f32x4 (&acc1)[4][4], char* wave_lds, int row0, int col0, int lane) {
    static_assert(EPI == 3 || EPI == 4 || EPI == 5, "read-modify-write epilogues only");  // 5: LayerScale decided at run time (128x128 kernel)
    static_assert(ROWS1 % 16 == 0 && ROWS1 >= 0 && ROWS1 <= 64, "ROWS1");
    constexpr bool HAS_GAMMA = EPI == 3 || EPI == 5;
    const int fr = lane & 15, fq = lane >> 4;
    const int rr = lane >> 4, c = lane & 15;
    const int nb = col0 + c * 4;
    const float* rbase = p.res + (size_t)(row0 + rr) * p.ldres + nb;
    float* obase = (float*)p.out + (size_t)(row0 + rr) * p.ldo + nb;
    // A compiler-visible vmcnt(0): the K loop's LDS-DMA waits are inline asm, so hipcc still has the DMAs on its scoreboard and would
    // put vmcnt(0) -- not a counted wait -- in front of the first use of ANY load result below (they have all landed: free).
    __builtin_amdgcn_s_waitcnt(0x0F70);
    f32x4 rv0[16], rv1[ROWS1 / 4 > 0 ? ROWS1 / 4 : 1];
    f32x4 bv[4], gv[4];  // first: vmcnt counts in issue order, and the staging below needs these, not the residual rows
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        bv[n] = *(const f32x4*)(p.bias + col0 + n * 16 + fq * 4);
        if (EPI == 5) gv[n] = p.gamma ? *(const f32x4*)(p.gamma + col0 + n * 16 + fq * 4) : f32x4{1.f, 1.f, 1.f, 1.f};  // x * 1.0f is exact
        else if (HAS_GAMMA) gv[n] = *(const f32x4*)(p.gamma + col0 + n * 16 + fq * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) rv0[i] = *(const f32x4*)(rbase + (size_t)(4 * i) * p.ldres);
    __builtin_amdgcn_sched_barrier(0);
    auto stage = [&](f32x4 (&acc)[4][4], auto rows_c) {
        constexpr int ROWS = decltype(rows_c)::value;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int m = 0; m < ROWS / 16; ++m) {
                f32x4 v = acc[n][m] + bv[n];
                if (HAS_GAMMA) v *= gv[n];
                const int r = m * 16 + fr;
                *(f32x4*)(wave_lds + r * 256 + (((n * 4 + fq) ^ (r & 15)) << 4)) = v;
            }
    };
    stage(acc0, std::integral_constant<int, 64>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ROWS1 > 0) {  // the second slice's residual rows, into the registers acc0 just left
#pragma unroll
        for (int i = 0; i < ROWS1 / 4; ++i) rv1[i] = *(const f32x4*)(rbase + (size_t)(64 + 4 * i) * p.ldres);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = 4 * i + rr;
        f32x4 v = *(const f32x4*)(wave_lds + r * 256 + ((c ^ (r & 15)) << 4));
        v += rv0[i];
        *(f32x4*)(obase + (size_t)(4 * i) * p.ldo) = v;
    }
    if constexpr (ROWS1 > 0) {
        __builtin_amdgcn_sched_barrier(0);
        stage(acc1, std::integral_constant<int, ROWS1>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < ROWS1 / 4; ++i) {
            const int r = 4 * i + rr;
            f32x4 v = *(const f32x4*)(wave_lds + r * 256 + ((c ^ (r & 15)) << 4));
            v += rv1[i];
            *(f32x4*)(obase + (size_t)(64 + 4 * i) * p.ldo) = v;
        }
    }
}

// The two slices of a wave through whichever epilogue applies: the pipelined read-modify-write form for whole slices of the
// residual epilogues, epilogue_lds otherwise (bf16 outputs, ragged last row panel, run-time switched EPI 0).
template <int OUT_BF16, int ROWS1, int EPI>
__device__ __forceinline__ void epilogue_two_slices(const GemmArgs& p, f32x4 (&acc0)[4][4], f32x4 (&acc1)[4][4], char* wave_lds, int row0, int col0, int lane) {
    if constexpr (!OUT_BF16 && (EPI == 3 || EPI == 4)) {
        if (row0 + 64 + ROWS1 <= p.M && !lab_get(p.debug, gemm_lab::SERIAL_RMW)) {  // the old serial read-out (A/B)
            epilogue_lds_rmw2<ROWS1, EPI>(p, acc0, acc1, wave_lds, row0, col0, lane);
            return;
        }
    }
    epilogue_lds<OUT_BF16, 64, EPI>(p, acc0, wave_lds, row0, col0, lane);
    if constexpr (ROWS1 > 0) epilogue_lds<OUT_BF16, ROWS1, EPI>(p, acc1, wave_lds, row0 + 64, col0, lane);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace

// gemm_bf16_8ph.hip: 256x256 8-phase kernel (N % 256 == 0, K >= 128, 32-bit operand offsets)
// nf = 16-row fragments per wave (5..8): tile height 32 * nf rows
// epi: EpiTraits code the host has verified against the argument block (0 = generic)
// p.stamps != nullptr selects the stamped diagnostic instantiation where one exists (8-phase: nf 6 / 8 with epi 1 / 3; pair: epi 1 / 3)
int ufm_launch_gemm_8ph(const GemmArgs& p, int out_dtype, hipStream_t stream, int nf = 8, int epi = 0);
// gemm_bf16_8ph_persist.hip: the persistent form for bf16-output launches of whole tiles (epi 1 / 2), next tile's prologue under the epilogue
int ufm_launch_gemm_8ph_persist(const GemmArgs& p, hipStream_t stream, int epi, int ncu);
// gemm_bf16_pair.hip: 256x128 4-wave kernel, two resident workgroups per CU (N % 128 == 0, K >= 128, 32-bit operand offsets)
int ufm_launch_gemm_pair(const GemmArgs& p, int out_dtype, hipStream_t stream, int nf = 8, int epi = 0);
