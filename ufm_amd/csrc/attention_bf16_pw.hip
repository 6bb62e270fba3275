// Flash attention forward, head_dim 64, bf16 operands with Q pre-scaled by softmax_scale*log2(e):
// the "one wave per SIMD, 64 query rows per wave" structure (cdna_hip_programming.md Appendix B,
// "4-wave, one-wave-per-SIMD, persistent structure"), re-balanced for head_dim 64 where the softmax
// VALU work, not the matrix pipe, is the longer pole (16 MFMA = 512 pipe cycles against ~460 cycles of
// VALU issue + 128 of MFMA issue per 32 queries x 64 keys).
//
//   * workgroup = NW waves, each wave owns 64 query rows = TWO 32-row q-blocks (A, B) of one
//     (image, head) and the whole register file (__launch_bounds__(.., 1)); the q-blocks run half a
//     key tile out of phase so that in every quarter of a tile ("slot") the wave issues the 8 MFMAs of
//     one q-block beside the exp/sum/convert stream of the other:
//         slot 0   QK^T_A(t)    | softmax_B(t-1), keys 32..63      -> check B
//         slot 1   P.V_B(t-1)   | softmax_A(t),   keys  0..31
//         slot 2   QK^T_B(t)    | softmax_A(t),   keys 32..63      -> check A   (+ V(t) fragment reads)
//         slot 3   P.V_A(t)     | softmax_B(t),   keys  0..31                   (+ K(t+1) fragment reads)
//     K and V fragments are read from LDS once per wave and tile and feed both q-blocks.
//   * swapped QK^T (S^T = K.Q^T, lane = one query column), P stays in registers as the B operand of
//     O^T += V^T.P^T, V^T by ds_read_b64_tr_b16 -- the fragment maps of attention_bf16.hip.
//   * NO per-tile row maximum.  The reference -m_ref (carried as the C operand of the first QK^T MFMA)
//     is set to the exact row maximum of the first key tile and afterwards moved only when needed:
//     floating point is scale free, so P = 2^(s - m_ref) may be large; what must not happen is overflow.
//     The per-lane tile sum that l needs anyway is the detector: if it exceeds 2^64 (or is inf/NaN) for
//     any query of the wave, the wave takes a cold path that finds the tile maximum, rescales O, l and
//     m_ref and re-exponentiates this tile (nothing of the tile has been accumulated yet).  That removes
//     the 16 v_max3 + swap + compare per tile of the classical online softmax.
//   * K/V tiles (64 keys x 128 B) arrive by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave
//     instruction) into 2-deep rings, K two tiles and V one tile ahead of their first use; one
//     s_barrier per key tile; swizzles on the DMA source address and on the reads (rule 21).
//   * O is normalised, staged through LDS and stored as whole 128-B rows.
#include <type_traits>

#include "common.h"

namespace {

constexpr int KB = 64;  // keys per tile
constexpr float NEG_BIG = -1.0e30f;
constexpr float SUM_BIG = 1.8446744e19f;  // 2^64
constexpr int KOFF = 0, VOFF = 16384, OOFF = 32768, OSTRIDE = 144;

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;
__device__ __forceinline__ bf16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_ptr)LDS_PTR(p));
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int V>
using IC = std::integral_constant<int, V>;

// ---- MFMA by inline asm: hipcc picks ONE accumulator form per kernel (with a 512-register budget: AGPR C/D for every
// MFMA, so each score would cost an extra v_accvgpr_read before its v_exp).  Here the register classes are stated:
//   S^T = K.Q^T   D/C in VGPRs (the softmax reads them),  K and Q fragments in AGPRs
//   O^T += V^T.P^T   D/C in AGPRs (only MFMAs touch O),   V fragments in AGPRs, P in VGPRs
// Hazards the compiler no longer sees are covered by placement (a QK^T group is always followed by an MFMA of the other
// q-block before any VALU reads its result) or by explicit s_nop (pad16) on the cold / first / last paths.
// O^T accumulators live in FIXED accumulator registers a[0:63] (q-block A: a[0:31], B: a[32:63]; 16 per 32-wide d tile),
// named literally and listed as clobbers of every MFMA statement, so the compiler keeps nothing else there
// (cdna_hip_programming.md section 5.7 item 4; audited by tools/check_attn_isa.py at build time).
#define UFM_O_CLOBBERS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"
template <bool KV>  // KV: K fragments in VGPRs (experiment), else AGPRs
__device__ __forceinline__ void mfma_qk_first(f32x16& d, const bf16x8& k, const bf16x8& q, const f32x16& c) {
    if constexpr (KV) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(k), "a"(q), "v"(c) : UFM_O_CLOBBERS);
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "a"(k), "a"(q), "v"(c) : UFM_O_CLOBBERS);
}
template <bool KV>
__device__ __forceinline__ void mfma_qk(f32x16& d, const bf16x8& k, const bf16x8& q) {
    if constexpr (KV) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(k), "a"(q) : UFM_O_CLOBBERS);
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(k), "a"(q) : UFM_O_CLOBBERS);
}
#define UFM_PV_CASE(IDX, LO, HI)                                                                                   \
    if constexpr (O == IDX) {                                                                                      \
        if constexpr (PAD)                                                                                         \
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[" #LO ":" #HI "], %0, %1, a[" #LO ":" #HI "]" ::"v"(v), "v"(p) : UFM_O_CLOBBERS); \
        else                                                                                                       \
            asm volatile("v_mfma_f32_32x32x16_bf16 a[" #LO ":" #HI "], %0, %1, a[" #LO ":" #HI "]" ::"v"(v), "v"(p) : UFM_O_CLOBBERS);         \
    }
template <int O, bool PAD>  // O = 2*q-block + d-tile; PAD: first of a P.V group (P was just written by VALU)
__device__ __forceinline__ void mfma_pv(const bf16x8& v, const bf16x8& p) {
    UFM_PV_CASE(0, 0, 15) UFM_PV_CASE(1, 16, 31) UFM_PV_CASE(2, 32, 47) UFM_PV_CASE(3, 48, 63)
}
// one accumulator register: zero / scale by a per-lane factor / read out
template <int R>
__device__ __forceinline__ void o_zero() {
    asm volatile("v_accvgpr_write_b32 a%0, 0" ::"n"(R) : UFM_O_CLOBBERS);
}
template <int R>
__device__ __forceinline__ void o_scale(float f) {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a%2\n\ts_nop 1\n\tv_mul_f32 %0, %0, %1\n\ts_nop 1\n\tv_accvgpr_write_b32 a%2, %0" : "=&v"(t) : "v"(f), "n"(R) : UFM_O_CLOBBERS);
}
template <int R>
__device__ __forceinline__ float o_read() {
    float t;
    asm volatile("v_accvgpr_read_b32 %0, a%1" : "=v"(t) : "n"(R));
    return t;
}
template <int I = 0>
__device__ __forceinline__ void o_read_all(float (&ov)[2][2][16]) {
    if constexpr (I < 64) {
        ov[I >> 5][(I >> 4) & 1][I & 15] = o_read<I>();
        o_read_all<I + 1>(ov);
    }
}
template <int R0, int N, int I = 0>
__device__ __forceinline__ void o_zero_range() {
    static_assert(R0 == 0 && N == 64, "one statement zeroes all of a[0:63]");
    asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\tv_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\tv_accvgpr_write_b32 a10, 0\n\tv_accvgpr_write_b32 a11, 0\n\tv_accvgpr_write_b32 a12, 0\n\tv_accvgpr_write_b32 a13, 0\n\tv_accvgpr_write_b32 a14, 0\n\tv_accvgpr_write_b32 a15, 0\n\tv_accvgpr_write_b32 a16, 0\n\tv_accvgpr_write_b32 a17, 0\n\tv_accvgpr_write_b32 a18, 0\n\tv_accvgpr_write_b32 a19, 0\n\tv_accvgpr_write_b32 a20, 0\n\tv_accvgpr_write_b32 a21, 0\n\tv_accvgpr_write_b32 a22, 0\n\tv_accvgpr_write_b32 a23, 0\n\tv_accvgpr_write_b32 a24, 0\n\tv_accvgpr_write_b32 a25, 0\n\tv_accvgpr_write_b32 a26, 0\n\tv_accvgpr_write_b32 a27, 0\n\tv_accvgpr_write_b32 a28, 0\n\tv_accvgpr_write_b32 a29, 0\n\tv_accvgpr_write_b32 a30, 0\n\tv_accvgpr_write_b32 a31, 0\n\tv_accvgpr_write_b32 a32, 0\n\tv_accvgpr_write_b32 a33, 0\n\tv_accvgpr_write_b32 a34, 0\n\tv_accvgpr_write_b32 a35, 0\n\tv_accvgpr_write_b32 a36, 0\n\tv_accvgpr_write_b32 a37, 0\n\tv_accvgpr_write_b32 a38, 0\n\tv_accvgpr_write_b32 a39, 0\n\tv_accvgpr_write_b32 a40, 0\n\tv_accvgpr_write_b32 a41, 0\n\tv_accvgpr_write_b32 a42, 0\n\tv_accvgpr_write_b32 a43, 0\n\tv_accvgpr_write_b32 a44, 0\n\tv_accvgpr_write_b32 a45, 0\n\tv_accvgpr_write_b32 a46, 0\n\tv_accvgpr_write_b32 a47, 0\n\tv_accvgpr_write_b32 a48, 0\n\tv_accvgpr_write_b32 a49, 0\n\tv_accvgpr_write_b32 a50, 0\n\tv_accvgpr_write_b32 a51, 0\n\tv_accvgpr_write_b32 a52, 0\n\tv_accvgpr_write_b32 a53, 0\n\tv_accvgpr_write_b32 a54, 0\n\tv_accvgpr_write_b32 a55, 0\n\tv_accvgpr_write_b32 a56, 0\n\tv_accvgpr_write_b32 a57, 0\n\tv_accvgpr_write_b32 a58, 0\n\tv_accvgpr_write_b32 a59, 0\n\tv_accvgpr_write_b32 a60, 0\n\tv_accvgpr_write_b32 a61, 0\n\tv_accvgpr_write_b32 a62, 0\n\tv_accvgpr_write_b32 a63, 0" ::: UFM_O_CLOBBERS);
}
template <int R0, int N, int I = 0>
__device__ __forceinline__ void o_scale_range(float f) {
    if constexpr (I < N) {
        o_scale<R0 + I>(f);
        o_scale_range<R0, N, I + 1>(f);
    }
}
// LDS-DMA by inline asm (saddr + 32-bit per-lane offset form): hipcc orders every later ds_read behind a builtin
// global_load_lds with s_waitcnt vmcnt(0) (it cannot prove the LDS ranges disjoint), which would serialise the DMA latency
// into the slot that issues it.  Here the counting is by hand: one wait_vm<0>() per key tile, in front of the barrier.
// M0 = LDS byte address of the piece (wave uniform); the s_nop is the M0-write -> LDS-DMA wait state.  M0 is written in the
// same statement that uses it; the kernel has no compiler-generated M0 use (checked by tools/check_attn_isa.py).
__device__ __forceinline__ void glds16(const char* gbase, unsigned voff, unsigned lds_addr) {
    // leading s_nop 4: gbase / lds_addr may have been reloaded from a spilled SGPR by v_readlane right in front of the
    // statement (VALU write of an SGPR -> VMEM read of it needs 5 wait states; nothing inside an asm string is padded)
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(gbase) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const char* p) { return (unsigned)(uintptr_t)LDS_PTR(p); }
__device__ __forceinline__ void pad16() { asm volatile("s_nop 15\n\ts_nop 3" ::: "memory"); }
__device__ __forceinline__ void gap() { __builtin_amdgcn_sched_barrier(0); }

__device__ __forceinline__ unsigned long long stamp() {  // cdna_hip_programming.md section 7, "In-kernel stamps" (diagnostic build only)
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

// q / k / v: first element of head 0 of batch item 0; ld*: row strides (elements); *_bs: rows between consecutive batch items
struct PwArgs {
    const uint16_t *q, *k, *v;
    uint16_t* out;
    int ldq, ldkv, ldo, q_bs, kv_bs, o_bs, Nq, Nk, H, nqb, nunits;
};

template <int NW, bool DIAG>
__global__ __launch_bounds__(NW * 64, 1) void attn_pw_kernel(PwArgs a_, unsigned long long* __restrict__ diag) {
    // two sources (self-attention: q / k / v are column blocks of one qkv buffer): N = keys per batch item, Nq = query rows
    const uint16_t* __restrict__ qsrc = a_.q;
    uint16_t* __restrict__ out = a_.out;
    const int N = a_.Nk, Nq = a_.Nq, H = a_.H, nqb = a_.nqb, nunits = a_.nunits;
    unsigned long long seg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0, t_begin = 0, rt_begin = 0;  // 0-4 key-tile slots, 5-10 unit seams
    if (DIAG) {
        t_begin = stamp();
        rt_begin = __builtin_amdgcn_s_memrealtime();
    }
#define UFM_STAMP(K)                                  \
    if (DIAG) {                                       \
        const unsigned long long now_ = stamp();      \
        seg[K] += now_ - t_prev;                      \
        t_prev = now_;                                \
    }
    constexpr int QB = NW * 64;
    constexpr int NP = 8 / NW;  // 1-KiB DMA pieces per wave for one 8-KiB tile
    __shared__ __attribute__((aligned(16))) char smem[OOFF + NW * 64 * OSTRIDE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Persistent: the workgroup walks units (image, head, 64*NW-row query block) u, u + gridDim.x, ...; the units of one
    // (image, head) are consecutive logical ids and, through xcd_remap, concurrent on one XCD (shared K/V stream in its L2).
    int unit = xcd_remap(blockIdx.x, gridDim.x);
    const int ld = a_.ldq;               // query row stride (elements)
    const unsigned ldb = 2u * a_.ldkv;   // key / value row stride (bytes)
    const int ql = lane & 31, hh = lane >> 5;
    const int nt = (N + KB - 1) / KB;
    const uint16_t* base;
    const char *kp, *vp;
    uint16_t* og;
    int q0;
    auto set_unit = [&](int u) __attribute__((always_inline)) {
        const int qblk = u % nqb, head = (u / nqb) % H, b = u / (nqb * H);
        base = qsrc + (size_t)b * a_.q_bs * ld + head * 64;
        kp = (const char*)(a_.k + (size_t)b * a_.kv_bs * a_.ldkv + head * 64);
        vp = (const char*)(a_.v + (size_t)b * a_.kv_bs * a_.ldkv + head * 64);
        q0 = qblk * QB + wave * 64;
        og = out + ((size_t)b * a_.o_bs + q0) * a_.ldo + head * 64;
    };
    set_unit(unit);

    // ---- Q fragments (B operand of S^T = K.Q^T): lane = query column ----
    bf16x8 qf[2][4];
    auto load_q = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const uint16_t* qr = base + (size_t)min(q0 + 32 * a + ql, Nq - 1) * ld + 8 * hh;
#pragma unroll
            for (int s = 0; s < 4; ++s) qf[a][s] = *(const bf16x8*)(qr + 16 * s);
        }
    };
    load_q();

    // ---- LDS-DMA source offsets: piece = 8 rows x 128 B, lane -> (row lane>>3, LDS slot lane&7) ----
    const int prow = lane >> 3, pslot = lane & 7;
    int dma_row[NP];
    unsigned dma_kc[NP], dma_vc[NP], dma_ko[NP], dma_vo[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int r = 8 * (wave + NW * i) + prow;
        dma_row[i] = r;
        dma_kc[i] = (unsigned)((pslot ^ ((r >> 1) & 7)) << 4);
        dma_vc[i] = (unsigned)((pslot ^ (((r >> 1) & 1) << 2)) << 4);
        dma_ko[i] = (unsigned)r * ldb + dma_kc[i];
        dma_vo[i] = (unsigned)r * ldb + dma_vc[i];
    }
    auto dma = [&](const char* gsrc, const unsigned (&chunk)[NP], const unsigned (&full)[NP], char* lbuf, int tt) __attribute__((always_inline)) {
        const char* g = gsrc + (size_t)tt * KB * ldb;
        if (tt * KB + KB <= N) {
#pragma unroll
            for (int i = 0; i < NP; ++i)
                glds16(g, full[i], __builtin_amdgcn_readfirstlane(lds_addr_of(lbuf + (wave + NW * i) * 1024)));
        } else {  // ragged last tile: clamp the row (masked in the scores)
            const int last = N - 1 - tt * KB;
#pragma unroll
            for (int i = 0; i < NP; ++i)
                glds16(g, (unsigned)min(dma_row[i], last) * ldb + chunk[i], __builtin_amdgcn_readfirstlane(lds_addr_of(lbuf + (wave + NW * i) * 1024)));
        }
    };

    // ---- fragment read addresses (byte offsets inside a tile buffer) ----
    int k_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) k_off[s] = ql * 128 + (((2 * s + hh) ^ ((ql >> 1) & 7)) << 4);  // key tile kt: + 4096*kt
    const int ti = lane & 15, tq = ti >> 2, tp = ti & 3;
    int v_off[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const int row = 4 * hh + tq, dcol = dt * 32 + 16 * ((lane >> 4) & 1) + 4 * tp;
        v_off[dt] = row * 128 + (((dcol >> 3) ^ (((row >> 1) & 1) << 2)) << 4) + ((dcol & 7) << 1);  // + 128*(32kt + 16s2 + 8e)
    }

    f32x16 negm[2], st[2][2];  // O^T: a[0:63], asm-owned
    bf16x8 pf[2][2][2], kf[2][4], vf[2][2][2];
    float lrun[2] = {0.f, 0.f}, ls[2] = {0.f, 0.f}, ls2[2] = {0.f, 0.f};
    float ev[16];     // exponentials of the half being processed (software pipelined one MFMA gap deep)
    unsigned pk[8];   // its packed bf16 pairs

    // fragment g (0..7) of the K tile: kf[g >> 2][g & 3];  tr-read pair g (0..7) of the V tile: vf[dt][kt][s2], g = 4dt + 2kt + s2
    auto read_k1 = [&](const char* kb, auto g_) __attribute__((always_inline)) {
        constexpr int G = decltype(g_)::value;
        kf[G >> 2][G & 3] = *(const bf16x8*)(kb + k_off[G & 3] + 4096 * (G >> 2));
    };
    auto read_v1 = [&](const char* vb, auto g_) __attribute__((always_inline)) {
        constexpr int G = decltype(g_)::value, DT = G >> 2, KT = (G >> 1) & 1, S2 = G & 1;
        const bf16x4 lo = tr_read(vb + v_off[DT] + 128 * (32 * KT + 16 * S2));
        const bf16x4 hi = tr_read(vb + v_off[DT] + 128 * (32 * KT + 16 * S2 + 8));
        vf[DT][KT][S2] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    // MFMA g (0..7) of S^T(a) = K.Q^T(a) - m_ref(a) / of O^T(a) += V^T.P^T(a)
    auto qk1 = [&](auto a_, auto g_) __attribute__((always_inline)) {
        constexpr int A = decltype(a_)::value, G = decltype(g_)::value, KT = G >> 2, S = G & 3;
        if (S == 0) mfma_qk_first<false>(st[A][KT], kf[KT][0], qf[A][0], negm[A]);
        else mfma_qk<false>(st[A][KT], kf[KT][S], qf[A][S]);
    };
    auto pv1 = [&](auto a_, auto g_) __attribute__((always_inline)) {
        constexpr int A = decltype(a_)::value, G = decltype(g_)::value, DT = G >> 2, KT = (G >> 1) & 1, S2 = G & 1;
        mfma_pv<2 * A + DT, G == 0>(vf[DT][KT][S2], pf[A][KT][S2]);
    };
    // step i (0..8) of P = 2^S for the 32-key half (a, kt): two new exponentials, sum + pack of the previous pair
    auto sm1 = [&](auto a_, auto kt_, auto i_) __attribute__((always_inline)) {
        constexpr int A = decltype(a_)::value, KT = decltype(kt_)::value, I = decltype(i_)::value;
        if constexpr (I >= 1) {  // two independent partial sums: a serial v_add chain issues at its dependent latency
            ls[A] += ev[2 * I - 2];
            ls2[A] += ev[2 * I - 1];
            pk[I - 1] = pack_bf16x2(ev[2 * I - 2], ev[2 * I - 1]);
        }
        if constexpr (I < 8) {
            ev[2 * I] = __builtin_amdgcn_exp2f(st[A][KT][2 * I]);
            ev[2 * I + 1] = __builtin_amdgcn_exp2f(st[A][KT][2 * I + 1]);
        }
        if constexpr (I == 8) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) pf[A][KT][s2] = __builtin_bit_cast(bf16x8, u32x4{pk[4 * s2], pk[4 * s2 + 1], pk[4 * s2 + 2], pk[4 * s2 + 3]});
            asm volatile("" ::"v"(pf[A][KT][0]), "v"(pf[A][KT][1]), "v"(ls[A]), "v"(ls2[A]));  // produced HERE (no sinking into later blocks)
        }
    };
    auto sm_all = [&](auto a_, auto kt_) __attribute__((always_inline)) {  // un-interleaved form (drain)
        sm1(a_, kt_, IC<0>{}); sm1(a_, kt_, IC<1>{}); sm1(a_, kt_, IC<2>{}); sm1(a_, kt_, IC<3>{}); sm1(a_, kt_, IC<4>{});
        sm1(a_, kt_, IC<5>{}); sm1(a_, kt_, IC<6>{}); sm1(a_, kt_, IC<7>{}); sm1(a_, kt_, IC<8>{});
    };
    // cold path: both halves of q-block a again, results moved INTO the registers the hot path uses (tied asm operands), so
    // the merge after the cold path needs no copies on the hot edge
    auto sm_cold = [&](auto a_) __attribute__((always_inline)) {
        constexpr int A = decltype(a_)::value;
        float acc = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4 cur = __builtin_bit_cast(u32x4, pf[A][kt][s2]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float p0 = __builtin_amdgcn_exp2f(st[A][kt][8 * s2 + 2 * j]);
                    const float p1 = __builtin_amdgcn_exp2f(st[A][kt][8 * s2 + 2 * j + 1]);
                    acc += p0;
                    acc += p1;
                    unsigned x = cur[j];
                    asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(pack_bf16x2(p0, p1)));
                    cur[j] = x;
                }
                pf[A][kt][s2] = __builtin_bit_cast(bf16x8, cur);
            }
        float l = ls[A];
        asm volatile("v_mov_b32 %0, %1" : "+v"(l) : "v"(acc));
        ls[A] = l;
    };
    auto row_max = [&](auto a_) __attribute__((always_inline)) -> float {
        constexpr int A = decltype(a_)::value;
        float mx = fmaxf(st[A][0][0], st[A][1][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(st[A][0][r], st[A][1][r]));
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    };
    auto shift = [&](auto a_, float d) __attribute__((always_inline)) {  // move the reference of q-block a up by d (per query)
        constexpr int A = decltype(a_)::value;
        // in place (tied operands): the hot path keeps its registers, no copies at the merge
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float x = negm[A][r], y = st[A][0][r], z = st[A][1][r];
            asm volatile("v_sub_f32 %0, %0, %3\n\tv_sub_f32 %1, %1, %3\n\tv_sub_f32 %2, %2, %3" : "+v"(x), "+v"(y), "+v"(z) : "v"(d));
            negm[A][r] = x;
            st[A][0][r] = y;
            st[A][1][r] = z;
        }
    };
    // after both halves of softmax(a): accept the tile sum, or (cold) move the reference and redo the tile
    auto check = [&](auto a_) __attribute__((always_inline)) {
        constexpr int A = decltype(a_)::value;
        ls[A] += ls2[A];
        ls2[A] = 0.f;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(ls[A] <= SUM_BIG)) != 0, 0)) {
            pad16();
            const float d = fmaxf(row_max(a_), 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-d);
            {
                float l = lrun[A];
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(l) : "v"(alpha));
                lrun[A] = l;
            }
            o_scale_range<32 * A, 32>(alpha);
            shift(a_, d);
            sm_cold(a_);
            pad16();
        }
        lrun[A] += ls[A];
        ls[A] = 0.f;
    };
    auto mask_tail = [&](auto a_, int t) __attribute__((always_inline)) {  // keys >= N of the last tile
        constexpr int A = decltype(a_)::value;
        pad16();
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * KB + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (key >= N) st[A][kt][r] = NEG_BIG;
            }
    };
    auto anchor = [&](auto a_) __attribute__((always_inline)) {  // first tile: reference = exact row maximum
        pad16();
        const float mx = row_max(a_);
        shift(a_, mx);
        pad16();
    };

    // one MFMA gap: MFMA g of `what` + one softmax step + LDS fragment reads, fenced so that the order below is the order issued
    char* const ob = smem + OOFF + wave * 64 * OSTRIDE;  // this wave's O staging rows
    // one 1-KiB piece of tile tt's K (ISV = 0) or V (1) image; SAFE: tt is a full tile known to exist
    auto dma_piece = [&](auto isv_, auto i_, auto safe_, char* lbuf, int tt) __attribute__((always_inline)) {
        constexpr int ISV = decltype(isv_)::value, I = decltype(i_)::value;
        constexpr bool SAFE = decltype(safe_)::value != 0;
        if constexpr (I < NP) {
            const char* g = (ISV ? vp : kp) + (size_t)tt * KB * ldb;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr_of(lbuf + (wave + NW * I) * 1024));
            if (SAFE || tt * KB + KB <= N) {
                if (SAFE || tt < nt) glds16(g, ISV ? dma_vo[I] : dma_ko[I], dst);
            } else if (tt < nt) {  // ragged last tile: clamp the row (masked in the scores)
                glds16(g, (unsigned)min(dma_row[I], N - 1 - tt * KB) * ldb + (ISV ? dma_vc[I] : dma_kc[I]), dst);
            }
        }
    };
    // DMA issue points: piece G/DSTEP in gap G when G % DSTEP == 1 (4 waves: gaps 1, 5; 2 waves: 1, 3, 5, 7), so the
    // vector-memory unit sees the tile's 16 KiB spread over two slots instead of 16 pieces at the barrier
    constexpr int DSTEP = 8 / NP;

    // one key tile.  Every MFMA gap is fenced (gap()) so that the order below is the order issued.
    auto iter = [&](int t, auto par_, auto first_, auto safe_) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_)::value;
        constexpr bool FIRST = decltype(first_)::value != 0, SAFE = decltype(safe_)::value != 0;
        char* kb_cur = smem + KOFF + PAR * 8192;
        char* kb_nxt = smem + KOFF + (PAR ^ 1) * 8192;
        char* vb_cur = smem + VOFF + PAR * 8192;
        char* vb_nxt = smem + VOFF + (PAR ^ 1) * 8192;
        const bool ragged_last = !SAFE && (t + 1) * KB > N;
        if (DIAG) t_prev = stamp();
        if (!FIRST) {  // (the first key tile was synchronised by the unit prologue)
            // own DMA pieces of the previous iteration have landed; every wave has read the fragments of the buffers re-filled below
            wait_vm<0>();
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
            gap();
            __builtin_amdgcn_s_barrier();
            gap();
        }
        UFM_STAMP(0)
#define UFM_GAPS(BODY) BODY(0) BODY(1) BODY(2) BODY(3) BODY(4) BODY(5) BODY(6) BODY(7)
        // ---- slot 0: QK^T_A(t) | softmax_B(t-1) keys 32..63 | DMA K(t+2) ----
#define UFM_S0(G)                                                                          \
    qk1(IC<0>{}, IC<G>{});                                                                 \
    if (!FIRST) sm1(IC<1>{}, IC<1>{}, IC<G>{});                                            \
    if (G % DSTEP == 1) dma_piece(IC<0>{}, IC<G / DSTEP>{}, safe_, kb_cur, t + 2);         \
    gap();
        UFM_GAPS(UFM_S0)
        if (!FIRST) {
            sm1(IC<1>{}, IC<1>{}, IC<8>{});
            check(IC<1>{});
        }
        if (ragged_last) mask_tail(IC<0>{}, t);
        if (FIRST) anchor(IC<0>{});
        gap();
        UFM_STAMP(1)
        // ---- slot 1: P.V_B(t-1) | softmax_A(t) keys 0..31 | V(t) fragments behind the MFMAs that free them | DMA V(t+1) ----
#define UFM_S1(G)                                                                          \
    if (!FIRST) pv1(IC<1>{}, IC<G>{});                                                     \
    sm1(IC<0>{}, IC<0>{}, IC<G>{});                                                        \
    if (G >= 1) read_v1(vb_cur, IC<(G >= 1 ? G - 1 : 0)>{});                              \
    if (G % DSTEP == 1) dma_piece(IC<1>{}, IC<G / DSTEP>{}, safe_, vb_nxt, t + 1);         \
    gap();
        UFM_GAPS(UFM_S1)
        UFM_STAMP(2)
        // ---- slot 2: QK^T_B(t) | softmax_A(t) keys 32..63 | K(t+1) fragments behind the MFMAs that free them ----
#define UFM_S2(G)                                                                          \
    qk1(IC<1>{}, IC<G>{});                                                                 \
    if (G == 0) {                                                                          \
        sm1(IC<0>{}, IC<0>{}, IC<8>{});                                                    \
        read_v1(vb_cur, IC<7>{});                                                          \
    }                                                                                      \
    sm1(IC<0>{}, IC<1>{}, IC<G>{});                                                        \
    if (G >= 1) read_k1(kb_nxt, IC<(G >= 1 ? G - 1 : 0)>{});                              \
    gap();
        UFM_GAPS(UFM_S2)
        sm1(IC<0>{}, IC<1>{}, IC<8>{});
        check(IC<0>{});
        if (ragged_last) mask_tail(IC<1>{}, t);
        if (FIRST) anchor(IC<1>{});
        gap();
        UFM_STAMP(3)
        // ---- slot 3: P.V_A(t) | softmax_B(t) keys 0..31 ----
#define UFM_S3(G)                                                                          \
    pv1(IC<0>{}, IC<G>{});                                                                 \
    if (G == 0) read_k1(kb_nxt, IC<7>{});                                                  \
    sm1(IC<1>{}, IC<0>{}, IC<G>{});                                                        \
    gap();
        UFM_GAPS(UFM_S3)
        sm1(IC<1>{}, IC<0>{}, IC<8>{});
        gap();
        UFM_STAMP(4)
#undef UFM_S0
#undef UFM_S1
#undef UFM_S2
#undef UFM_S3
    };

    auto first_tiles = [&]() __attribute__((always_inline)) {  // K(0), V(0), K(1) of the current unit into the rings
        dma(kp, dma_kc, dma_ko, smem + KOFF, 0);
        dma(vp, dma_vc, dma_vo, smem + VOFF, 0);
        if (nt > 1) dma(kp, dma_kc, dma_ko, smem + KOFF + 8192, 1);
    };
    first_tiles();
    int units_done = 0;
    bool stores_in_flight = false;  // wave-uniform: the previous unit ended with exactly 8 (unconditional) store instructions
    for (;;) {
        if (DIAG) t_prev = stamp();
        // ---- unit prologue: state, first tiles landed and visible, K(0) fragments ----
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            lrun[a] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[a][r] = 0.f;
        }
        o_zero_range<0, 64>();
        UFM_STAMP(5)
        // The first tiles and Q of this unit were requested BEFORE the previous unit's 8 O-row stores (vmcnt is in issue
        // order): waiting for all but the 8 youngest operations leaves those stores draining under this unit's first key
        // tile instead of in front of it (a CU's store path takes ~100 cycles per wave-instruction).
        if (stores_in_flight) wait_vm<8>();
        else wait_vm<0>();
        gap();
        __builtin_amdgcn_s_barrier();
        gap();
        read_k1(smem + KOFF, IC<0>{}); read_k1(smem + KOFF, IC<1>{}); read_k1(smem + KOFF, IC<2>{}); read_k1(smem + KOFF, IC<3>{});
        read_k1(smem + KOFF, IC<4>{}); read_k1(smem + KOFF, IC<5>{}); read_k1(smem + KOFF, IC<6>{}); read_k1(smem + KOFF, IC<7>{});
        // Tile 0's slot 0 re-fills K buffer 0 with K(2): every wave's K(0) fragments must be in registers first.  (Without
        // this second barrier a wave delayed between the barrier above and its reads -- a cold instruction cache on a
        // workgroup's first unit, another stream's kernels on the chip -- read rows of K(2) as K(0).)
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        gap();
        __builtin_amdgcn_s_barrier();
        gap();
        UFM_STAMP(6)

        iter(0, IC<0>{}, IC<1>{}, IC<0>{});
        {
            int t = 1;
            for (; t + 4 < nt; t += 2) {  // tiles t+2 / t+3 exist and are full: no conditions around the DMA
                iter(t, IC<1>{}, IC<0>{}, IC<1>{});
                iter(t + 1, IC<0>{}, IC<0>{}, IC<1>{});
            }
            for (; t < nt; t += 2) {
                iter(t, IC<1>{}, IC<0>{}, IC<0>{});
                if (t + 1 < nt) iter(t + 1, IC<0>{}, IC<0>{}, IC<0>{});
            }
        }
        if (DIAG) t_prev = stamp();
        // ---- seam: the NEXT unit's first tiles and Q are requested now and land under this unit's drain + epilogue ----
        uint16_t* const og_cur = og;
        const int q0_cur = q0;
        const int next = unit + (int)gridDim.x;
        const bool has_next = next < nunits;
        ++units_done;
        if (has_next) {
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's last K/V fragment reads are back ...
            gap();
            __builtin_amdgcn_s_barrier();        // ... and so are every other wave's: the rings are free
            gap();
            set_unit(next);
            first_tiles();
            load_q();  // q fragments are dead after the last QK^T MFMA; the drain below only reads V and P
        }
        UFM_STAMP(7)
        // ---- drain: second half of q-block B's last tile ----
        sm_all(IC<1>{}, IC<1>{});
        check(IC<1>{});
#define UFM_DR(G) pv1(IC<1>{}, IC<G>{});
        UFM_GAPS(UFM_DR)
#undef UFM_DR
        pad16();

        UFM_STAMP(8)
        // ---- epilogue: O[q][d] = O^T[d][q] / l, staged through this wave's LDS rows, stored as 128-B rows ----
        float ov[2][2][16];
        o_read_all(ov);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lrun[a]), __float_as_uint(lrun[a]), false, false);
            const float inv = 1.0f / (__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    u32x2 w = {pack_bf16x2(ov[a][dt][4 * g] * inv, ov[a][dt][4 * g + 1] * inv),
                               pack_bf16x2(ov[a][dt][4 * g + 2] * inv, ov[a][dt][4 * g + 3] * inv)};
                    *(u32x2*)(ob + (32 * a + ql) * OSTRIDE + 2 * (32 * dt + 8 * g + 4 * hh)) = w;
                }
        }
        UFM_STAMP(9)
        // Exactly 8 store instructions per wave when all its 64 rows exist (the count the next unit's vmcnt(8) relies on)
        stores_in_flight = q0_cur + 64 <= Nq;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * i + prow;
            const u32x4 v = *(const u32x4*)(ob + row * OSTRIDE + 16 * pslot);
            if (stores_in_flight || q0_cur + row < Nq) *(u32x4*)(og_cur + (size_t)row * a_.ldo + 8 * pslot) = v;
        }
        UFM_STAMP(10)
        if (!has_next) break;
        unit = next;
    }
#undef UFM_GAPS
    if (DIAG) {  // stamps leave through a buffer of their own, never through an output element
        const unsigned long long t_end = stamp();
        const unsigned long long rt_end = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            unsigned long long* d = diag + (size_t)blockIdx.x * 16;
#pragma unroll
            for (int k = 0; k < 5; ++k) d[k] = seg[k];
            d[5] = t_end - t_begin;
            d[6] = rt_end - rt_begin;
            d[7] = (unsigned long long)nt * units_done;
            d[8] = (unsigned long long)units_done;
#pragma unroll
            for (int k = 0; k < 6; ++k) d[9 + k] = seg[5 + k];
        }
    }
#undef UFM_STAMP
}

}  // namespace

static int attn_pw_grid(int nunits, int nw) {  // persistent: one workgroup of 4 waves (two of 2 waves) per compute unit
    const int ncu = ufm_device_cu_count();
    const int slots = ncu * (nw == 4 ? 1 : 2);
    return nunits < slots ? nunits : slots;
}

static PwArgs pw_self_args(const uint16_t* qkv, uint16_t* out, int B, int N, int H, int nw) {
    const int qb = nw * 64, nqb = (N + qb - 1) / qb;
    return PwArgs{qkv, qkv + H * 64, qkv + 2 * H * 64, out, 3 * H * 64, 3 * H * 64, H * 64, N, N, N, N, N, H, nqb, nqb * H * B};
}

int ufm_launch_attn_pw(const uint16_t* qkv, uint16_t* out, int B, int N, int H, int variant, hipStream_t stream) {
    const int nw = (variant & 1) ? 2 : 4;
    const PwArgs a = pw_self_args(qkv, out, B, N, H, nw);
    dim3 grid(attn_pw_grid(a.nunits, nw)), block(nw * 64);
    if (nw == 4) hipLaunchKernelGGL((attn_pw_kernel<4, false>), grid, block, 0, stream, a, nullptr);
    else hipLaunchKernelGGL((attn_pw_kernel<2, false>), grid, block, 0, stream, a, nullptr);
    return 0;
}

// Two-source form (cross-attention of the "cross_attention" info-sharing variant, ufm.py:193; the last joint-attention layer's
// view-1 queries against both views' keys): the same kernel, queries / keys+values / output with their own base pointers,
// row strides and per-batch-item row counts.  Q must be pre-scaled by softmax_scale * log2(e) (the projection's epilogue).
int ufm_launch_attn_pw2(const uint16_t* q, int ldq, int q_bs, const uint16_t* k, const uint16_t* v, int ldkv, int kv_bs, uint16_t* out, int ldo, int o_bs,
                        int B, int Nq, int Nk, int H, int variant, hipStream_t stream) {
    const int nw = (variant & 1) ? 2 : 4;
    const int qb = nw * 64, nqb = (Nq + qb - 1) / qb;
    const PwArgs a{q, k, v, out, ldq, ldkv, ldo, q_bs, kv_bs, o_bs, Nq, Nk, H, nqb, nqb * H * B};
    dim3 grid(attn_pw_grid(a.nunits, nw)), block(nw * 64);
    if (nw == 4) hipLaunchKernelGGL((attn_pw_kernel<4, false>), grid, block, 0, stream, a, nullptr);
    else hipLaunchKernelGGL((attn_pw_kernel<2, false>), grid, block, 0, stream, a, nullptr);
    return 0;
}

// Diagnostics (tools/attn_stamps.py): the same kernel with s_memtime stamps around the four slots of every key tile.
// diag: 16 x uint64 per workgroup = {sync + DMA issue, slot 0..3, whole kernel, s_memrealtime ticks (100 MHz), tiles, units,
//       state init, prologue wait + K reads, seam, drain, epilogue compute, store issue, 0}.
extern "C" int ufm_debug_attention_stamps(const uint16_t* qkv, uint16_t* out, int B, int N, int H, int waves, unsigned long long* diag, void* stream) {
    UFM_REQUIRE(qkv && out && diag && (waves == 2 || waves == 4), "ufm_debug_attention_stamps: bad arguments");
    const PwArgs a = pw_self_args(qkv, out, B, N, H, waves);
    dim3 grid(attn_pw_grid(a.nunits, waves)), block(waves * 64);
    if (waves == 4) hipLaunchKernelGGL((attn_pw_kernel<4, true>), grid, block, 0, (hipStream_t)stream, a, diag);
    else hipLaunchKernelGGL((attn_pw_kernel<2, true>), grid, block, 0, (hipStream_t)stream, a, diag);
    UFM_CHECK_LAUNCH("ufm_debug_attention_stamps");
    return UFM_OK;
}
