// Flash-style multi-head attention forward for head_dim 64, bf16 operands, fp32 accumulate.
//
// gfx950 design (cdna_hip_programming.md Appendix B "Fused attention prefill", section 3
// "An accumulator tile as the next MFMA's operand", T2, T10):
//   * workgroup = 4 waves = 128 query rows of one (image, head); each wave owns 32 query rows
//     and streams all keys in 64-key tiles shared through LDS (K and V: 8 KiB each, 2 stages).
//   * swapped QK^T: S^T[key][q] = K . Q^T with v_mfma_f32_32x32x16_bf16, so a lane holds ONE
//     query column and 16 keys per 32-key tile: the online-softmax max/sum are in-lane plus a
//     single lane^32 exchange; the rescale factor is a per-lane scalar.
//   * P never leaves registers: the S^T accumulator, converted pairwise to bf16, IS the B operand
//     of O^T[d][q] += V^T[d][key] . P^T[key][q]; V^T fragments come from a row-major V tile via
//     ds_read_b64_tr_b16 (hardware transpose) in the accumulator's permuted key order.
//   * K tile: 128-B rows, chunk XOR ((key>>1)&7) -> conflict-free ds_read_b128 for the 32-row
//     operand; V tile: chunk XOR (((key>>1)&1)<<2) -> conflict-free transposed reads.
//   * K/V: global -> registers -> LDS (register staging, T14): next tile's loads are issued
//     before this tile's MFMAs, written to the other stage afterwards; one barrier per tile.
//   * exp2 with scale*log2(e) folded into one FMA per score; ragged tail masked in the last tile.
#include <type_traits>

#include "common.h"

namespace {

constexpr int QB = 128;  // query rows per workgroup
constexpr int KB = 64;   // keys per tile
constexpr int STAGE = 16384;
constexpr float NEG_BIG = -1.0e30f;

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

__device__ __forceinline__ bf16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_ptr)LDS_PTR(p));
}

// Two-source form: queries [B*Nq][ldq], keys / values [B*Nk][ldkv] (self-attention = the three column blocks of one QKV
// buffer with Nq == Nk; cross-attention = Q of one view against K, V of the other), output [B*Nq][ldo].
__global__ __launch_bounds__(256, 2) void attn_bf16_kernel(const uint16_t* __restrict__ qp_, int ldq, const uint16_t* __restrict__ kp_,
                                                           const uint16_t* __restrict__ vp_, int ldkv, uint16_t* __restrict__ out, int ldo,
                                                           int Nq, int N, int H, float c) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1-D grid, XCD-aware: all query blocks of one (image, head) are consecutive logical ids and
    // therefore share one XCD's L2 for their K/V stream (T1).
    const int nqb = (Nq + QB - 1) / QB;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int qblk = lid % nqb, head = (lid / nqb) % H, b = lid / (nqb * H);
    const int ld = ldkv;
    const uint16_t* base = qp_ + (size_t)b * Nq * ldq + head * 64;
    const uint16_t* kp = kp_ + (size_t)b * N * ldkv + head * 64;
    const uint16_t* vp = vp_ + (size_t)b * N * ldkv + head * 64;
    const int ql = lane & 31, hh = lane >> 5;
    const int q = qblk * QB + wave * 32 + ql;

    bf16x8 qf[4];
    {
        const uint16_t* qr = base + (size_t)min(q, Nq - 1) * ldq + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *(const bf16x8*)(qr + 16 * s);
    }

    // staging: thread -> rows (tid>>3) and (tid>>3)+32 of the tile, 16-byte chunk tid&7
    const int srow = tid >> 3, schunk = tid & 7;
    int k_lds[2], v_lds[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = srow + 32 * i;
        k_lds[i] = r * 128 + ((schunk ^ ((r >> 1) & 7)) << 4);
        v_lds[i] = 8192 + r * 128 + ((schunk ^ (((r >> 1) & 1) << 2)) << 4);
    }
    u32x4 kreg[2], vreg[2];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const size_t row = (size_t)min(t * KB + srow + 32 * i, N - 1);  // clamp: masked below
            kreg[i] = *(const u32x4*)(kp + row * ld + schunk * 8);
            vreg[i] = *(const u32x4*)(vp + row * ld + schunk * 8);
        }
    };
    auto store_tile = [&](int stage) {
        char* s = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(u32x4*)(s + k_lds[i]) = kreg[i];
            *(u32x4*)(s + v_lds[i]) = vreg[i];
        }
    };

    // fragment read offsets
    int k_off[2][4];  // [key tile][k-step]
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int key = kt * 32 + ql;
            k_off[kt][s] = key * 128 + (((2 * s + hh) ^ ((key >> 1) & 7)) << 4);
        }
    // V^T transposed-read lane address: group G = lane>>4, i = lane&15: row kb + (i>>2), cols d0 + 4*(i&3)
    const int ti = lane & 15, tq = ti >> 2, tp = ti & 3;
    const int tdc = 16 * ((lane >> 4) & 1) + 4 * tp;  // d column within the 32-wide d tile
    int v_off[2][2][2][2];                            // [dt][kt][s2][lo/hi]
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
                    v_off[dt][kt][s2][e] = 8192 + row * 128 + ((((dcol >> 3)) ^ (((row >> 1) & 1) << 2)) << 4) + ((dcol & 7) << 1);
                }

    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;

    const int nt = (N + KB - 1) / KB;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) load_tile(t + 1);
        const char* s = smem + (t & 1) * STAGE;

        // ---- S^T = K . Q^T ----
        f32x16 st[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kf = *(const bf16x8*)(s + k_off[kt][ks]);
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], st[kt], 0, 0, 0);
            }
        }
        if ((t + 1) * KB > N) {  // ragged tail (block-uniform branch)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * KB + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (key >= N) st[kt][r] = NEG_BIG;
                }
        }
        // ---- online softmax (per lane = per query column) ----
        float mloc = st[0][0];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[kt][r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        m_run = m_new;
        const float mc = m_new * c;
        float lsum = 0.f;
        bf16x8 pf[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            unsigned pk[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][r], c, -mc));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][r + 1], c, -mc));
                lsum += p0 + p1;
                pk[r >> 1] = pack_bf16x2(p0, p1);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4 w = {pk[4 * s2], pk[4 * s2 + 1], pk[4 * s2 + 2], pk[4 * s2 + 3]};
                pf[kt][s2] = __builtin_bit_cast(bf16x8, w);
            }
        }
        l_run = l_run * alpha + lsum;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;

        // ---- O^T += V^T . P^T ----
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x4 lo = tr_read(s + v_off[dt][kt][s2][0]);
                    const bf16x4 hi = tr_read(s + v_off[dt][kt][s2][1]);
                    const bf16x8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kt][s2], oacc[dt], 0, 0, 0);
                }

        if (t + 1 < nt) store_tile((t + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: O[q][d] = O^T[d][q] / l ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q < Nq) {
        uint16_t* orow = out + ((size_t)b * Nq + q) * ldo + head * 64 + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                u32x2 pk = {pack_bf16x2(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv),
                            pack_bf16x2(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv)};
                *(u32x2*)(orow + dt * 32 + 8 * g) = pk;
            }
    }
}


}  // namespace

int ufm_launch_attn_pw(const uint16_t* qkv, uint16_t* out, int B, int N, int H, int variant, hipStream_t stream);  // attention_bf16_pw.hip
int ufm_launch_attn_pw2(const uint16_t* q, int ldq, int q_bs, const uint16_t* k, const uint16_t* v, int ldkv, int kv_bs, uint16_t* out, int ldo, int o_bs,
                        int B, int Nq, int Nk, int H, int variant, hipStream_t stream);  // attention_bf16_pw.hip

static int g_attn_debug = 0;   // bit 0 of the variant: the bf16 LDS-DMA kernel with 2 waves per workgroup
static int g_attn_x3_old = 0;  // bit 1: ufm_attention_bf16x3 / ufm_cross_attention_bf16x3 on the round-1 kernel (A/B, bitwise reference)
static int g_attn_x3_waves = 4;  // bit 2: the round-5 split-precision kernel with 8 waves per workgroup (A/B)
static int g_attn_x3_fixref = 1; // bit 3 SET: the split-precision LDS-DMA kernel with the per-tile running maximum (bitwise the round-1 kernel) instead of round 6's fixed softmax reference
int ufm_attn_x3_use_old() { return g_attn_x3_old; }
int ufm_attn_x3_waves() { return g_attn_x3_waves; }
int ufm_attn_x3_fixref() { return g_attn_x3_fixref && g_attn_x3_waves == 4; }
extern "C" int ufm_debug_set_attn_variant(int v) {  // bit 0: 0 = 4 waves per workgroup (default), 1 = 2 waves; bit 1: the round-1 split-precision kernel
    if (v < 0 || v > 15) {
        ufm_set_error("ufm_debug_set_attn_variant: %d is not in 0..15 (bit 0: 2-wave bf16 kernel, bit 1: round-1 bf16x3 kernel, bit 2: 8-wave bf16x3 kernel, bit 3: bf16x3 LDS-DMA kernel with the running maximum)", v);
        return UFM_ERR_ARG;
    }
    g_attn_x3_fixref = (v & 8) ? 0 : 1;
    g_attn_debug = v & 1;
    g_attn_x3_old = (v >> 1) & 1;
    g_attn_x3_waves = (v & 4) ? 8 : 4;
    return UFM_OK;
}

extern "C" int ufm_attention_bf16(const uint16_t* qkv, uint16_t* out, int B, int N, int H, float scale, void* stream) {
    UFM_REQUIRE(qkv && out, "ufm_attention_bf16: null pointer");
    UFM_REQUIRE(B > 0 && N > 0 && H > 0 && (int64_t)((N + QB - 1) / QB) * H * B < (1ll << 31), "ufm_attention_bf16: bad shape B=%d N=%d H=%d", B, N, H);
    UFM_REQUIRE(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 8) == 0, "ufm_attention_bf16: misaligned pointer");
    dim3 grid(((N + QB - 1) / QB) * H * B), block(256);
    if (scale == 0.0f)  // Q pre-scaled by softmax_scale*log2(e): 64 rows per wave, one wave per SIMD (attention_bf16_pw.hip)
        ufm_launch_attn_pw(qkv, out, B, N, H, g_attn_debug, (hipStream_t)stream);
    else
        hipLaunchKernelGGL(attn_bf16_kernel, grid, block, 0, (hipStream_t)stream, qkv, 3 * H * 64, qkv + H * 64, qkv + 2 * H * 64, 3 * H * 64, out, H * 64, N, N, H,
                           scale * 1.44269504088896340736f);
    UFM_CHECK_LAUNCH("ufm_attention_bf16");
    return UFM_OK;
}

extern "C" int ufm_cross_attention_bf16(const uint16_t* q, int ldq, const uint16_t* k, const uint16_t* v, int ldkv, uint16_t* out, int ldo,
                                        int B, int Nq, int Nk, int H, float scale, void* stream) {
    return ufm_attention_bf16_strided(q, ldq, Nq, k, v, ldkv, Nk, out, ldo, Nq, B, Nq, Nk, H, scale, stream);
}

// The general two-source form: batch item b's queries start at row b * q_batch_rows of `q`, its keys / values at row
// b * kv_batch_rows of `k` / `v`, its output at row b * out_batch_rows of `out` (rows of ldq / ldkv / ldo elements).
// scale == 0: Q is pre-scaled by softmax_scale * log2(e) (the projection's epilogue) -> the persistent LDS-DMA kernel of
// attention_bf16_pw.hip; scale > 0: the register-staged kernel above applies it.
extern "C" int ufm_attention_bf16_strided(const uint16_t* q, int ldq, int q_batch_rows, const uint16_t* k, const uint16_t* v, int ldkv, int kv_batch_rows,
                                          uint16_t* out, int ldo, int out_batch_rows, int B, int Nq, int Nk, int H, float scale, void* stream) {
    UFM_REQUIRE(q && k && v && out, "ufm_attention_bf16_strided: null pointer");
    UFM_REQUIRE(B > 0 && Nq > 0 && Nk > 0 && H > 0 && scale >= 0.0f && (int64_t)((Nq + QB - 1) / QB) * H * B < (1ll << 31), "ufm_attention_bf16_strided: bad shape B=%d Nq=%d Nk=%d H=%d", B, Nq, Nk, H);
    // the scale > 0 kernel (unchanged since round 1) stores 8-byte pieces: it keeps its original, looser output contract (ldo % 4, 8-byte
    // aligned out); the LDS-DMA kernel of the scale == 0 form stores whole 16-byte chunks
    const int ldo_mult = scale == 0.0f ? 8 : 4, out_align = scale == 0.0f ? 16 : 8;
    UFM_REQUIRE(ldq >= H * 64 && ldkv >= H * 64 && ldo >= H * 64 && ldq % 8 == 0 && ldkv % 8 == 0 && ldo % ldo_mult == 0, "ufm_attention_bf16_strided: bad leading dimensions %d/%d/%d", ldq, ldkv, ldo);
    UFM_REQUIRE(q_batch_rows >= Nq && kv_batch_rows >= Nk && out_batch_rows >= Nq, "ufm_attention_bf16_strided: batch strides %d/%d/%d rows are shorter than Nq=%d / Nk=%d", q_batch_rows, kv_batch_rows, out_batch_rows, Nq, Nk);
    UFM_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)out % out_align) == 0, "ufm_attention_bf16_strided: misaligned pointer");
    UFM_REQUIRE((int64_t)kv_batch_rows * ldkv * 2 < (1ll << 31), "ufm_attention_bf16_strided: one batch item's K/V rows exceed the 32-bit DMA offset");
    if (scale == 0.0f) {
        ufm_launch_attn_pw2(q, ldq, q_batch_rows, k, v, ldkv, kv_batch_rows, out, ldo, out_batch_rows, B, Nq, Nk, H, g_attn_debug, (hipStream_t)stream);
    } else {
        UFM_REQUIRE(q_batch_rows == Nq && kv_batch_rows == Nk && out_batch_rows == Nq, "ufm_attention_bf16_strided: the scale > 0 kernel takes densely packed batch items only");
        dim3 grid(((Nq + QB - 1) / QB) * H * B), block(256);
        hipLaunchKernelGGL(attn_bf16_kernel, grid, block, 0, (hipStream_t)stream, q, ldq, k, v, ldkv, out, ldo, Nq, Nk, H, scale * 1.44269504088896340736f);
    }
    UFM_CHECK_LAUNCH("ufm_attention_bf16_strided");
    return UFM_OK;
}
