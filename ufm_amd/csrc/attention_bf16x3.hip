// Flash-style multi-head attention forward for head_dim 64 on the SPLIT format (numerics mode "precise"):
// fp32-class accuracy on the bf16 matrix cores.  Q, K, V arrive as (hi, lo) bf16 planes (UFM_BF16X2, written by
// ufm_gemm_bf16x3's epilogue); every contraction is evaluated as hi*hi + hi*lo + lo*hi with fp32 accumulation:
//     S^T = Kh.Ql^T + Kl.Qh^T + Kh.Qh^T                      (3 x 8 v_mfma_f32_32x32x16_bf16 per 64-key tile)
//     P   = exp2(c*S - c*m)  in fp32;  Ph = bf16(P), Pl = bf16(P - Ph)   (split in registers)
//     O^T += Vl^T.Ph^T + Vh^T.Pl^T + Vh^T.Ph^T               (3 x 8 MFMA)
// The dropped lo*lo terms are 2^-16 relative, so a score carries ~2^-17 * sum|q||k| absolute error and the output
// ~2^-17 relative -- what the 1e-3 px end-to-end gate needs through 36 transformer layers (the single-pass bf16
// kernel's 2^-9 does not).  The softmax statistics (running max, row sum, rescale) are fp32 as everywhere else.
//
// Structure = attention_bf16.hip (workgroup = 4 waves x 32 query rows, 64-key tiles, swapped QK^T so that one lane
// owns one query column, P never leaves registers, V^T by ds_read_b64_tr_b16, register-staged K/V with the next
// tile's loads issued before this tile's MFMAs), with both planes of K and V in LDS (32 KiB per stage, two stages,
// two workgroups per CU).  With three MFMAs per fragment pair the loop is matrix-core bound: 48 MFMAs (1536 cycles)
// per wave and tile against ~1100 cycles of softmax/split VALU issue, and the second wave of each SIMD (the other
// co-resident workgroup) fills the gaps.  The O rescale is skipped while no row's maximum moved (wave-uniform test).
#include "common.h"

namespace {

constexpr int QB = 128;  // query rows per workgroup
constexpr int KB = 64;   // keys per tile
constexpr int PLANE = 8192;        // one 64-key x 64-d bf16 tile
constexpr int STAGE = 4 * PLANE;   // [K hi][K lo][V hi][V lo]
constexpr float NEG_BIG = -1.0e30f;

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

__device__ __forceinline__ bf16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_ptr)LDS_PTR(p));
}

// Two-source form (as attention_bf16.hip): Q [2][B*Nq][ldq] (lo plane at +q_plane), K / V [2][B*N][ldkv] (+kv_plane),
// out [2][B*Nq][ldo] (+out_plane).
__global__ __launch_bounds__(256, 2) void attn_x3_kernel(const uint16_t* __restrict__ qp_, int ldq, long long q_plane,
                                                         const uint16_t* __restrict__ kp_, const uint16_t* __restrict__ vp_, int ldkv, long long in_plane,
                                                         uint16_t* __restrict__ out, int ldo, long long out_plane, int Nq, int N, int H, float c) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nqb = (Nq + QB - 1) / QB;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);  // all query blocks of one (image, head) share an XCD's L2
    const int qblk = lid % nqb, head = (lid / nqb) % H, b = lid / (nqb * H);
    const int ld = ldkv;
    const uint16_t* base = qp_ + (size_t)b * Nq * ldq + head * 64;
    const uint16_t* kp = kp_ + (size_t)b * N * ldkv + head * 64;
    const uint16_t* vp = vp_ + (size_t)b * N * ldkv + head * 64;
    const int ql = lane & 31, hh = lane >> 5;
    const int q = qblk * QB + wave * 32 + ql;

    bf16x8 qh[4], qlo[4];
    {
        const uint16_t* qr = base + (size_t)min(q, Nq - 1) * ldq + 8 * hh;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qh[s] = *(const bf16x8*)(qr + 16 * s);
            qlo[s] = *(const bf16x8*)(qr + q_plane + 16 * s);
        }
    }

    // staging: thread -> rows (tid>>3) and (tid>>3)+32 of the tile, 16-byte chunk tid&7, both planes of K and V
    const int srow = tid >> 3, schunk = tid & 7;
    int k_lds[2], v_lds[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = srow + 32 * i;
        k_lds[i] = r * 128 + ((schunk ^ ((r >> 1) & 7)) << 4);
        v_lds[i] = 2 * PLANE + r * 128 + ((schunk ^ (((r >> 1) & 1) << 2)) << 4);
    }
    u32x4 kreg[2][2], vreg[2][2];  // [row][plane]
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const size_t row = (size_t)min(t * KB + srow + 32 * i, N - 1);  // clamp: masked below
            const uint16_t* kr = kp + row * ld + schunk * 8;
            const uint16_t* vr = vp + row * ld + schunk * 8;
            kreg[i][0] = *(const u32x4*)kr;
            kreg[i][1] = *(const u32x4*)(kr + in_plane);
            vreg[i][0] = *(const u32x4*)vr;
            vreg[i][1] = *(const u32x4*)(vr + in_plane);
        }
    };
    auto store_tile = [&](int stage) {
        char* s = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(u32x4*)(s + k_lds[i]) = kreg[i][0];
            *(u32x4*)(s + PLANE + k_lds[i]) = kreg[i][1];
            *(u32x4*)(s + v_lds[i]) = vreg[i][0];
            *(u32x4*)(s + PLANE + v_lds[i]) = vreg[i][1];
        }
    };

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
    int v_off[2][2][2][2];                            // [dt][kt][s2][lo/hi half of the fragment]
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
                    v_off[dt][kt][s2][e] = 2 * PLANE + row * 128 + ((((dcol >> 3)) ^ (((row >> 1) & 1) << 2)) << 4) + ((dcol & 7) << 1);
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

        // ---- S^T = K . Q^T, small terms first ----
        f32x16 st[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 kfh = *(const bf16x8*)(s + k_off[kt][ks]);
                const bf16x8 kfl = *(const bf16x8*)(s + PLANE + k_off[kt][ks]);
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfl, qh[ks], st[kt], 0, 0, 0);
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, qlo[ks], st[kt], 0, 0, 0);
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfh, qh[ks], st[kt], 0, 0, 0);
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
        const bool moved = m_new != m_run;
        m_run = m_new;
        const float mc = m_new * c;
        float lsum = 0.f;
        bf16x8 ph[2][2], pl[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            unsigned pkh[8], pkl[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][r], c, -mc));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kt][r + 1], c, -mc));
                lsum += p0 + p1;
                const unsigned h = pack_bf16x2(p0, p1);
                pkh[r >> 1] = h;
                pkl[r >> 1] = pack_bf16x2(p0 - __uint_as_float(h << 16), p1 - __uint_as_float(h & 0xffff0000u));
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                u32x4 wh = {pkh[4 * s2], pkh[4 * s2 + 1], pkh[4 * s2 + 2], pkh[4 * s2 + 3]};
                u32x4 wl = {pkl[4 * s2], pkl[4 * s2 + 1], pkl[4 * s2 + 2], pkl[4 * s2 + 3]};
                ph[kt][s2] = __builtin_bit_cast(bf16x8, wh);
                pl[kt][s2] = __builtin_bit_cast(bf16x8, wl);
            }
        }
        l_run = l_run * alpha + lsum;
        if (__any(moved)) {  // wave-uniform: after the first tiles the running maxima rarely move
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
        }

        // ---- O^T += V^T . P^T, small terms first ----
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x4 a0 = tr_read(s + v_off[dt][kt][s2][0]);
                    const bf16x4 a1 = tr_read(s + v_off[dt][kt][s2][1]);
                    const bf16x4 b0 = tr_read(s + PLANE + v_off[dt][kt][s2][0]);
                    const bf16x4 b1 = tr_read(s + PLANE + v_off[dt][kt][s2][1]);
                    const bf16x8 vh = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    const bf16x8 vl = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph[kt][s2], oacc[dt], 0, 0, 0);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl[kt][s2], oacc[dt], 0, 0, 0);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph[kt][s2], oacc[dt], 0, 0, 0);
                }

        if (t + 1 < nt) store_tile((t + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: O[q][d] = O^T[d][q] / l, stored as (hi, lo) planes ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q < Nq) {
        uint16_t* orow = out + ((size_t)b * Nq + q) * ldo + head * 64 + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4], hi[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = oacc[dt][4 * g + j] * inv;
                const unsigned h0 = pack_bf16x2(v[0], v[1]), h1 = pack_bf16x2(v[2], v[3]);
                hi[0] = __uint_as_float(h0 << 16), hi[1] = __uint_as_float(h0 & 0xffff0000u);
                hi[2] = __uint_as_float(h1 << 16), hi[3] = __uint_as_float(h1 & 0xffff0000u);
                u32x2 pkh = {h0, h1};
                u32x2 pkl = {pack_bf16x2(v[0] - hi[0], v[1] - hi[1]), pack_bf16x2(v[2] - hi[2], v[3] - hi[3])};
                *(u32x2*)(orow + dt * 32 + 8 * g) = pkh;
                *(u32x2*)(orow + out_plane + dt * 32 + 8 * g) = pkl;
            }
    }
}

}  // namespace

// attention_bf16x3_pw.hip: the round-5 kernel (one wave per SIMD, two q-blocks per wave, LDS-DMA rings); bitwise this file's kernel
int ufm_launch_attn_x3_pw(const uint16_t* q, int ldq, long long q_plane, const uint16_t* k, const uint16_t* v, int ldkv, long long in_plane,
                          uint16_t* out, int ldo, long long out_plane, int B, int Nq, int Nk, int H, float c, hipStream_t stream, int waves, int out_il = 0, int fixref = 0);
int ufm_attn_x3_use_old();  // attention_bf16.hip: ufm_debug_set_attn_variant bit 1
int ufm_attn_x3_waves();
int ufm_attn_x3_fixref();  // attention_bf16.hip: ufm_debug_set_attn_variant bit 3 clear    // bit 2: eight waves per workgroup instead of four

extern "C" int ufm_attention_bf16x3(const uint16_t* qkv, uint16_t* out, int B, int N, int H, float scale, void* stream) {
    UFM_REQUIRE(qkv && out, "ufm_attention_bf16x3: null pointer");
    UFM_REQUIRE(B > 0 && N > 0 && H > 0 && (int64_t)((N + QB - 1) / QB) * H * B < (1ll << 31), "ufm_attention_bf16x3: bad shape B=%d N=%d H=%d", B, N, H);
    UFM_REQUIRE(scale > 0.0f, "ufm_attention_bf16x3: scale must be positive");
    UFM_REQUIRE(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 8) == 0, "ufm_attention_bf16x3: misaligned pointer");
    const long long rows = (long long)B * N;
    dim3 grid(((N + QB - 1) / QB) * H * B), block(256);
    // the LDS-DMA kernel addresses one batch item's K / V rows with 32-bit byte offsets and stages 16-byte chunks
    const bool pw_ok = !ufm_attn_x3_use_old() && (long long)N * 3 * H * 64 * 2 < (1ll << 31);
    if (pw_ok)
        ufm_launch_attn_x3_pw(qkv, 3 * H * 64, rows * 3 * H * 64, qkv + H * 64, qkv + 2 * H * 64, 3 * H * 64, rows * 3 * H * 64, out, H * 64, rows * H * 64, B, N, N, H,
                              scale * 1.44269504088896340736f, (hipStream_t)stream, ufm_attn_x3_waves(), 0, ufm_attn_x3_fixref());
    else
    hipLaunchKernelGGL(attn_x3_kernel, grid, block, 0, (hipStream_t)stream, qkv, 3 * H * 64, rows * 3 * H * 64, qkv + H * 64, qkv + 2 * H * 64, 3 * H * 64,
                       rows * 3 * H * 64, out, H * 64, rows * H * 64, N, N, H, scale * 1.44269504088896340736f);
    UFM_CHECK_LAUNCH("ufm_attention_bf16x3");
    return UFM_OK;
}

// ufm_attention_bf16x3 with the output stored INTERLEAVED ([B N][H 64 / 32][hi 32 | lo 32], UFM_BF16X2_IL): the A operand of ufm_gemm_bf16x3_il
// (the proj Linear of numerics "precise").  Same kernel, same values as ufm_attention_bf16x3, other store addresses.
extern "C" int ufm_attention_bf16x3_il(const uint16_t* qkv, uint16_t* out, int B, int N, int H, float scale, void* stream) {
    UFM_REQUIRE(qkv && out, "ufm_attention_bf16x3_il: null pointer");
    UFM_REQUIRE(B > 0 && N > 0 && H > 0 && (int64_t)((N + QB - 1) / QB) * H * B < (1ll << 31), "ufm_attention_bf16x3_il: bad shape B=%d N=%d H=%d", B, N, H);
    UFM_REQUIRE(scale >= 0.0f, "ufm_attention_bf16x3_il: scale must be positive, or 0 for q pre-scaled by softmax_scale * log2(e)");
    UFM_REQUIRE(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 8) == 0, "ufm_attention_bf16x3_il: misaligned pointer");
    UFM_REQUIRE(!ufm_attn_x3_use_old() && (long long)N * 3 * H * 64 * 2 < (1ll << 31), "ufm_attention_bf16x3_il: the interleaved output exists in the LDS-DMA kernel only (variant bit 1 clear, N * 3 H * 128 B < 2 GiB)");
    UFM_REQUIRE(scale > 0.0f || ufm_attn_x3_fixref(), "ufm_attention_bf16x3_il: scale == 0 (pre-scaled q) needs the fixed-reference kernel (variant bits 2 and 3 clear)");
    const long long rows = (long long)B * N;
    ufm_launch_attn_x3_pw(qkv, 3 * H * 64, rows * 3 * H * 64, qkv + H * 64, qkv + 2 * H * 64, 3 * H * 64, rows * 3 * H * 64, out, H * 64, rows * H * 64, B, N, N, H,
                          scale * 1.44269504088896340736f, (hipStream_t)stream, ufm_attn_x3_waves(), 1, ufm_attn_x3_fixref());
    UFM_CHECK_LAUNCH("ufm_attention_bf16x3_il");
    return UFM_OK;
}

extern "C" int ufm_cross_attention_bf16x3(const uint16_t* q, int ldq, const uint16_t* k, const uint16_t* v, int ldkv, uint16_t* out, int ldo,
                                          int B, int Nq, int Nk, int H, float scale, void* stream) {
    UFM_REQUIRE(q && k && v && out, "ufm_cross_attention_bf16x3: null pointer");
    UFM_REQUIRE(B > 0 && Nq > 0 && Nk > 0 && H > 0 && scale > 0.0f && (int64_t)((Nq + QB - 1) / QB) * H * B < (1ll << 31), "ufm_cross_attention_bf16x3: bad shape");
    UFM_REQUIRE(ldq >= H * 64 && ldkv >= H * 64 && ldo >= H * 64 && ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 4 == 0, "ufm_cross_attention_bf16x3: bad leading dimensions %d/%d/%d", ldq, ldkv, ldo);
    UFM_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)out % 8) == 0, "ufm_cross_attention_bf16x3: misaligned pointer");
    dim3 grid(((Nq + QB - 1) / QB) * H * B), block(256);
    if (!ufm_attn_x3_use_old() && (long long)Nk * ldkv * 2 < (1ll << 31))
        ufm_launch_attn_x3_pw(q, ldq, (long long)B * Nq * ldq, k, v, ldkv, (long long)B * Nk * ldkv, out, ldo, (long long)B * Nq * ldo, B, Nq, Nk, H,
                              scale * 1.44269504088896340736f, (hipStream_t)stream, ufm_attn_x3_waves(), 0, ufm_attn_x3_fixref());
    else
    hipLaunchKernelGGL(attn_x3_kernel, grid, block, 0, (hipStream_t)stream, q, ldq, (long long)B * Nq * ldq, k, v, ldkv, (long long)B * Nk * ldkv, out, ldo,
                       (long long)B * Nq * ldo, Nq, Nk, H, scale * 1.44269504088896340736f);
    UFM_CHECK_LAUNCH("ufm_cross_attention_bf16x3");
    return UFM_OK;
}
