// fp32 attention forward (numerics mode "parity"): same flash structure as attention_bf16.hip but
// every contraction runs on the exact-fp32 MFMA v_mfma_f32_32x32x2_f32, so the result tracks an
// fp32 CPU reference to ~1e-6.  Correctness path, single LDS stage; not the benchmarked kernel.
//   S^T[key][q] = K . Q^T : lane (key, h) reads K[key][8*st+4*h+e] (one ds_read_b128 per 4 MFMAs)
//   O^T[d][q]  += V^T . P^T: MFMA r of a 32-key tile takes the lane's accumulator register r as
//   the B operand (key (r&3)+8*(r>>2)+4*h on lane half h) and V[that key][d] as the A operand
//   (one conflict-free ds_read_b32 per MFMA).  P stays fp32.
#include "common.h"

namespace {

constexpr int QB = 128, KB = 64;
constexpr float NEG_BIG = -1.0e30f;

// Two-source form (as attention_bf16.hip): Q [B*Nq][ldq], K / V [B*N][ldkv], out [B*Nq][ldo].
__global__ __launch_bounds__(256) void attn_f32_kernel(const float* __restrict__ qp_, int ldq, const float* __restrict__ kp_,
                                                       const float* __restrict__ vp_, int ldkv, float* __restrict__ out, int ldo,
                                                       int Nq, int N, int H, float scale) {
    __shared__ __attribute__((aligned(16))) float sK[KB * 64];
    __shared__ __attribute__((aligned(16))) float sV[KB * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nqb = (Nq + QB - 1) / QB;
    const int lid = blockIdx.x;
    const int qblk = lid % nqb, head = (lid / nqb) % H, b = lid / (nqb * H);
    const int ld = ldkv;
    const float* base = qp_ + (size_t)b * Nq * ldq + head * 64;
    const float* kp = kp_ + (size_t)b * N * ldkv + head * 64;
    const float* vp = vp_ + (size_t)b * N * ldkv + head * 64;
    const int ql = lane & 31, hh = lane >> 5;
    const int q = qblk * QB + wave * 32 + ql;

    f32x4 qf[8];
    {
        const float* qr = base + (size_t)min(q, Nq - 1) * ldq + 4 * hh;
#pragma unroll
        for (int st = 0; st < 8; ++st) qf[st] = *(const f32x4*)(qr + 8 * st);
    }
    const int srow = tid >> 4, schunk = tid & 15;

    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;

    const int nt = (N + KB - 1) / KB;
    for (int t = 0; t < nt; ++t) {
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = srow + 16 * i;
            const size_t row = (size_t)min(t * KB + r, N - 1);
            const f32x4 kv = *(const f32x4*)(kp + row * ld + schunk * 4);
            const f32x4 vv = *(const f32x4*)(vp + row * ld + schunk * 4);
            *(f32x4*)(sK + r * 64 + ((schunk ^ (r & 15)) << 2)) = kv;
            *(f32x4*)(sV + r * 64 + schunk * 4) = vv;
        }
        __syncthreads();

        f32x16 st[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
            const int key = kt * 32 + ql;
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                const f32x4 kf = *(const f32x4*)(sK + key * 64 + (((2 * s8 + hh) ^ (key & 15)) << 2));
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    st[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[s8][e], st[kt], 0, 0, 0);
            }
        }
        float mloc = NEG_BIG;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * KB + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                st[kt][r] = key < N ? st[kt][r] * scale : NEG_BIG;
                mloc = fmaxf(mloc, st[kt][r]);
            }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = expf(m_run - m_new);
        m_run = m_new;
        float lsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[kt][r] = expf(st[kt][r] - m_new);
                lsum += st[kt][r];
            }
        l_run = l_run * alpha + lsum;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const float vf = sV[key * 64 + dt * 32 + ql];
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, st[kt][r], oacc[dt], 0, 0, 0);
                }
            }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (q < Nq) {
        float* orow = out + ((size_t)b * Nq + q) * ldo + head * 64 + 4 * hh;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv, oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv};
                *(f32x4*)(orow + dt * 32 + 8 * g) = v;
            }
    }
}

}  // namespace

extern "C" int ufm_attention_f32(const float* qkv, float* out, int B, int N, int H, float scale, void* stream) {
    UFM_REQUIRE(qkv && out, "ufm_attention_f32: null pointer");
    UFM_REQUIRE(B > 0 && N > 0 && H > 0 && (int64_t)((N + QB - 1) / QB) * H * B < (1ll << 31), "ufm_attention_f32: bad shape");
    UFM_REQUIRE(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0, "ufm_attention_f32: misaligned pointer");
    dim3 grid(((N + QB - 1) / QB) * H * B), block(256);
    hipLaunchKernelGGL(attn_f32_kernel, grid, block, 0, (hipStream_t)stream, qkv, 3 * H * 64, qkv + H * 64, qkv + 2 * H * 64, 3 * H * 64, out, H * 64, N, N, H, scale);
    UFM_CHECK_LAUNCH("ufm_attention_f32");
    return UFM_OK;
}

extern "C" int ufm_cross_attention_f32(const float* q, int ldq, const float* k, const float* v, int ldkv, float* out, int ldo, int B,
                                       int Nq, int Nk, int H, float scale, void* stream) {
    UFM_REQUIRE(q && k && v && out, "ufm_cross_attention_f32: null pointer");
    UFM_REQUIRE(B > 0 && Nq > 0 && Nk > 0 && H > 0 && scale > 0.0f && (int64_t)((Nq + QB - 1) / QB) * H * B < (1ll << 31), "ufm_cross_attention_f32: bad shape");
    UFM_REQUIRE(ldq >= H * 64 && ldkv >= H * 64 && ldo >= H * 64 && ldq % 4 == 0 && ldkv % 4 == 0 && ldo % 4 == 0, "ufm_cross_attention_f32: bad leading dimensions %d/%d/%d", ldq, ldkv, ldo);
    UFM_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)out % 16) == 0, "ufm_cross_attention_f32: misaligned pointer");
    dim3 grid(((Nq + QB - 1) / QB) * H * B), block(256);
    hipLaunchKernelGGL(attn_f32_kernel, grid, block, 0, (hipStream_t)stream, q, ldq, k, v, ldkv, out, ldo, Nq, Nk, H, scale);
    UFM_CHECK_LAUNCH("ufm_cross_attention_f32");
    return UFM_OK;
}
