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
};

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

    // ---- epilogue: lane holds out[m][nb..nb+3] for each (n_rep, m_rep) ----
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int row = m0 + wr * 64 + m * 16 + fr;
        if (row >= p.M) continue;
        const int rrow = (p.res_row_mod > 0) ? (row % p.res_row_mod) : row;
        const int orow = (p.out_row_group > 0)
                             ? (row / p.out_row_group) * (p.out_row_group + 1) + 1 + row % p.out_row_group
                             : row;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int nb = n0 + wc * 64 + n * 16 + fq * 4;
            f32x4 v = acc[n][m];
            if (p.bias) {
                const f32x4 bv = *(const f32x4*)(p.bias + nb);
                v += bv;
            }
            if (p.act != UFM_ACT_NONE) {
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

}  // namespace

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
    GemmArgs p{A, W, bias, gamma, res, out, lda, ldw, M, N, K, act, ldres, res_row_mod, ldo, out_row_group};
    const int ntm = (M + BM - 1) / BM, ntn = N / BN;
    dim3 grid(ntm * ntn), block(256);
    if (out_dtype == UFM_BF16)
        hipLaunchKernelGGL(gemm_bf16_kernel<1>, grid, block, 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(gemm_bf16_kernel<0>, grid, block, 0, (hipStream_t)stream, p);
    UFM_CHECK_LAUNCH("ufm_gemm_bf16");
    return UFM_OK;
}
