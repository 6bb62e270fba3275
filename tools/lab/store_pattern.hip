// Store-pattern microbenchmark for the GEMM epilogue (gfx950): 256 workgroups x 8 waves (one per CU, like the 8-phase kernel)
// write a [M][N] bf16 (or fp32) matrix tile by tile, each wave instruction covering SEG-byte row segments:
//   bf16, SEG = 128  : the shipped epilogue (a wave owns a 64-column slice: 8 rows x 128 B per instruction)
//   bf16, SEG = 512  : 2 rows x 512 B per instruction (a wave-row of 4 waves pooled through LDS would allow this)
//   SEG = 1024       : 1 row x 1 KiB (upper bound: fully contiguous per instruction)
// Same bytes per block and the same 256x256 tile walk (tile = 256 rows x 256 columns).  Prints GB/s per pattern.
//   hipcc -O3 --offload-arch=gfx950 tools/lab/store_pattern.hip -o tools/lab/bin/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// ESZ = bytes per element (2 or 4); SEG = contiguous bytes per row segment of one wave instruction; RMW = read-add-write
template <int ESZ, int SEG, int RMW>
__global__ __launch_bounds__(512, 1) void k(char* out, int M, int N, int ntm, int ntn) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t ld = (size_t)N * ESZ;                 // row pitch in bytes
    constexpr int TROW = 256 * ESZ;                    // bytes of one tile row
    constexpr int LPS = SEG / 16;                      // lanes per segment
    constexpr int RPI = 64 / LPS;                      // rows (segments) per wave instruction
    constexpr int SPR = TROW / SEG;                    // segments per tile row
    for (int t = blockIdx.x; t < ntm * ntn; t += gridDim.x) {
        const int tm = t / ntn, tn = t % ntn;
        char* base = out + (size_t)tm * 256 * ld + (size_t)tn * TROW;
        // the tile's 256 rows x SPR segments are dealt to the 8 waves: instruction i of wave w covers segment-rows
        // (i * 8 + w) * RPI .. + RPI of the tile's (row, segment) list, row-major
        constexpr int NINST = 256 * SPR / RPI / 8;
#pragma unroll 4
        for (int i = 0; i < NINST; ++i) {
            const int sr = (i * 8 + wave) * RPI + lane / LPS;   // index into (row, segment)
            const int row = sr / SPR, seg = sr % SPR;
            if (tm * 256 + row < M) {
                u32x4* p = (u32x4*)(base + (size_t)row * ld + seg * SEG + (lane % LPS) * 16);
                u32x4 v = {(unsigned)t, (unsigned)i, (unsigned)lane, 7u};
                if (RMW) { u32x4 o = *p; v += o; }
                *p = v;
            }
        }
    }
}

template <int ESZ, int SEG, int RMW>
static double run(char* buf, int M, int N) {
    const int ntm = (M + 255) / 256, ntn = N / 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int r = 0; r < 12; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<ESZ, SEG, RMW>), dim3(256), dim3(512), 0, 0, buf, M, N, ntm, ntn);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double bytes = (double)M * N * ESZ * (RMW ? 2 : 1);
    return bytes / (ts[ts.size() / 2] * 1e-3) / 1e9;
}

int main() {
    const int M = 10952;
    char* buf;
    hipMalloc(&buf, (size_t)11008 * 4096 * 4);
    hipMemset(buf, 0, (size_t)11008 * 4096 * 4);
    for (int N : {3072, 4096, 1024}) {
        printf("M=%d N=%d bf16 store : seg128 %.0f GB/s | seg256 %.0f | seg512 %.0f | (%.1f us at seg128)\n", M, N, run<2, 128, 0>(buf, M, N), run<2, 256, 0>(buf, M, N),
               run<2, 512, 0>(buf, M, N), (double)M * N * 2 / run<2, 128, 0>(buf, M, N) / 1e3);
        printf("M=%d N=%d fp32 store : seg256 %.0f GB/s | seg512 %.0f | seg1024 %.0f\n", M, N, run<4, 256, 0>(buf, M, N), run<4, 512, 0>(buf, M, N), run<4, 1024, 0>(buf, M, N));
        printf("M=%d N=%d fp32 RMW   : seg256 %.0f GB/s | seg512 %.0f | seg1024 %.0f\n", M, N, run<4, 256, 1>(buf, M, N), run<4, 512, 1>(buf, M, N), run<4, 1024, 1>(buf, M, N));
    }
    return 0;
}
