// Lab probe (round 6): does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` write ZEROS to LDS (like the register form returns 0)?
// The row-window halo convolution (ufm_amd/csrc/conv_bf16x3_halo.hip) relies on it for its zero padding.  Build: hipcc --offload-arch=gfx950 -O2 buffer_lds_oob.hip -o bin/buffer_lds_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const char* in, unsigned* out, int nbytes) {
    __shared__ __attribute__((aligned(16))) unsigned smem[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) smem[i] = 0xDEADBEEFu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, nbytes, 0x00020000);
    const unsigned voff = (lane % 3 == 1) ? 0xFFFFFFF0u : (lane % 3 == 2 ? (unsigned)nbytes - 8u : (unsigned)lane * 16u);  // OOB far, straddling the end, in range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + 256), 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = smem[i];
}
int main() {
    const int n = 4096;
    std::vector<unsigned> h(n / 4);
    for (int i = 0; i < n / 4; ++i) h[i] = 0x10000u + i;
    char* d; unsigned* o;
    hipMalloc(&d, n); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
    k<<<1, 64>>>(d, o, n);
    std::vector<unsigned> r(1024);
    hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        for (int j = 0; j < 4; ++j) {
            const unsigned got = r[256 + lane * 4 + j];
            unsigned want;
            if (lane % 3 == 1) want = 0;
            else if (lane % 3 == 2) want = 0xFFFFFFFFu;  // straddling: report only
            else want = 0x10000u + lane * 4 + j;
            if (want != 0xFFFFFFFFu && got != want) ++bad;
            if (lane < 6) printf("lane %d word %d: %08x%s\n", lane, j, got, want == 0xFFFFFFFFu ? " (straddles the end)" : "");
        }
    }
    printf("untouched before: %08x after: %08x\n", r[255], r[256 + 256]);
    printf(bad ? "OOB PROBE FAIL: %d words differ\n" : "OOB PROBE OK: out-of-range lanes wrote zeros, in-range lanes their data (%d bad)\n", bad);
    return bad != 0;
}
