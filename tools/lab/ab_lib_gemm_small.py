#!/usr/bin/env python3
"""Same-box A/B of ufm_gemm_bf16 at the one-pair shapes (M = 2740 / 2738: at most one 128x128 block per CU) between the in-tree
library and another build (tools/lab/bin/libufm_hip_old.so): interleaved rounds in one process, random data, bitwise comparison."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
libs = {"new": C.CDLL(os.path.join(ROOT, "ufm_amd", "libufm_hip.so")), "old": C.CDLL(os.path.join(ROOT, "tools", "lab", "bin", "libufm_hip_old.so"))}
vp, i = C.c_void_p, C.c_int
for l in libs.values():
    l.ufm_gemm_bf16.argtypes = [vp, i, vp, i, i, i, i, vp, i, vp, vp, i, i, vp, i, i, i, vp]
    l.ufm_gemm_bf16.restype = i
st = torch.cuda.current_stream().cuda_stream
for name, M, N, K, res in (("proj", 2740, 1024, 1024, True), ("fc2", 2740, 1024, 4096, True), ("i_proj", 2738, 768, 768, True), ("i_fc2", 2738, 768, 3072, True),
                           ("qkv", 2740, 3072, 1024, False), ("fc1", 2740, 4096, 1024, False), ("proj B2", 5480, 1024, 1024, True), ("fc2 B2", 5480, 1024, 4096, True)):
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16()
    bias = torch.randn(N, device="cuda") * 0.1
    gamma = 1 + 0.1 * torch.randn(N, device="cuda")
    x0 = torch.randn(M, N, device="cuda")
    outs = {k: (x0.clone() if res else torch.empty(M, N, device="cuda", dtype=torch.bfloat16)) for k in libs}
    def run(k):
        o = outs[k]
        rc = libs[k].ufm_gemm_bf16(A.data_ptr(), K, W.data_ptr(), K, M, N, K, bias.data_ptr(), 0, gamma.data_ptr(), o.data_ptr() if res else None, N, 0, o.data_ptr(), 0 if res else 1, N, 0, st)
        assert rc == 0
    times = {k: [] for k in libs}
    for rnd in range(9):
        for k in libs:
            run(k); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): run(k)
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 10 * 1e3)
    for k in libs:  # bitwise: one call each from the same start
        outs[k].copy_(x0) if res else None
        run(k)
    torch.cuda.synchronize()
    msg = f"{name:8s} M={M} N={N} K={K}: "
    for k in libs:
        t = sorted(times[k]); msg += f" {k} {t[len(t)//2]:6.1f} us (min {t[0]:6.1f})"
    print(msg, " bitwise equal:", torch.equal(outs["new"], outs["old"]), flush=True)
