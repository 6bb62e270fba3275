#!/usr/bin/env python3
"""Where does a one-pair GEMM (M = 2740: at most one 128x128 block per CU) spend its time?  Ablations of the 128x128 kernel:
ufm_debug_set_gemm_flags 2 = no DMA (operands stale in LDS), 4 = no epilogue traffic, 6 = neither (MFMA + LDS reads + barriers)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip
lib = hip.lib()
for name, M, N, K in (("proj", 2740, 1024, 1024), ("fc2", 2740, 1024, 4096), ("i_fc2", 2738, 768, 3072)):
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16()
    bias = torch.randn(N, device="cuda") * 0.1
    out = torch.randn(M, N, device="cuda")
    row = []
    for flags in (0, 4, 2, 6):
        lib.ufm_debug_set_gemm_flags(flags)
        f = lambda: hip.gemm_bf16(A, W, M, N, K, out, bias=bias, res=out)
        for _ in range(3): f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        row.append(f"flags {flags}: {sorted(ts)[3]:6.1f} us")
    lib.ufm_debug_set_gemm_flags(0)
    print(f"{name:6s} M={M} N={N} K={K}:  " + "   ".join(row), flush=True)
