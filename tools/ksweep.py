#!/usr/bin/env python3
"""GEMM time vs K at fixed (M, N): separates the K-independent cost (launch, prologue, epilogue) from the main-loop rate."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ufm_amd import hip
from tools.kbench import timeit
lib = hip.lib()
M = 21920
for N, resid in ((3072, False), (1024, True)):
    for variant, vname in ((1, "128x128 LDS-staged epilogue"), (28, "128x128 direct epilogue"), (24, "128x128 no-epilogue")):
        lib.ufm_debug_set_gemm_variant(variant)
        row = []
        for K in (1024, 4096):
            A = torch.randn(M, K, device="cuda").bfloat16()
            W = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16()
            out = torch.randn(M, N, device="cuda") if resid else torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            med, mn = timeit(lambda: hip.gemm_bf16(A, W, M, N, K, out, res=out if resid else None))
            row.append((K, round(med * 1e3, 1), round(2.0 * M * N * K / med / 1e9)))
        print(f"N={N} {'f32 RMW' if resid else 'bf16 out'} {vname}: " + "  ".join(f"K={k}: {us}us {tf}TF" for k, us, tf in row), flush=True)
lib.ufm_debug_set_gemm_variant(0)
# empty-kernel launch floor
x = torch.zeros(1024, device="cuda")
med, _ = timeit(lambda: hip.add_f32(x, x, x))
print("tiny kernel launch+run:", round(med * 1e3, 1), "us")
