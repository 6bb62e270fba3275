#!/usr/bin/env python3
"""8-phase GEMM: grouped-rasterization height GM (M-tiles per group) vs time, with and without the epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip
lib = hip.lib()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K, mode in ((10952, 3072, 1024, "bf16"), (10952, 4096, 1024, "gelu"), (10952, 1024, 1024, "res"), (10952, 1024, 4096, "res"), (10952, 2304, 768, "bf16"), (10952, 3072, 768, "gelu"), (10952, 768, 3072, "res")):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.randn(M, N, device="cuda") if mode == "res" else torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    lib.ufm_debug_set_gemm_variant(0)
    for flags in (0, 4):
        r = []
        for gm in (1, 2, 4, 6, 8, 11, 16, 43):
            lib.ufm_debug_set_gemm_flags(flags | (gm << 8))
            us = t(lambda: hip.gemm_bf16(A, W, M, N, K, out, bias=bias, act=hip.ACT_GELU if mode == "gelu" else hip.ACT_NONE, res=out if mode == "res" else None))
            r.append(f"GM={gm}: {us:6.1f}")
        print(f"M={M} N={N} K={K} {mode} flags={flags}: " + " | ".join(r), flush=True)
lib.ufm_debug_set_gemm_flags(0)
