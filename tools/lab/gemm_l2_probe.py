#!/usr/bin/env python3
"""Is the 8-phase GEMM main loop limited by operand misses?  Same launch with lda = 0 / ldw = 0 (every tile reads the
same 256 rows: everything hits L2) against the real strides, with and without the epilogue (flags 4)."""
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
for M, N, K, mode in ((10952, 3072, 1024, "bf16"), (10952, 4096, 1024, "gelu"), (10952, 1024, 1024, "res"), (10952, 1024, 4096, "res"), (21904, 3072, 1024, "bf16")):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    out = torch.randn(M, N, device="cuda") if mode == "res" else torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for variant in (4, 0):
        lib.ufm_debug_set_gemm_variant(variant)
        for flags in (0, 4):
            r = []
            for lda, ldw in ((K, K), (0, K), (K, 0), (0, 0)):
                lib.ufm_debug_set_gemm_flags(flags | (0 if lda else 16) | (0 if ldw else 32))
                us = t(lambda: hip.gemm_bf16(A, W, M, N, K, out, bias=bias, act=hip.ACT_GELU if mode == "gelu" else hip.ACT_NONE, res=out if mode == "res" else None))
                r.append(f"lda={lda:4d} ldw={ldw:4d}: {us:6.1f}us {2.0*M*N*K/us/1e6:5.0f}TF")
            print(f"M={M} N={N} K={K} {mode} v{variant} flags={flags}: " + " | ".join(r), flush=True)
lib.ufm_debug_set_gemm_flags(0); lib.ufm_debug_set_gemm_variant(0)
