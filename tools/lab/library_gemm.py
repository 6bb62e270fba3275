#!/usr/bin/env python3
"""Reference point for the GEMM roofline discussion: the vendor library (torch.nn.functional.linear -> hipBLASLt / rocBLAS,
bf16 in, bf16 out, bias) on the UFM shapes next to ufm_gemm_bf16 with the same epilogue.  Not used by the product."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip
hip.lib()
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (10952, 21904):
    for N, K in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096), (2304, 768), (768, 3072), (8192, 8192)):
        A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16()
        bias = torch.randn(N, device="cuda"); bias_b = bias.bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        us_lib = t(lambda: F.linear(A, W, bias_b))
        us_lib_nb = t(lambda: torch.matmul(A, W.t()))
        us_ours = t(lambda: hip.gemm_bf16(A, W, M, N, K, out, bias=bias))
        f = 2.0 * M * N * K / 1e6
        print(f"M={M} N={N} K={K}: library linear+bias {us_lib:6.1f}us {f/us_lib:5.0f}TF | library matmul {us_lib_nb:6.1f}us {f/us_lib_nb:5.0f}TF | ufm_gemm_bf16 (bias, bf16 out) {us_ours:6.1f}us {f/us_ours:5.0f}TF", flush=True)
