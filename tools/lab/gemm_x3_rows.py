#!/usr/bin/env python3
"""ufm_gemm_bf16x3 (numerics "precise") on the eight pipeline shapes: the shipped dispatch against the 8-phase kernel pinned to tiles
of 160 / 192 / 224 / 256 rows (ufm_debug_set_conv_variant 2 | nf << 8) and against the 128-row kernels (1); interleaved, medians."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip
lib = hip.lib()
MDIV = int(os.environ.get("MDIV", "1"))
SHAPES = ((21920, 3072, 1024, "split"), (21920, 1024, 1024, "res"), (21920, 4096, 1024, "gelu"), (21920, 1024, 4096, "res"),
          (21904, 2304, 768, "split"), (21904, 768, 768, "res"), (21904, 3072, 768, "gelu"), (21904, 768, 3072, "res"))
zero = torch.zeros(256, device="cuda")
for M, N, K, mode in SHAPES:
    M //= MDIV
    A = torch.randn(2, M, K, device="cuda").bfloat16(); A[1] *= 2.0 ** -9
    W = (torch.randn(2, N, K, device="cuda") * K ** -0.5).bfloat16(); W[1] *= 2.0 ** -9
    bias, gamma = torch.randn(N, device="cuda") * 0.1, 1 + 0.1 * torch.randn(N, device="cuda")
    out = torch.randn(M, N, device="cuda") if mode == "res" else torch.empty(2, M, N, device="cuda", dtype=torch.bfloat16)
    def run():
        hip.gemm_x3(A, W, M, N, K, out, zero, bias=bias, act=hip.ACT_GELU if mode == "gelu" else hip.ACT_NONE, gamma=gamma if mode != "gelu" else None, res=out if mode == "res" else None)
    arms = [("auto", 0), ("128-row", 1)] + [(f"r{32 * nf}", 2 | (nf << 8)) for nf in (5, 6, 7, 8)]
    times = {a: [] for a, _ in arms}
    for _ in range(5):
        for a, v in arms:
            lib.ufm_debug_set_conv_variant(v)
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): run()
            e1.record(); torch.cuda.synchronize()
            times[a].append(e0.elapsed_time(e1) / 4 * 1e3)
    lib.ufm_debug_set_conv_variant(0)
    med = lambda t: sorted(t)[len(t) // 2]
    print(f"M={M} N={N} K={K} {mode:5s}: " + " | ".join(f"{a}: {med(t):6.1f}" for a, t in times.items()), flush=True)
