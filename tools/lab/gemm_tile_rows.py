#!/usr/bin/env python3
"""8-phase GEMM tile height (160/192/224/256 rows) per UFM shape: time, and bitwise equality with the 128x128 kernel."""
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
Ms = [int(x) for x in os.environ.get("MS", "10952").split(",")]
for M in Ms:
  for N, K, mode in ((3072, 1024, "bf16"), (1024, 1024, "res"), (4096, 1024, "gelu"), (1024, 4096, "res"), (2304, 768, "bf16"), (768, 768, "res"), (3072, 768, "gelu"), (768, 3072, "res")):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16()
    bias = torch.randn(N, device="cuda")
    res0 = torch.randn(M, N, device="cuda")
    def run(out):
        hip.gemm_bf16(A, W, M, N, K, out, bias=bias, act=hip.ACT_GELU if mode == "gelu" else hip.ACT_NONE, res=out if mode == "res" else None)
    def fresh():
        return res0.clone() if mode == "res" else torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    lib.ufm_debug_set_gemm_variant(1); lib.ufm_debug_set_gemm_tile_rows(0)
    ref = fresh(); run(ref); torch.cuda.synchronize()
    scratch = fresh()
    r = [f"128x128: {t(lambda: run(scratch)):6.1f}"]
    for variant, rows in ((4, 160), (4, 192), (4, 224), (4, 256), (5, 0), (0, 0)):
        lib.ufm_debug_set_gemm_variant(variant); lib.ufm_debug_set_gemm_tile_rows(rows)
        o = fresh(); run(o); torch.cuda.synchronize()
        same = torch.equal(o.view(torch.uint8), ref.view(torch.uint8))
        r.append(f"{'v%d' % variant if not rows else rows}: {t(lambda: run(scratch)):6.1f}{'' if same else ' MISMATCH'}")
    lib.ufm_debug_set_gemm_variant(0); lib.ufm_debug_set_gemm_tile_rows(0)
    print(f"M={M} N={N} K={K} {mode}: " + " | ".join(r), flush=True)
