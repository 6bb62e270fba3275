#!/usr/bin/env python3
"""A/B of GEMM variants on the UFM shapes, interleaved rounds in one process (methodology rule 24), random operands."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ufm_amd import hip
lib = hip.lib()
# variant codes: 0 auto, 1 = 128x128, 4 = 8-phase, 5 = hybrid; +40 = the same with flags 4 (no epilogue traffic: main loop only)
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "4"])]


def set_variant(v):  # +80: generic (run-time switched) epilogue instead of the compile-time specialised one
    lib.ufm_debug_set_gemm_flags(64 if v >= 80 else 4 if v >= 40 else 0)
    lib.ufm_debug_set_gemm_variant(v - 80 if v >= 80 else v - 40 if v >= 40 else v)
shapes = [(4096, 4096, 4096, "bf16"), (8192, 8192, 8192, "bf16"), (21904, 3072, 128, "bf16"), (21904, 3072, 128, "f32"), (21904, 3072, 256, "bf16"), (21904, 1024, 128, "res"), (21904, 1024, 128, "f32"),
          (21904, 3072, 1024, "bf16"), (21904, 1024, 1024, "res"), (21904, 4096, 1024, "gelu"), (21904, 1024, 4096, "res"),
          (10952, 3072, 1024, "bf16"), (10952, 1024, 1024, "res"), (10952, 4096, 1024, "gelu"), (10952, 1024, 4096, "res"),
          (21904, 2304, 768, "bf16"), (21904, 768, 768, "res"), (21904, 3072, 768, "gelu"), (21904, 768, 3072, "res")]
for M, N, K, mode in shapes:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16()
    bias = torch.randn(N, device="cuda") * 0.1
    gamma = (1 + 0.1 * torch.randn(N, device="cuda")) if os.environ.get("GAMMA") == "1" else None
    out = torch.randn(M, N, device="cuda") if mode in ("res", "f32") else torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    def run():
        hip.gemm_bf16(A, W, M, N, K, out, bias=bias, act=hip.ACT_GELU if mode == "gelu" else hip.ACT_NONE, res=out if mode == "res" else None,
                      gamma=gamma if mode != "gelu" else None)
    times = {v: [] for v in variants}
    for rnd in range(7):
        for v in variants:
            set_variant(v)
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 5)
    set_variant(0)
    msg = f"M={M} N={N} K={K} {mode}: "
    for v in variants:
        t = sorted(times[v]); med = t[len(t) // 2]
        msg += f" v{v}: {med*1e3:.0f}us {2.0*M*N*K/med/1e9:.0f}TF (min {t[0]*1e3:.0f})"
    print(msg, flush=True)
