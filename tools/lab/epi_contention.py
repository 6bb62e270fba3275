#!/usr/bin/env python3
"""How much of a GEMM epilogue is contention?  One-round launches of the 8-phase kernel on 32..256 CUs (tile rows 256), with
and without epilogue traffic (flags 4), and the first round's start staggered (flags 0x10000 * units).  Interleaved rounds."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip
lib = hip.lib()


def bench(M, N, K, mode, flag_sets, rows=256, reps=7, inner=5):
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16()
    bias = torch.randn(N, device="cuda") * 0.1
    gamma = 1 + 0.1 * torch.randn(N, device="cuda")
    out = torch.randn(M, N, device="cuda") if mode == "res" else torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    def run():
        hip.gemm_bf16(A, W, M, N, K, out, bias=bias, act=hip.ACT_GELU if mode == "gelu" else hip.ACT_NONE, res=out if mode == "res" else None,
                      gamma=gamma if mode != "gelu" else None)
    lib.ufm_debug_set_gemm_variant(4)
    lib.ufm_debug_set_gemm_tile_rows(rows)
    times = {f: [] for f in flag_sets}
    for _ in range(reps):
        for f in flag_sets:
            lib.ufm_debug_set_gemm_flags(f)
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner):
                run()
            e1.record(); torch.cuda.synchronize()
            times[f].append(e0.elapsed_time(e1) / inner * 1e3)
    lib.ufm_debug_set_gemm_flags(0); lib.ufm_debug_set_gemm_variant(0); lib.ufm_debug_set_gemm_tile_rows(0)
    return {f: sorted(t)[len(t) // 2] for f, t in times.items()}


if "--nt-only" not in sys.argv:
    print("== one round, varying the number of busy CUs: time with epilogue / without (flags 4) ==")
    for N, K, mode in ((1024, 1024, "res"), (3072, 1024, "bf16"), (4096, 1024, "gelu"), (1024, 4096, "res")):
        ntn = N // 256
        for tiles in (32, 64, 128, 256):
            if tiles % ntn: continue
            M = 256 * (tiles // ntn)
            r = bench(M, N, K, mode, (0, 4))
            print(f"N={N} K={K} {mode}: tiles={tiles:4d} M={M:6d}  full {r[0]:7.1f} us  no-epilogue {r[4]:7.1f} us  epilogue {r[0]-r[4]:6.1f} us", flush=True)

    print("== staggered first round at the benchmark shapes (units x ~1 us x (block/8)%4) ==")
    for M, N, K, mode, rows in ((21920, 1024, 1024, "res", 192), (21920, 1024, 4096, "res", 192), (21920, 3072, 1024, "bf16", 256), (21920, 4096, 1024, "gelu", 256),
                                (10960, 1024, 1024, "res", 192), (10960, 3072, 1024, "bf16", 256)):
        r = bench(M, N, K, mode, (0, 0x10000, 0x20000, 0x30000, 0x50000), rows=rows)
        print(f"M={M} N={N} K={K} {mode} rows={rows}: " + "  ".join(f"stagger {f >> 16}: {t:6.1f} us" for f, t in r.items()), flush=True)

# (A third arm -- nontemporal residual loads / stores in the read-modify-write epilogue -- was measured in a lab build and reverted:
#  6-28 % slower on every shape; profiles/r04/gemm_epilogue_contention.log section (c).)
