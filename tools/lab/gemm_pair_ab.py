#!/usr/bin/env python3
"""A/B of the 256x128 two-resident-workgroups GEMM (variant 6, gemm_bf16_pair.hip) against the shipped dispatch (variant 0) on the
eight pipeline shapes.  Two timings per arm, interleaved rounds in ONE process, medians:
  hot  = back-to-back launches (the residual / output stays in the Infinity Cache)
  cold = one launch behind a 1 GiB streaming write (the pipeline's state: the residual was last touched ~0.5 GB of traffic ago)
Also checks that variant 6 is bitwise the 128x128 kernel.  STAGGER=a,b,c (units of s_sleep(64)) adds first-round stagger arms."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip
lib = hip.lib()

# (rows per pair-batch of 8: encoder 16 x 1370, info sharing 8 x 2738), N, K, epilogue
SHAPES = ((21920, 3072, 1024, "bf16"), (21920, 1024, 1024, "res"), (21920, 4096, 1024, "gelu"), (21920, 1024, 4096, "res"),
          (21904, 2304, 768, "bf16"), (21904, 768, 768, "res"), (21904, 3072, 768, "gelu"), (21904, 768, 3072, "res"))
if os.environ.get("SHAPES"):
    keep = set(os.environ["SHAPES"].split(","))
    SHAPES = tuple(s for s in SHAPES if f"{s[1]}x{s[2]}" in keep)
MDIV = int(os.environ.get("MDIV", "1"))  # 2 = the micro-batch shapes (4 pairs per stream)
STAG = [int(x) for x in os.environ.get("STAGGER", "0").split(",")]
REPS = int(os.environ.get("REPS", "9"))
flush = torch.empty(1 << 28, device="cuda", dtype=torch.float32)  # 1 GiB


def med(x):
    return sorted(x)[len(x) // 2]


for _one in (0,):
    for M, N, K, mode in SHAPES:
        Mx = M // MDIV
        A = torch.randn(Mx, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") * K**-0.5).bfloat16()
        bias = torch.randn(N, device="cuda") * 0.1
        gamma = 1 + 0.1 * torch.randn(N, device="cuda")
        res0 = torch.randn(Mx, N, device="cuda")

        def run(out):
            hip.gemm_bf16(A, W, Mx, N, K, out, bias=bias, act=hip.ACT_GELU if mode == "gelu" else hip.ACT_NONE, res=out if mode == "res" else None,
                          gamma=gamma if mode != "gelu" else None)

        def fresh():
            return res0.clone() if mode == "res" else torch.zeros(Mx, N, device="cuda", dtype=torch.bfloat16)

        arms = [("auto", 0, 0, 0)] + [(f"pair s{s}", 6, s << 16, 0) for s in STAG]
        arms += [(f"pair r{r}", 6, 0, int(r)) for r in os.environ.get("ROWS", "").split(",") if r]
        if os.environ.get("PERSIST") and mode != "res":  # auto with / without the persistent 8-phase kernel (flag bit 28 = never)
            arms = [("auto", 0, 0, 0), ("auto, no persistent", 0, 1 << 28, 0)]
        if os.environ.get("STAG8"):  # first-round start stagger of the 8-phase kernel (flag bits 16..18: (block / 8) % 4 x units x ~1 us)
            arms = [("8-phase", 0, 0, 0)] + [(f"8-phase stagger {u}", 0, int(u) << 16, 0) for u in os.environ["STAG8"].split(",")]
        if os.environ.get("OLD_EPI") and mode == "res":  # the serial read-modify-write read-out of rounds 1-4 (flag 0x800000)
            arms += [("auto old-epi", 0, 0x800000, 0), ("pair old-epi", 6, 0x800000, 0)]
        lib.ufm_debug_set_gemm_variant(1); lib.ufm_debug_set_gemm_flags(0); lib.ufm_debug_set_gemm_tile_rows(0)
        ref = fresh(); run(ref); torch.cuda.synchronize()
        same = {}
        for name, v, f, r in arms:
            lib.ufm_debug_set_gemm_variant(v); lib.ufm_debug_set_gemm_flags(f); lib.ufm_debug_set_gemm_tile_rows(r)
            o = fresh(); run(o); torch.cuda.synchronize()
            same[name] = torch.equal(o.view(torch.uint8), ref.view(torch.uint8))
        scratch = fresh()
        hot = {a[0]: [] for a in arms}; cold = {a[0]: [] for a in arms}
        arms = [a for a in arms]
        for _ in range(REPS):
            for name, v, f, r in arms:
                lib.ufm_debug_set_gemm_variant(v); lib.ufm_debug_set_gemm_flags(f); lib.ufm_debug_set_gemm_tile_rows(r)
                run(scratch); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): run(scratch)
                e1.record(); torch.cuda.synchronize()
                hot[name].append(e0.elapsed_time(e1) / 5 * 1e3)
                c = []
                for _ in range(3):
                    flush.fill_(1.0)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); run(scratch); e1.record(); torch.cuda.synchronize()
                    c.append(e0.elapsed_time(e1) * 1e3)
                cold[name].append(med(c))
        lib.ufm_debug_set_gemm_variant(0); lib.ufm_debug_set_gemm_flags(0); lib.ufm_debug_set_gemm_tile_rows(0)
        fl = 2.0 * Mx * N * K
        print(f"M={Mx} N={N} K={K} {mode:4s}: " + " | ".join(
            f"{n}: hot {med(hot[n]):6.1f} cold {med(cold[n]):6.1f} us ({fl / med(cold[n]) / 1e6:5.0f} TF){'' if same[n] else ' MISMATCH'}" for n, _, _, _ in arms), flush=True)
