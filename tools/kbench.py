#!/usr/bin/env python3
"""Kernel micro-benchmarks at the UFM-Base B=8 518^2 shapes (run on the GPU box).
Interleaved rounds in ONE process (cdna_hip_programming.md rule 24), random data (rule 25)."""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ufm_amd import hip  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def gemm_cases(M):
    return [("qkv", M, 3072, 1024, 0, False), ("proj", M, 1024, 1024, 0, True), ("fc1", M, 4096, 1024, 1, False), ("fc2", M, 1024, 4096, 0, True),
            ("i_qkv", M // 2 * 2, 2304, 768, 0, False), ("i_proj", M, 768, 768, 0, True), ("i_fc1", M, 3072, 768, 1, False), ("i_fc2", M, 768, 3072, 0, True)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="gemm,attn,conv")
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    lib = hip.lib()
    B = args.batch
    res = {}
    if "gemm" in args.what:
        M = 2 * B * 1370
        for name, m, n, k, act, resid in gemm_cases(M):
            A = torch.randn(m, k, device=DEV).bfloat16()
            W = (torch.randn(n, k, device=DEV) * k**-0.5).bfloat16()
            bias = torch.randn(n, device=DEV)
            gamma = torch.ones(n, device=DEV) if resid else None
            out = torch.randn(m, n, device=DEV) if resid else torch.empty(m, n, device=DEV, dtype=torch.bfloat16)
            row = {}
            for variant, vname in ((0, "auto"), (1, "128x128"), (4, "8-phase")):
                lib.ufm_debug_set_gemm_variant(variant)
                med, mn = timeit(lambda: hip.gemm_bf16(A, W, m, n, k, out, bias=bias, act=act, gamma=gamma, res=out if resid else None))
                row[vname] = dict(ms=med, tflops=2.0 * m * n * k / med / 1e9)
            lib.ufm_debug_set_gemm_variant(0)
            res[f"gemm_{name}_{m}x{n}x{k}"] = row
            print(name, {k_: (round(v["ms"], 4), round(v["tflops"], 1)) for k_, v in row.items()}, flush=True)
    if "attn" in args.what:
        for name, b, n, h in (("enc", 2 * B, 1370, 16), ("info", B, 2738, 12)):
            qkv = torch.randn(b * n, 3 * h * 64, device=DEV).bfloat16()
            out = torch.empty(b * n, h * 64, device=DEV, dtype=torch.bfloat16)
            fl = 4.0 * b * h * n * n * 64
            qkv2 = qkv.clone()
            qkv2[:, : h * 64] = (qkv[:, : h * 64].float() * (0.125 * 1.4426950408889634)).bfloat16()
            for sc, vn, dbg in ((0.0, "pw4", 0), (0.0, "pw2", 1), (0.125, "scale>0 kernel", 0), (0.0, "pw4", 0), (0.0, "pw2", 1), (0.125, "scale>0 kernel", 0)):
                src = qkv if sc else qkv2
                lib.ufm_debug_set_attn_variant(dbg)
                med, mn = timeit(lambda: hip.attention(src, out, b, n, h, sc))
                lib.ufm_debug_set_attn_variant(0)
                res[f"attn_{name}_{vn}"] = dict(ms=med, tflops=fl / med / 1e9)
                print("attn", name, vn, round(med, 4), "ms", round(fl / med / 1e9, 1), "TF", flush=True)
    if "conv" in args.what:
        zero = torch.zeros(256, device=DEV)
        for name, h, cin, cout, k in (("rcu148", 148, 256, 256, 3), ("rcu74", 74, 256, 256, 3), ("pc1_296", 296, 256, 128, 3), ("pc2_518", 518, 128, 32, 3), ("out148", 148, 256, 256, 1)):
            x = torch.randn(2, B, h, h, cin, device=DEV).bfloat16()
            w = (torch.randn(2, cout, k, k, cin, device=DEV) * (cin * k * k) ** -0.5).bfloat16()
            out = torch.empty(2, B, h, h, cout, device=DEV, dtype=torch.bfloat16)
            med, mn = timeit(lambda: hip.conv2d_x3(x, B, h, h, cin, w, cout, k, k, 1, k // 2, out, zero), iters=6, warm=2)
            fl = 2.0 * B * h * h * cout * k * k * cin
            res[f"convx3_{name}"] = dict(ms=med, tflops=fl / med / 1e9)
            print("convx3", name, round(med, 4), "ms", round(fl / med / 1e9, 1), "TF(alg)", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
