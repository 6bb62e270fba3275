#!/usr/bin/env python3
"""Round 6 A/B: the bf16x3 Linear layer on interleaved split operands (ufm_gemm_bf16x3_il: 128-byte DMA rows) against the
planar form (ufm_gemm_bf16x3: two 64-byte halves a plane apart) at the eight pipeline shapes, micro-batch rows on a stream flagged as concurrent
(the engine's dispatch: full-height tiles) and full-batch rows unflagged; interleaved on one box, operands cold (a 512-MB fill between launches).
The outputs (and, for proj / fc2, the fp32 residual stream) are written in their usual formats by both arms."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip

lib = hip.lib()
DEV = "cuda"
flush = torch.empty(512 << 20, device=DEV, dtype=torch.uint8)
side = torch.cuda.Stream()


def timeit(fn, stream, iters=10, warm=3):
    with torch.cuda.stream(stream):
        for _ in range(warm):
            fn()
        stream.synchronize()
        ts = []
        for _ in range(iters):
            flush.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            fn()
            e1.record(stream)
            stream.synchronize()
            ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2] * 1e3


zero = torch.zeros(512, device=DEV)
torch.manual_seed(0)
shapes = (("enc QKV", 3072, 1024, "split"), ("enc proj", 1024, 1024, "rmw"), ("enc fc1", 4096, 1024, "gelu"), ("enc fc2", 1024, 4096, "rmw"),
          ("info QKV", 2304, 768, "split"), ("info proj", 768, 768, "rmw"), ("info fc1", 3072, 768, "gelu"), ("info fc2", 768, 3072, "rmw"))
for flagged, M in ((True, 10960), (False, 21920)):
    hip.hint_concurrent_stream(side, flagged)
    print(f"## M = {M}, stream {'flagged concurrent (CU-time tile policy)' if flagged else 'unflagged (latency tile policy)'}")
    for tag, N, K, mode in shapes:
        A = torch.randn(2, M, K, device=DEV).bfloat16(); A[1] *= 2.0 ** -9
        W = (torch.randn(2, N, K, device=DEV) * K ** -0.5).bfloat16(); W[1] *= 2.0 ** -9
        Ai, Wi = hip.interleave_split(A), hip.interleave_split(W)
        bias, gamma = torch.randn(N, device=DEV) * 0.1, 1 + 0.1 * torch.randn(N, device=DEV)
        if mode == "rmw":
            out = torch.randn(M, N, device=DEV)
            kw = dict(bias=bias, gamma=gamma, res=out)
        else:
            out = torch.empty(2, M, N, device=DEV, dtype=torch.bfloat16)
            kw = dict(bias=bias, act=hip.ACT_GELU if mode == "gelu" else hip.ACT_NONE, gamma=None if mode == "gelu" else gamma)
        t = {"planar": [], "il": []}
        for rep in range(2):
            t["planar"].append(timeit(lambda: hip.gemm_x3(A, W, M, N, K, out, zero, **kw), side))
            t["il"].append(timeit(lambda: hip.gemm_x3_il(Ai, Wi, M, N, K, out, zero, **kw), side))
        a, b = min(t["planar"]), min(t["il"])
        fl = 2.0 * M * N * K
        print(f"{tag:10s} N{N:5d} K{K:5d} {mode:5s} planar {t['planar'][0]:7.1f} {t['planar'][1]:7.1f} us | interleaved {t['il'][0]:7.1f} {t['il'][1]:7.1f} us | {100 * (b / a - 1):+5.1f} %"
              f"  il = {fl / b / 1e6 / 833.3:.3f} of the /3 peak", flush=True)
hip.hint_concurrent_stream(side, False)
