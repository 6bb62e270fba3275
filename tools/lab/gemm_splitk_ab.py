#!/usr/bin/env python3
"""Round 6 A/B: a layer-keyed deterministic 2-way split-K of the read-modify-write GEMMs (proj / fc2: `x = x + ls(proj/fc2(..))`
of every block, /root/reference/uniflowmatch/models/base.py:272-274 through the encoder / info-sharing blocks) -- ufm_debug_set_gemm_splitk.
 (1) isolated, residual cold: the four read-modify-write shapes at M = 10 960 on a stream flagged concurrent and M = 21 920 unflagged;
 (2) the two-stream pipeline (UFM-Base, B = 8, 518^2): pairs/s with the split on both micro-batch streams (min_k = 3072: fc2 only; min_k = 768: proj too)
     against off, interleaved rounds in one process.
Kill criterion of the review: keep per shape only where faster; if the pipeline gains < 1 % pairs/s, commit the log and stop."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ufm_amd
from ufm_amd import hip
from ufm_amd.modules import init_weights_
import ctypes as C

lib = hip.lib()
DEV = "cuda"
flush = torch.empty(512 << 20, device=DEV, dtype=torch.uint8)


def set_ws(stream, ws, min_k):
    rc = lib.ufm_debug_set_gemm_splitk(C.c_void_p(stream.cuda_stream), C.c_void_p(ws.data_ptr()) if ws is not None else None, ws.numel() * 4 if ws is not None else 0, min_k)
    assert rc == 0, lib.ufm_last_error()


def timeit(fn, stream, iters=12, warm=3):
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
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


torch.manual_seed(0)
side = torch.cuda.Stream()
ws = torch.zeros((65536 + 400 * 2 * 262144) // 4, device=DEV)
print("## (1) isolated launches, residual cold (a 512-MB fill between launches)")
for flagged, M in ((True, 10960), (False, 21920)):
    hip.hint_concurrent_stream(side, flagged)
    for tag, N, K in (("enc proj", 1024, 1024), ("enc fc2", 1024, 4096), ("info proj", 768, 768), ("info fc2", 768, 3072)):
        A = torch.randn(M, K, device=DEV).bfloat16()
        W = (torch.randn(N, K, device=DEV) * K ** -0.5).bfloat16()
        bias, gamma = torch.randn(N, device=DEV) * 0.1, 1 + 0.1 * torch.randn(N, device=DEV)
        out = torch.randn(M, N, device=DEV)
        t = {0: [], 1: []}
        for rep in range(2):
            for on in (0, 1):
                set_ws(side, ws if on else None, 256)
                t[on].append(timeit(lambda: hip.gemm_bf16(A, W, M, N, K, out, bias=bias, gamma=gamma, res=out), side))
        set_ws(side, None, 256)
        a, b = min(t[0]), min(t[1])
        print(f"M {M:5d} {'flagged  ' if flagged else 'unflagged'} {tag:10s} N{N:5d} K{K:5d}: unsplit {t[0][0]:7.1f} {t[0][1]:7.1f} us | split-K 2 {t[1][0]:7.1f} {t[1][1]:7.1f} us | {100 * (b / a - 1):+5.1f} %"
              f"   ({2.0 * M * N * K / b / 1e6 / 2500:.3f} of peak split, {2.0 * M * N * K / a / 1e6 / 2500:.3f} unsplit)", flush=True)
hip.hint_concurrent_stream(side, False)

print("## (2) the two-stream pipeline: UFM-Base, B = 8, 518^2, numerics fast")
B = 8
m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config()).eval()
init_weights_(m, 0)
m = m.to(DEV).set_numerics("fast")
g = torch.Generator().manual_seed(1)
src = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
tgt = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
for _ in range(3):
    m.predict_correspondences_batched(src, tgt)
torch.cuda.synchronize()
streams = list(m.engine()._streams)
wss = [torch.zeros((65536 + 200 * 2 * 262144) // 4, device=DEV) for _ in streams]
variants = {"off": None, "fc2 only (K >= 3072)": 3072, "proj + fc2 (K >= 768)": 768}
times = {k: [] for k in variants}
ref = None
for r in range(6):
    for name, mk in variants.items():
        for st, w in zip(streams, wss):
            set_ws(st, w if mk else None, mk or 256)
        o = m.predict_correspondences_batched(src, tgt)
        torch.cuda.synchronize()
        if r == 0:
            f = o.flow.flow_output.clone()
            if name == "off":
                ref = f
            else:
                print(f"   {name}: flow max-abs difference to the unsplit run {float((f - ref).abs().max()):.3g} px (other last bits: the K halves are summed separately)")
        t0 = time.perf_counter()
        for _ in range(5):
            m.predict_correspondences_batched(src, tgt)
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) * 1e3 / 5)
for st in streams:
    set_ws(st, None, 256)
base = sorted(times["off"])[3]
for name in variants:
    ts = sorted(times[name])
    print(f"{name:26s} median {ts[len(ts) // 2]:7.2f} ms  min {ts[0]:7.2f}  ({B * 1e3 / ts[len(ts) // 2]:.1f} pairs/s, {100 * (base / ts[len(ts) // 2] - 1):+.1f} % vs off)", flush=True)
