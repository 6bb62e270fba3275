#!/usr/bin/env python3
"""ufm_attention_bf16 launch time at the UFM-Base shapes, for same-box A/B of two builds (OLD=1: tolerate an older library)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip
if os.environ.get("OLD") == "1":
    import ctypes
    probe = ctypes.CDLL(hip.LIB_PATH)
    for k in list(hip.SIGNATURES):
        if not hasattr(probe, k):
            hip.SIGNATURES.pop(k)
hip.lib()
torch.manual_seed(0)
out = []
for b, n, h in ((8, 1370, 16), (4, 2738, 12), (16, 1370, 16), (2, 10954, 12)):
    qkv = torch.randn(b * n, 3 * h * 64, device="cuda").bfloat16()
    o = torch.empty(b * n, h * 64, device="cuda", dtype=torch.bfloat16)
    for _ in range(5): hip.attention(qkv, o, b, n, h, 0.0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): hip.attention(qkv, o, b, n, h, 0.0)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 4)
    ts.sort()
    us = ts[len(ts) // 2] * 1e3
    out.append(f"B{b} N{n} H{h}: {us:6.1f}us {4.0*b*h*n*n*64/us/1e6:5.0f}TF")
print(os.environ.get("TAG", "lib") + ": " + " | ".join(out), flush=True)
