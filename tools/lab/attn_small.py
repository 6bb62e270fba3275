#!/usr/bin/env python3
"""ufm_attention_bf16 at small batches: 4-wave (256-row units) vs 2-wave (128-row units) workgroups."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip
lib = hip.lib()
torch.manual_seed(0)
for b, n, h in ((2, 1370, 16), (1, 2738, 12), (8, 1370, 16), (4, 2738, 12), (16, 1370, 16), (8, 2738, 12), (2, 10954, 12)):
    qkv = torch.randn(b * n, 3 * h * 64, device="cuda").bfloat16()
    o = torch.empty(b * n, h * 64, device="cuda", dtype=torch.bfloat16)
    r = []
    for v in (0, 1):
        lib.ufm_debug_set_attn_variant(v)
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
        r.append(f"v{v}: {ts[len(ts)//2]*1e3:6.1f}us")
    lib.ufm_debug_set_attn_variant(0)
    nqb4, nqb2 = -(-n // 256), -(-n // 128)
    print(f"B{b} N{n} H{h} (units {b*h*nqb4} / {b*h*nqb2}): " + " | ".join(r), flush=True)
