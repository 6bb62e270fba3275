#!/usr/bin/env python3
"""ufm_dpt_tail_fused at the UFM-Base tail shape: round 5's half-swapped stage-A tile (default) against the plain tile of rounds 1-4
(ufm_debug_set_upsample_variant 3), interleaved rounds, medians, bitwise equality."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip
lib = hip.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
h = w = 296; H = W = 518
x = torch.randn(2, B, h, w, 128, device="cuda").bfloat16()
w2 = (torch.randn(2, 32, 3, 3, 128, device="cuda") * 0.03).bfloat16()
b2 = torch.randn(32, device="cuda") * 0.1
wt, bt = torch.randn(2, 32, device="cuda") * 0.3, torch.randn(2, device="cuda") * 0.1
outs = {v: torch.empty(B, 2, H, W, device="cuda") for v in (1, 3)}
def run(v):
    lib.ufm_debug_set_upsample_variant(v)
    hip.dpt_tail_fused(x, B, h, w, 128, w2, b2, 32, H, W, wt, bt, 2, [0, 0], [1.0, 1.0], [0.0, 0.0], outs[v], None)
times = {1: [], 3: []}
for _ in range(9):
    for v in (1, 3):
        run(v); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run(v)
        e1.record(); torch.cuda.synchronize()
        times[v].append(e0.elapsed_time(e1) / 5 * 1e3)
lib.ufm_debug_set_upsample_variant(1)
med = lambda t: sorted(t)[len(t) // 2]
print(f"B={B}: half-swapped T {med(times[1]):.1f} us | plain T {med(times[3]):.1f} us | bitwise equal: {torch.equal(outs[1], outs[3])}")
