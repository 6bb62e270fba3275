#!/usr/bin/env python3
"""Time ufm_dpt_tail_fused at the UFM-Base tail shape (B x 296^2 x 128 -> B x 518^2)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ufm_amd import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
h = w = 296; H = W = 518
x = torch.randn(2, B, h, w, 128, device="cuda").bfloat16()
w2 = (torch.randn(2, 32, 3, 3, 128, device="cuda") * 0.03).bfloat16()
b2 = torch.randn(32, device="cuda") * 0.1
wt, bt = torch.randn(2, 32, device="cuda") * 0.3, torch.randn(2, device="cuda") * 0.1
out = torch.empty(B, 2, H, W, device="cuda")
def run():
    hip.dpt_tail_fused(x, B, h, w, 128, w2, b2, 32, H, W, wt, bt, 2, [0, 0], [1.0, 1.0], [0.0, 0.0], out, None)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"B={B}: {ms*1e3:.0f} us  ({2.0*B*H*W*32*9*128*3/ms/1e9:.0f} TF of bf16 MFMA)")
