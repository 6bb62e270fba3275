#!/usr/bin/env python3
"""bf16x3 conv at the RCU-148 shape: cost of relu-on-load and of the split residual epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ufm_amd import hip
from tools.kbench import timeit
B, h, cin, cout, k = 8, 148, 256, 256, 3
x = torch.randn(2, B, h, h, cin, device="cuda").bfloat16()
w = (torch.randn(2, cout, k, k, cin, device="cuda") * (cin * k * k) ** -0.5).bfloat16()
bias = torch.randn(cout, device="cuda")
r1 = torch.randn(2, B, h, h, cout, device="cuda").bfloat16()
r2 = torch.randn(2, B, h, h, cout, device="cuda").bfloat16()
out = torch.empty(2, B, h, h, cout, device="cuda", dtype=torch.bfloat16)
zero = torch.zeros(256, device="cuda")
fl = 2.0 * B * h * h * cout * k * k * cin
for relu in (False, True):
    for nres in (0, 1, 2):
        med, _ = timeit(lambda: hip.conv2d_x3(x, B, h, h, cin, w, cout, k, k, 1, 1, out, zero, relu_in=relu, bias=bias, res1=r1 if nres > 0 else None, res2=r2 if nres > 1 else None), iters=8, warm=2)
        print(f"relu_in={relu} residuals={nres}: {med*1e3:.1f} us  {fl/med/1e9:.1f} TF(alg)", flush=True)
