#!/usr/bin/env python3
"""bf16x3 conv variants (1 = 128-row kernels, 2 = 256x256 8-phase, 0 = auto/hybrid) on the Cout=256 UFM layers,
interleaved rounds in one process (methodology rule 24), random operands."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ufm_amd import hip
lib = hip.lib()
B = 8
shapes = [(296, 256, 128, 3, False, 0), (148, 256, 128, 3, False, 0), (296, 128, 128, 3, True, 1), (148, 256, 256, 3, True, 1), (74, 256, 256, 3, True, 1), (37, 256, 256, 3, True, 1), (296, 256, 256, 3, False, 0),
          (148, 192, 256, 3, False, 0), (74, 384, 256, 3, False, 0), (148, 96, 256, 1, False, 0)]
for h, cin, cout, k, relu, nres in shapes:
    x = torch.randn(2, B, h, h, cin, device="cuda").bfloat16()
    w = (torch.randn(2, cout, k, k, cin, device="cuda") * (cin * k * k) ** -0.5).bfloat16()
    bias = torch.randn(cout, device="cuda")
    r1 = torch.randn(2, B, h, h, cout, device="cuda").bfloat16()
    out = torch.empty(2, B, h, h, cout, device="cuda", dtype=torch.bfloat16)
    zero = torch.zeros(256, device="cuda")
    fl = 2.0 * B * h * h * cout * k * k * cin
    def run():
        hip.conv2d_x3(x, B, h, h, cin, w, cout, k, k, 1, k // 2, out, zero, relu_in=relu, bias=bias, res1=r1 if nres else None)
    times = {v: [] for v in (1, 2, 4, 0)}
    for rnd in range(5):
        for v in times:
            lib.ufm_debug_set_conv_variant(v)
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                run()
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 4)
    lib.ufm_debug_set_conv_variant(0)
    msg = f"{h}x{h} Cin={cin} Cout={cout} k={k} relu={relu} res={nres}:"
    for v, t in times.items():
        t = sorted(t); med = t[len(t) // 2]
        msg += f"  v{v}: {med*1e3:.0f}us {fl/med/1e9:.0f}TF(alg)"
    print(msg, flush=True)
