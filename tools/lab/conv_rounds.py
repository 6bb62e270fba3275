#!/usr/bin/env python3
"""Per-round cost of the bf16x3 convolution kernels: a 3x3 256->256 layer on B x S x S pixels where B*S*S is an exact number
of 256-pixel tiles (whole rounds of the chip for the 8-phase kernel), through each kernel choice (ufm_debug_set_conv_variant:
0 auto, 1 = 128-row kernels only, 2 = 8-phase on everything).  Prints us per launch, TF-alg and the cycles per K-tile and CU
the time corresponds to at 2.0 GHz (a K-tile of the 8-phase kernel is 3072 cycles of MFMA issue per SIMD)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip

lib = hip.lib()
DEV = "cuda"


def timeit(fn, iters=8, warm=3):
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
    return ts[len(ts) // 2]


zero = torch.zeros(512, device=DEV)
torch.manual_seed(0)
for (B, S, cin, cout, k, tag) in ((4, 128, 256, 256, 3, "1 round"), (8, 128, 256, 256, 3, "2 rounds"), (12, 128, 256, 256, 3, "3 rounds"),
                                  (8, 148, 256, 256, 3, "RCU 148^2 (2.67 rounds)"), (8, 128, 256, 256, 1, "1x1, 2 rounds"),
                                  (8, 296, 256, 128, 3, "pc1 296^2"),
                                  (8, 74, 256, 256, 3, "RCU 74^2 B8"), (4, 74, 256, 256, 3, "RCU 74^2 B4"), (4, 148, 256, 256, 3, "RCU 148^2 B4"),
                                  (8, 37, 256, 256, 3, "RCU 37^2 B8"), (4, 37, 256, 256, 3, "RCU 37^2 B4"), (8, 148, 96, 256, 3, "rn0 148^2 96->256"),
                                  (8, 148, 256, 256, 1, "out_conv 148^2 1x1")):
    x = torch.randn(2, B, S, S, cin, device=DEV).bfloat16()
    w = (torch.randn(2, cout, k, k, cin, device=DEV) * (cin * k * k) ** -0.5).bfloat16()
    out = torch.empty(2, B, S, S, cout, device=DEV, dtype=torch.bfloat16)
    fl = 2.0 * B * S * S * cout * k * k * cin
    row = []
    for variant, name in ((0, "auto"), (1, "128-row"), (2, "8-phase")):
        if variant == 2 and cout % 256:
            continue
        lib.ufm_debug_set_conv_variant(variant)
        ms = timeit(lambda: hip.conv2d_x3(x, B, S, S, cin, w, cout, k, k, 1, k // 2, out, zero))
        lib.ufm_debug_set_conv_variant(0)
        tiles = (B * S * S + 255) // 256 * (cout // 256 if cout % 256 == 0 else 1)
        rounds = -(-tiles // 256)
        nt = k * k * cin // 32
        row.append(f"{name}: {ms*1e3:7.1f} us {fl/ms/1e9:6.1f} TF-alg" + (f" ({ms*1e-3*2.0e9/rounds/nt:6.0f} cyc/K-tile at 2 GHz)" if variant == 2 else ""))
    print(f"{tag:26s} " + " | ".join(row), flush=True)
