#!/usr/bin/env python3
"""Where do the halo and the gather form of the 8-phase 3x3 kernel differ?  (debug aid, round 6)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip
lib = hip.lib()
DEV = "cuda"
zero = torch.zeros(512, device=DEV)
torch.manual_seed(0)
for (B, H, W, cin, cout) in ((2, 40, 40, 32, 256), (1, 64, 32, 64, 256), (5, 148, 148, 64, 256)):
    x = torch.randn(2, B, H, W, cin, device=DEV).bfloat16(); x[1] *= 2.0 ** -9
    w = (torch.randn(2, cout, 3, 3, cin, device=DEV) * (cin * 9) ** -0.5).bfloat16(); w[1] *= 2.0 ** -9
    outs = {}
    for name, v in (("gather", 2 | 32), ("halo", 2)):
        assert lib.ufm_debug_set_conv_variant(v) == 0
        out = torch.full((2, B, H, W, cout), 7.0, device=DEV, dtype=torch.bfloat16)
        hip.conv2d_x3(x, B, H, W, cin, w, cout, 3, 3, 1, 1, out, zero)
        torch.cuda.synchronize()
        outs[name] = (out[0].float() + out[1].float()).cpu()
    lib.ufm_debug_set_conv_variant(0)
    d = (outs["gather"] - outs["halo"]).abs().amax(dim=-1)  # [B, H, W]
    bad = (d > 0).nonzero()
    print(f"B{B} {H}x{W} {cin}->{cout}: {bad.shape[0]} pixels differ of {B*H*W}; max diff {d.max().item():.3g}")
    flat = (bad[:, 0] * H * W + bad[:, 1] * W + bad[:, 2]).tolist()
    print("  first flat indices:", flat[:40])
    print("  (b, y, x):", bad[:24].tolist())
    # per-channel pattern of the first bad pixel
    if bad.shape[0]:
        b0, y0, x0 = bad[0].tolist()
        dc = (outs["gather"][b0, y0, x0] - outs["halo"][b0, y0, x0]).abs()
        print("  channels differing at the first bad pixel:", int((dc > 0).sum()), "of", cout)
