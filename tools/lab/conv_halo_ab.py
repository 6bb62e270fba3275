#!/usr/bin/env python3
"""Round 6 A/B (three arms: gather, halo, halo with interleaved weights): the row-window halo form of the 8-phase bf16x3 3x3 kernel (conv_bf16x3_halo.hip) against the gather form (conv_bf16x3_8ph.hip;
ufm_debug_set_conv_variant HALO field = 1) on the DPT heads' 3x3 / stride 1 layers, interleaved on one box, cold-ish (a 512-MB buffer is
written between launches so that neither arm finds its operands in L2 / the Infinity Cache by accident), median of `iters` launches.
Also the bitwise comparison of the two outputs.  HALO_AB_NF=n pins the tile height (32 n rows) in both arms."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip

lib = hip.lib()
DEV = "cuda"
HALO_OFF, HALO_PLANAR_W = 1 << 5, 2 << 5  # conv variant field HALO: 1 = the gather form, 2 = the halo form with planar W staging even if an interleaved copy is registered
NF = int(os.environ.get("HALO_AB_NF", "0"))
KERNEL = 2 if NF else 0  # a pinned height needs the 8-phase kernel forced
flush = torch.empty(512 << 20, device=DEV, dtype=torch.uint8)


def timeit(fn, iters=12, warm=3, cold=True):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        if cold:
            flush.fill_(1)
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
shapes = ((8, 148, 256, 256, 1, "148^2 256->256 + skip (RCU conv2)"), (8, 148, 256, 256, 0, "148^2 256->256 (RCU conv1)"), (8, 148, 96, 256, 0, "148^2 96->256 (layer_rn)"),
          (8, 74, 256, 256, 1, "74^2 256->256 + skip"), (8, 74, 192, 256, 0, "74^2 192->256 (layer_rn)"), (4, 148, 256, 256, 1, "148^2 256->256 + skip, B = 4"),
          (8, 37, 256, 256, 1, "37^2 256->256 + skip"), (8, 112, 256, 256, 0, "default-res 120x160-ish: 112^2"))
for (B, S, cin, cout, nres, tag) in shapes:
    x = torch.randn(2, B, S, S, cin, device=DEV).bfloat16()
    x[1] *= 2.0 ** -9
    w = (torch.randn(2, cout, 3, 3, cin, device=DEV) * (cin * 9) ** -0.5).bfloat16()
    w[1] *= 2.0 ** -9
    bias = torch.randn(cout, device=DEV) * 0.1
    res = torch.randn(2, B, S, S, cout, device=DEV).bfloat16() if nres else None
    outs, t = {}, {}
    fl = 2.0 * B * S * S * cout * 9 * cin
    base = KERNEL | (NF << 8)
    w_il = hip.interleave_split(w.view(2, cout, 9 * cin))  # [cout][9 cin / 32][hi 32 | lo 32]
    assert lib.ufm_conv_x3_register_interleaved_weights(w.data_ptr(), w_il.data_ptr()) == 0
    for rep in range(2):  # interleaved arms
        for name, variant in (("gather", base | HALO_OFF), ("halo", base | HALO_PLANAR_W), ("halo_wil", base)):
            assert lib.ufm_debug_set_conv_variant(variant) == 0
            out = torch.empty(2, B, S, S, cout, device=DEV, dtype=torch.bfloat16)
            ms = timeit(lambda: hip.conv2d_x3(x, B, S, S, cin, w, cout, 3, 3, 1, 1, out, zero, bias=bias, res1=res))
            t.setdefault(name, []).append(ms * 1e3)
            outs[name] = out.view(torch.int16).clone()
    lib.ufm_debug_set_conv_variant(0)
    assert lib.ufm_conv_x3_register_interleaved_weights(w.data_ptr(), None) == 0
    g, h, hw = min(t["gather"]), min(t["halo"]), min(t["halo_wil"])
    print(f"{tag:40s} gather {t['gather'][0]:7.1f} {t['gather'][1]:7.1f} us | halo {t['halo'][0]:7.1f} {t['halo'][1]:7.1f} us ({100 * (h / g - 1):+5.1f} %) | halo + interleaved W {t['halo_wil'][0]:7.1f} {t['halo_wil'][1]:7.1f} us ({100 * (hw / g - 1):+5.1f} %)  "
          f"{fl / hw / 1e6:5.0f} TF-alg = {fl / hw / 1e6 / 833.3:.3f} of the /3 peak | bitwise equal: {bool(torch.equal(outs['gather'], outs['halo']))} {bool(torch.equal(outs['gather'], outs['halo_wil']))}", flush=True)
