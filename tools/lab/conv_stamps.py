#!/usr/bin/env python3
"""In-kernel stamps of the 8-phase bf16x3 convolution kernel (ufm_debug_set_conv_stamps): clock under load, cycles per workgroup in
the K loop and in the epilogue, the loop's MFMA issue share (24 MFMA x 16 cycles per phase and wave, two waves per SIMD), for the
heads' 8-phase layers; VARIANTS = conv variants to compare (2 = 8-phase everywhere, 34 = + product-major MFMA order)."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip
lib = hip.lib()
B = 8
SHAPES = [(148, 256, 256, 3, 1), (74, 256, 256, 3, 1), (148, 96, 256, 3, 0), (74, 192, 256, 3, 0), (37, 256, 256, 3, 1)]
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,2").split(",")]
if os.environ.get("ABLATE"):  # timing ablations of the stamped instantiation (conv variant bits 12..17: 1 no LDS-DMA, 2 no fragment reads, 4 no MFMA, 8 no slot barriers, 16 nothing removed, 32 one rendezvous per phase)
    VARIANTS = [2 | (int(a) << 12) for a in os.environ["ABLATE"].split(",")]
    SHAPES = [(148, 256, 256, 3, 1), (148, 1024, 256, 1, 0), (148, 96, 256, 3, 0)]
for h, cin, cout, k, nres in SHAPES:
    x = torch.randn(2, B, h, h, cin, device="cuda").bfloat16(); x[1] *= 2.0 ** -9
    w = (torch.randn(2, cout, k, k, cin, device="cuda") * (cin * k * k) ** -0.5).bfloat16(); w[1] *= 2.0 ** -9
    bias = torch.randn(cout, device="cuda")
    r1 = torch.randn(2, B, h, h, cout, device="cuda").bfloat16()
    out = torch.empty(2, B, h, h, cout, device="cuda", dtype=torch.bfloat16)
    zero = torch.zeros(256, device="cuda")
    fl = 2.0 * B * h * h * cout * k * k * cin
    def run():
        hip.conv2d_x3(x, B, h, h, cin, w, cout, k, k, 1, k // 2, out, zero, bias=bias, res1=r1 if nres else None)
    rows = 4096
    buf = torch.zeros(rows * 8, device="cuda", dtype=torch.int64)
    res = {}
    for v in VARIANTS:
        lib.ufm_debug_set_conv_variant(v)
        t0 = time.time()
        while time.time() - t0 < float(os.environ.get("WARM_S", "1.0")):
            for _ in range(20): run()
            torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): run()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        us = sorted(ts)[2]
        buf.zero_()
        hip._check(lib.ufm_debug_set_conv_stamps(buf.data_ptr(), rows), "stamps")
        for _ in range(5): run()
        torch.cuda.synchronize()
        hip._check(lib.ufm_debug_set_conv_stamps(None, 0), "stamps")
        d = buf.view(rows, 8).cpu(); d = d[d[:, 4] != 0].double()
        if d.shape[0] == 0:  # no stamped instantiation ran (a lower tile height, or the 128-row kernels)
            res[v] = dict(us=round(us, 1), tf_alg=round(fl / us / 1e6), frac_of_third_peak=round(fl / us / 1e6 / 833.3, 3))
            continue
        clock = float(((d[:, 4] - d[:, 2]) / (d[:, 6] - d[:, 5]).clamp_min(1) * 0.1).median())
        loop, epi = float((d[:, 3] - d[:, 2]).median()), float((d[:, 4] - d[:, 3]).median())
        nk = k * k * cin // 32
        mfma_cycles = nk * 4 * 24 * 16 * 2  # per SIMD: K-steps x phases x MFMAs x cycles x two waves
        res[v] = dict(us=round(us, 1), tf_alg=round(fl / us / 1e6), clock_ghz=round(clock, 3), workgroups=int(d.shape[0]), loop_cycles=loop, epilogue_cycles=epi,
                      loop_mfma_share=round(mfma_cycles / loop, 3), frac_of_third_peak=round(fl / us / 1e6 / 833.3, 3), frac_at_clock=round(fl / us / 1e6 / (833.3 * clock / 2.4), 3))
    lib.ufm_debug_set_conv_variant(0)
    print(f"B{B} {h}x{h} {cin}->{cout} k{k} res={nres}: " + json.dumps(res), flush=True)
