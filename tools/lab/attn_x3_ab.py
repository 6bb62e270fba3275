#!/usr/bin/env python3
"""ufm_attention_bf16x3: the LDS-DMA kernel (attention_bf16x3_pw.hip) against the round-1 kernel (ufm_debug_set_attn_variant 2) on the
benchmark shapes, interleaved rounds in one process, medians; and bitwise equality of the outputs.  Round 6: variant 0 = the fixed softmax
reference (default), 8 = the running-maximum form of round 5 (bitwise the round-1 kernel), 12 = that with eight waves per workgroup."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip
lib = hip.lib()
for name, B, N, H in (("encoder", 16, 1370, 16), ("info sharing", 8, 2738, 12), ("config 5 joint", 2, 10954, 12)):
    qkv = torch.randn(2, B * N, 3 * H * 64, device="cuda").bfloat16()
    qkv[1] *= 2.0 ** -9
    outs, times = {}, {0: [], 2: [], 8: [], 12: []}
    for v in (2, 8, 0, 12):
        lib.ufm_debug_set_attn_variant(v)
        o = torch.zeros(2, B * N, H * 64, device="cuda", dtype=torch.bfloat16)
        hip.attention_x3(qkv, o, B, N, H, 0.125); torch.cuda.synchronize()
        outs[v] = o
    for _ in range(7):
        for v in (2, 8, 0, 12):
            lib.ufm_debug_set_attn_variant(v)
            o = outs[v]
            hip.attention_x3(qkv, o, B, N, H, 0.125); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): hip.attention_x3(qkv, o, B, N, H, 0.125)
            e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 5 * 1e3)
    lib.ufm_debug_set_attn_variant(0)
    fl = 4.0 * B * H * N * N * 64
    med = lambda x: sorted(x)[len(x) // 2]
    print(f"{name:15s} B={B} N={N} H={H}: round-1 {med(times[2]):8.1f} us ({fl/med(times[2])/1e6:5.0f} TF-alg, {fl/med(times[2])/1e6/833.3:.3f} of /3 peak) | "
          f"running maximum (round 5) {med(times[8]):8.1f} us ({fl/med(times[8])/1e6/833.3:.3f}) | fixed reference (round 6) {med(times[0]):8.1f} us ({fl/med(times[0])/1e6:5.0f} TF-alg, {fl/med(times[0])/1e6/833.3:.3f}, {100 * (med(times[0]) / med(times[8]) - 1):+.1f} %) | "
          f"running maximum, 8 waves {med(times[12]):8.1f} us ({fl/med(times[12])/1e6/833.3:.3f}) | bitwise the round-1 kernel: running maximum {torch.equal(outs[8].view(torch.int16), outs[2].view(torch.int16))} "
          f"{torch.equal(outs[12].view(torch.int16), outs[2].view(torch.int16))}, fixed reference max-abs diff {float((outs[0][0].float() + outs[0][1].float() - outs[2][0].float() - outs[2][1].float()).abs().max()):.2g}", flush=True)
