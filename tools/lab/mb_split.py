#!/usr/bin/env python3
"""Does an UNEVEN micro-batch split help?  With 4 + 4 pairs the two streams run identical kernel sequences of identical length and
tend to sit in the same kind of kernel at the same time; 5 + 3 (or three streams) lets HBM-bound launches of one fall beside
matrix-bound launches of the other.  Interleaved rounds in one process, UFM-Base B = 8 518^2 fast."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ufm_amd
from ufm_amd.modules import init_weights_
m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config()).eval()
init_weights_(m, 0)
m = m.to("cuda")
g = torch.Generator().manual_seed(1)
src = torch.randint(0, 256, (8, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
tgt = torch.randint(0, 256, (8, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
eng = m.engine()
cases = {"4+4": [0, 4, 8], "5+3": [0, 5, 8], "3+5": [0, 3, 8], "6+2": [0, 6, 8], "3+3+2": [0, 3, 6, 8], "2+2+2+2": [0, 2, 4, 6, 8], "8 (one stream)": None}
times = {k: [] for k in cases}
ref = None
for rnd in range(5):
    for name, b in cases.items():
        if b is None:
            eng.mb_bounds, eng.micro_batches = None, 1
        else:
            eng.mb_bounds, eng.micro_batches = b, 2
        for _ in range(2):
            o = m.predict_correspondences_batched(src, tgt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            o = m.predict_correspondences_batched(src, tgt)
        torch.cuda.synchronize()
        times[name].append((time.perf_counter() - t0) / 8 * 1e3)
        if ref is None:
            ref = o.flow.flow_output.clone()
        assert torch.equal(ref, o.flow.flow_output), name
for name, t in times.items():
    t = sorted(t)
    print(f"{name:16s} median {t[len(t)//2]:6.2f} ms  min {t[0]:6.2f}  ({8e3 / t[len(t)//2]:.1f} pairs/s)", flush=True)
