#!/usr/bin/env python3
"""A/B of engine switches on UFM-Refine (BASELINE config 4: 518^2, B=8), and bitwise check joint vs per-micro-batch heads."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ufm_amd
from ufm_amd.modules import init_weights_
B = 8
m = ufm_amd.UniFlowMatchClassificationRefinement(**ufm_amd.ufm_refine_config(resolution_wh=(518, 518))).eval()
init_weights_(m, 0)
m = m.to("cuda").set_numerics(os.environ.get("NUMERICS", "fast"))
g = torch.Generator().manual_seed(1)
s = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
t = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
outs = {}
for jh in (0, 1):
    m.engine().joint_heads = jh
    o = m.predict_correspondences_batched(s, t)
    outs[jh] = (o.flow.flow_output.clone(), o.covisibility.mask.clone())
print("joint == per-micro-batch heads, bitwise:", all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])))
times = {0: [], 1: []}
for r in range(6):
    for jh in (0, 1):
        m.engine().joint_heads = jh
        m.predict_correspondences_batched(s, t); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): m.predict_correspondences_batched(s, t)
        torch.cuda.synchronize(); times[jh].append((time.perf_counter() - t0) * 200)
for jh in (0, 1):
    ts = sorted(times[jh]); print(f"joint_heads={jh}: median {ts[3]:.2f} ms ({B*1e3/ts[3]:.1f} pairs/s)")
