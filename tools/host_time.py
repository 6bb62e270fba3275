#!/usr/bin/env python3
"""Host enqueue time vs GPU time of one step (is the launch path the bottleneck?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ufm_amd
from ufm_amd.modules import init_weights_
B, res = 8, 518
model = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config(resolution_wh=(res, res)))
init_weights_(model, seed=0)
model = model.to("cuda").set_numerics("fast")
g = torch.Generator().manual_seed(0)
src = torch.randint(0, 256, (B, res, res, 3), generator=g, dtype=torch.uint8).cuda()
tgt = torch.randint(0, 256, (B, res, res, 3), generator=g, dtype=torch.uint8).cuda()
for mb in (1, 2):
    model.engine().micro_batches = mb
    for _ in range(3):
        model.predict_correspondences_batched(src, tgt)
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(8):
        t0 = time.perf_counter()
        model.predict_correspondences_batched(src, tgt)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append(t1 - t0); total.append(t2 - t0)
    print(f"micro_batches={mb}: host enqueue {1e3*sorted(host)[4]:.1f} ms, step {1e3*sorted(total)[4]:.1f} ms", flush=True)
