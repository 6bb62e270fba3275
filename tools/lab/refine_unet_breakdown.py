#!/usr/bin/env python3
"""UFM-Refine + UNet fine features (config-4 variant), 518^2, B=8: step time and per-kernel breakdown."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ufm_amd
from ufm_amd import hip
from ufm_amd.modules import init_weights_
B = 8
for method in ("conv",):
    m = ufm_amd.UniFlowMatchClassificationRefinement(**ufm_amd.ufm_refine_config(use_unet_feature=True, feature_combine_method=method)).eval()
    init_weights_(m, 0)
    m = m.to("cuda")
    g = torch.Generator().manual_seed(0)
    s = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
    t = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
    for _ in range(3): m.predict_correspondences_batched(s, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.predict_correspondences_batched(s, t)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"UFM-Refine + UNet ({method}) 518, B={B}: {B/dt:.1f} pairs/s  {dt*1e3:.2f} ms/step", flush=True)
    hip.TIMER = hip.KernelTimer()
    m.predict_correspondences_batched(s, t)
    summ = hip.TIMER.summary(); hip.TIMER = None
    for k, d in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:14]:
        print(f"   {k:30s} x{d['launches']:3d}  {d['ms']:6.2f} ms", flush=True)
