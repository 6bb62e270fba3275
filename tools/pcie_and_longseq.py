#!/usr/bin/env python3
"""Two side measurements for DESIGN.md: (1) the PCIe-inclusive rate of the benchmark step (inputs start in pinned host
memory and the outputs are copied back), (2) BASELINE config 5 (1036x1036 pairs: 5477 / 10954 tokens)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ufm_amd
from ufm_amd.modules import init_weights_

def make(res):
    m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config(resolution_wh=(res, res))).eval()
    init_weights_(m, seed=0)
    return m.to("cuda").set_numerics("fast")

def rate(fn, n, B):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    return B / dt, dt * 1e3

B, res = 8, 518
m = make(res)
g = torch.Generator().manual_seed(0)
hs = torch.randint(0, 256, (B, res, res, 3), dtype=torch.uint8, generator=g).pin_memory()
ht = torch.randint(0, 256, (B, res, res, 3), dtype=torch.uint8, generator=g).pin_memory()
ds, dt_ = hs.cuda(), ht.cuda()
print("resident inputs      : %.1f pairs/s  %.2f ms/step" % rate(lambda: m.predict_correspondences_batched(ds, dt_), 15, B))
def step_pcie():
    o = m.predict_correspondences_batched(hs.cuda(non_blocking=True), ht.cuda(non_blocking=True))
    return o.flow.flow_output.cpu(), o.covisibility.mask.cpu()
print("host in, host out    : %.1f pairs/s  %.2f ms/step" % rate(step_pcie, 15, B))
del m; torch.cuda.empty_cache()
B, res = 2, 1036
m = make(res)
s = torch.randint(0, 256, (B, res, res, 3), dtype=torch.uint8, generator=g).cuda()
t = torch.randint(0, 256, (B, res, res, 3), dtype=torch.uint8, generator=g).cuda()
print("1036x1036, batch 2   : %.2f pairs/s  %.1f ms/step" % rate(lambda: m.predict_correspondences_batched(s, t), 5, B))
del m; torch.cuda.empty_cache()
# BASELINE config 4: UFM-Refine 518x518 (classification-refinement head on top of the same trunk)
B, res = 8, 518
m = ufm_amd.UniFlowMatchClassificationRefinement(**ufm_amd.ufm_refine_config(resolution_wh=(res, res))).eval()
init_weights_(m, seed=0)
m = m.to("cuda").set_numerics("fast")
s = torch.randint(0, 256, (B, res, res, 3), dtype=torch.uint8, generator=g).cuda()
t = torch.randint(0, 256, (B, res, res, 3), dtype=torch.uint8, generator=g).cuda()
print("UFM-Refine 518, B=8  : %.1f pairs/s  %.2f ms/step" % rate(lambda: m.predict_correspondences_batched(s, t), 10, B))
from ufm_amd import hip
hip.TIMER = hip.KernelTimer()
m.predict_correspondences_batched(s, t)
for k, v in sorted(hip.TIMER.summary().items(), key=lambda kv: -kv[1]["ms"]):
    print(f"   {k:30s} x{v['launches']:3d} {v['ms']:7.2f} ms")
hip.TIMER = None
