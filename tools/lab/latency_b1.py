#!/usr/bin/env python3
"""One-pair p50 latency (UFM-Base 518^2), eager / hipGraph replay; A/B of the two-K-tiles-per-barrier 128x128 GEMM
(ufm_debug_set_gemm_flags(128) = off) in one process."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ufm_amd
from ufm_amd import hip
from ufm_amd.modules import init_weights_
m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config()).eval()
init_weights_(m, 0)
m = m.to("cuda")
g = torch.Generator().manual_seed(1)
src = torch.randint(0, 256, (1, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
tgt = torch.randint(0, 256, (1, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
def p50(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
outs = {}
for flag in (128, 0, 128, 0):
    hip.lib().ufm_debug_set_gemm_flags(flag)
    o = m.predict_correspondences_batched(src, tgt)
    outs[flag] = o.flow.flow_output.clone()
    e = p50(lambda: m.predict_correspondences_batched(src, tgt))
    gp = ufm_amd.GraphedPredictor(m, src, tgt)
    r = p50(lambda: gp(src, tgt))
    print(f"two K-tiles per barrier {'off' if flag else 'on '}: eager p50 {e:.2f} ms, graph replay p50 {r:.2f} ms", flush=True)
hip.lib().ufm_debug_set_gemm_flags(0)
print("bitwise equal:", torch.equal(outs[0], outs[128]))
