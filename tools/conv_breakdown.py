#!/usr/bin/env python3
"""Per-launch time of every kernel in one UFM-Base step (B=8, 518^2), grouped by (kernel, work)."""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ufm_amd
from ufm_amd import hip
from ufm_amd.modules import init_weights_
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
RES = int(sys.argv[2]) if len(sys.argv) > 2 else 518
m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config(resolution_wh=(RES, RES))).eval()
init_weights_(m, 0)
m = m.to("cuda")
src = torch.randint(0, 256, (B, RES, RES, 3), dtype=torch.uint8).cuda()
tgt = torch.randint(0, 256, (B, RES, RES, 3), dtype=torch.uint8).cuda()
for _ in range(2):
    m.predict_correspondences_batched(src, tgt)
hip.TIMER = hip.KernelTimer()
m.predict_correspondences_batched(src, tgt)
torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0])
for name, e0, e1, meta in hip.TIMER.records:
    w = meta[0] if isinstance(meta, tuple) else meta  # GEMM metas are (flops, shape tag)
    k = (name, round(w / 1e9, 2) if w else 0)
    rows[k][0] += 1
    rows[k][1] += e0.elapsed_time(e1)
hip.TIMER = None
tot = sum(v[1] for v in rows.values())
print(f"total {tot:.2f} ms")
for (name, gw), (n, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
    rate = gw * n / ms if ms else 0
    print(f"{name:28s} work={gw:9.2f} G  x{n:3d}  {ms:7.3f} ms  ({ms/n*1e3:7.1f} us each)  {rate:8.1f} G/ms")
