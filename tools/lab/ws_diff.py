#!/usr/bin/env python3
"""Which workspace buffers differ between two identical B=8 two-stream runs?  (localises a nondeterminism)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ufm_amd
from ufm_amd.modules import init_weights_
m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config()).eval()
init_weights_(m, 0)
m = m.to("cuda")
g = torch.Generator().manual_seed(1234)
src = torch.randint(0, 256, (8, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
tgt = torch.randint(0, 256, (8, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
eng = m.engine()
eng.micro_batches = int(os.environ.get("MB", "2"))
eng.concurrent_heads = os.environ.get("CH", "1") == "1"
def snap():
    torch.cuda.synchronize()
    d = {}
    for attr in ("_bufs",):
        ws = getattr(eng, attr, None)
        if isinstance(ws, dict):
            for k, v in ws.items():
                if isinstance(v, torch.Tensor):
                    d[attr + ":" + str(k)] = v.clone()
    return d
m.predict_correspondences_batched(src, tgt)
base = snap()
print("buffers:", len(base))
for rep in range(6):
    o = m.predict_correspondences_batched(src, tgt)
    cur = snap()
    bad = []
    for k, v in cur.items():
        b = base.get(k)
        if b is None or b.shape != v.shape:
            continue
        if not torch.equal(v.view(torch.uint8), b.view(torch.uint8)):
            x, y = v.float().flatten(), b.float().flatten()
            nz = (x != y).nonzero().flatten()
            bad.append((k, tuple(v.shape), nz.numel(), nz[0].item(), nz[-1].item(), (x - y).abs().max().item()))
    print("rep", rep, "differing buffers:", len(bad))
    for r in bad:
        print("   ", r)
