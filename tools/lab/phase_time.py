#!/usr/bin/env python3
"""Where a timed (two-stream, joint-heads) step spends its time: trunk (both micro-batch streams, until the join) vs heads."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ufm_amd
from ufm_amd.engine import Engine
from ufm_amd.modules import init_weights_

B = int(os.environ.get("B", "8"))
m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config()).eval()
init_weights_(m, 0)
m = m.to("cuda").set_numerics(os.environ.get("NUMERICS", "fast"))
g = torch.Generator().manual_seed(1)
src = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
tgt = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
marks = []
orig = Engine._heads_and_refine


def patched(self, *a, **k):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append(e)
    return orig(self, *a, **k)


Engine._heads_and_refine = patched
for _ in range(3):
    m.predict_correspondences_batched(src, tgt)
torch.cuda.synchronize()
rows = []
for _ in range(10):
    marks.clear()
    e0, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    m.predict_correspondences_batched(src, tgt)
    e2.record()
    torch.cuda.synchronize()
    if len(marks) == 1:
        rows.append((e0.elapsed_time(marks[0]), marks[0].elapsed_time(e2)))
rows.sort()
t, h = rows[len(rows) // 2]
print(f"B={B}: trunk (two streams, to the join) {t:.2f} ms, heads + un-map {h:.2f} ms, sum {t + h:.2f} ms")
