#!/usr/bin/env python3
"""Every head-format convolution launch of one UFM-Base step (B pairs, 518^2) in call order: shape tag, us, fraction of the bf16 peak / 3.
   python tools/lab/head_layers.py [B]      (serial: the event bracket of hip.TIMER turns the streams off)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ufm_amd
from ufm_amd import hip
from ufm_amd.modules import init_weights_

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config()).eval()
init_weights_(m, 0)
m = m.to("cuda")
if os.environ.get("CONV_VARIANT"):
    hip.lib().ufm_debug_set_conv_variant(int(os.environ["CONV_VARIANT"]))
src = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8).cuda()
tgt = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8).cuda()
for _ in range(2):
    m.predict_correspondences_batched(src, tgt)
acc = {}
order = []
REP = 5
for _ in range(REP):
    hip.TIMER = hip.KernelTimer()
    m.predict_correspondences_batched(src, tgt)
    torch.cuda.synchronize()
    recs = hip.TIMER.records
    hip.TIMER = None
    idx = 0
    for name, e0, e1, meta in recs:
        if not (name.startswith("ufm_conv2d") or name.startswith("ufm_upsample") or name.startswith("ufm_dpt_tail")):
            continue
        key = (idx, name, meta[1] if isinstance(meta, tuple) and len(meta) > 1 else "", meta[0] if isinstance(meta, tuple) else (meta or 0))
        acc.setdefault(key, []).append(e0.elapsed_time(e1) * 1e3)
        idx += 1
tot = 0.0
for (idx, name, tag, work), v in sorted(acc.items()):
    us = sorted(v)[len(v) // 2]
    tot += us
    frac = work / (us * 1e-6) / (2.5e15 / 3) if work and "conv2d" in name else 0.0
    print(f"{idx:3d} {name[4:]:26s} {str(tag):34s} {us:8.1f} us  {work / 1e9:8.2f} GF  frac/3 {frac:5.3f}")
print(f"total {tot / 1e3:.3f} ms")
