#!/usr/bin/env python3
"""Bitwise equality of the UFM-Base B=8 outputs under two engine settings: python ab_bitwise.py joint_info=0 joint_info=1"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ufm_amd
from ufm_amd.modules import init_weights_
m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config()).eval(); init_weights_(m, 0); m = m.to("cuda")
g = torch.Generator().manual_seed(1)
B = int(os.environ.get("B", "8"))
src = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
tgt = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
outs = []
for v in sys.argv[1:3]:
    for kv in v.split(","):
        k, val = kv.split("="); setattr(m.engine(), k, int(val))
    o = m.predict_correspondences_batched(src, tgt)
    outs.append((o.flow.flow_output.clone(), o.covisibility.mask.clone()))
print("bitwise equal:", all(torch.equal(a, b) for a, b in zip(*outs)))
