#!/usr/bin/env python3
"""Is predict_correspondences_batched bitwise independent of the batch a pair is in?  (UFM-Base 518^2, B = 8 vs 1)"""
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
for mb in (2, 1):
    m.engine().micro_batches = mb
    whole = m.predict_correspondences_batched(src, tgt)
    wf, wm = whole.flow.flow_output.clone(), whole.covisibility.mask.clone()
    again = m.predict_correspondences_batched(src, tgt)
    print("mb", mb, "repeat equal:", torch.equal(wf, again.flow.flow_output), torch.equal(wm, again.covisibility.mask))
    for i in (0, 3, 4, 7):
        one = m.predict_correspondences_batched(src[i:i+1], tgt[i:i+1])
        df = (one.flow.flow_output - wf[i:i+1]).abs().max().item()
        dm = (one.covisibility.mask - wm[i:i+1]).abs().max().item()
        print("  pair", i, "flow diff", df, "mask diff", dm)
