#!/usr/bin/env python3
"""Staged repro of the hipGraph capture crash with nested stream forks (DPT level chains on side streams inside the two-stream
heads): eager first, then capture, printing a line per stage; run under `python -X faulthandler` so a segfault leaves a traceback."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ufm_amd
from ufm_amd.modules import init_weights_
cfg = ufm_amd.ufm_base_config() if "--base" in sys.argv else ufm_amd.configs.make_config(enc_dim=128, enc_depth=2, enc_heads=2, info_dim=128, info_depth=4, info_heads=2, layer_dims=(32, 64, 96, 128), feature_dim=256, resolution_wh=(56, 70), native_img_size=56)
H, W = (518, 518) if "--base" in sys.argv else (70, 56)
m = ufm_amd.UniFlowMatchConfidence(**cfg).eval()
init_weights_(m, 0)
m = m.to("cuda")
g = torch.Generator().manual_seed(1)
src = torch.randint(0, 256, (1, H, W, 3), dtype=torch.uint8, generator=g).cuda()
tgt = torch.randint(0, 256, (1, H, W, 3), dtype=torch.uint8, generator=g).cuda()
eng = m.engine()
eng.level_streams, eng.level_streams_max_images = False, 4
ref = m.predict_correspondences_batched(src, tgt).flow.flow_output.clone()
torch.cuda.synchronize(); print("stage 1: eager, level streams off: ok", flush=True)
eng.level_streams = True
if "--serial-heads" in sys.argv:
    eng.concurrent_heads = False
out = m.predict_correspondences_batched(src, tgt).flow.flow_output.clone()
torch.cuda.synchronize(); print("stage 2: eager, level streams on: ok, bitwise", torch.equal(out, ref), flush=True)
gp = ufm_amd.GraphedPredictor(m, src, tgt)
torch.cuda.synchronize(); print("stage 3: capture done", flush=True)
o = gp(src, tgt).flow.flow_output
torch.cuda.synchronize(); print("stage 4: replay ok, bitwise", torch.equal(o, ref), flush=True)
