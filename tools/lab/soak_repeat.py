#!/usr/bin/env python3
"""Race soak: N identical two-stream forwards (UFM-Base B=8 and UFM-Refine+UNet B=4, 518^2) must all be bitwise equal."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ufm_amd
from ufm_amd.modules import init_weights_
REPS = int(os.environ.get("REPS", "60"))
def soak(model, B, tag):
    g = torch.Generator().manual_seed(7)
    src = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
    tgt = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
    o = model.predict_correspondences_batched(src, tgt)
    f0, m0 = o.flow.flow_output.clone(), o.covisibility.mask.clone()
    bad = 0
    for r in range(REPS):
        o = model.predict_correspondences_batched(src, tgt)
        if not (torch.equal(o.flow.flow_output, f0) and torch.equal(o.covisibility.mask, m0)):
            bad += 1
            print(f"  {tag}: rep {r} differs: flow {(o.flow.flow_output - f0).abs().max().item():.3g}", flush=True)
    print(f"{tag}: {bad} of {REPS} repeats differ", flush=True)
m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config()).eval(); init_weights_(m, 0); m = m.to("cuda")
soak(m, 8, "UFM-Base B=8 fast")
soak(m, 5, "UFM-Base B=5 fast")
del m
cfg = ufm_amd.ufm_refine_config(use_unet_feature=True) if hasattr(ufm_amd, "ufm_refine_config") else None
if cfg is not None:
    r = ufm_amd.UniFlowMatchClassificationRefinement(**cfg).eval(); init_weights_(r, 0); r = r.to("cuda")
    soak(r, 4, "UFM-Refine+UNet B=4 fast")
    del r
r = ufm_amd.UniFlowMatchClassificationRefinement(**ufm_amd.ufm_refine_config()).eval(); init_weights_(r, 0); r = r.to("cuda")
soak(r, 8, "UFM-Refine B=8 fast (joint heads)")
del r
m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config()).eval(); init_weights_(m, 0); m = m.to("cuda").set_numerics("precise")
soak(m, 8, "UFM-Base B=8 precise")
# round 4: the options that are off by default -- grouped head launches, split-K (agent-scope arrival counters), level-chain streams
del m
m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config()).eval(); init_weights_(m, 0); m = m.to("cuda")
e = m.engine()
e.group_heads, e.conv_splitk = True, True
soak(m, 8, "UFM-Base B=8 fast, grouped heads + split-K")
e.group_heads = False
soak(m, 3, "UFM-Base B=3 fast, split-K, two-stream heads")
e.conv_splitk, e.level_streams, e.level_streams_max_images = False, True, 8
soak(m, 2, "UFM-Base B=2 fast, level-chain streams")
# round 5: the 256x128 two-resident-workgroups GEMM on every eligible Linear (gemm flags bit 27), and precise mode on the rebuilt attention kernel
e.level_streams = False
from ufm_amd import hip
hip.lib().ufm_debug_set_gemm_flags(1 << 27)
soak(m, 8, "UFM-Base B=8 fast, pair GEMM everywhere")
hip.lib().ufm_debug_set_gemm_flags(0)
del m
m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config(resolution_wh=(1036, 1036))).eval(); init_weights_(m, 0); m = m.to("cuda").set_numerics("precise")
def soak_big(model, tag):
    g = torch.Generator().manual_seed(9)
    src = torch.randint(0, 256, (1, 1036, 1036, 3), dtype=torch.uint8, generator=g).cuda()
    tgt = torch.randint(0, 256, (1, 1036, 1036, 3), dtype=torch.uint8, generator=g).cuda()
    o = model.predict_correspondences_batched(src, tgt)
    f0 = o.flow.flow_output.clone()
    bad = sum(0 if torch.equal(model.predict_correspondences_batched(src, tgt).flow.flow_output, f0) else 1 for _ in range(max(4, REPS // 10)))
    print(f"{tag}: {bad} of {max(4, REPS // 10)} repeats differ", flush=True)
soak_big(m, "UFM-Base 1036^2 B=1 precise (10 954-token joint attention on the round-5 kernel)")
# round 6: the row-window halo convolution (counted waits of its own, buffer-load zero padding) on NON-square maps with tiles across image boundaries --
# the reference's class-default resolution (420 x 560: 120 x 160 / 60 x 80 head maps), 1080 x 810 inputs through the GPU antialias resize; the
# precise soak above already runs the interleaved bf16x3 Linear layers and the fixed-reference attention
del m
cfg = ufm_amd.ufm_base_config()
cfg.pop("inference_resolution")
m = ufm_amd.UniFlowMatchConfidence(**cfg).eval(); init_weights_(m, 0); m = m.to("cuda")
def soak_default(model, B, tag):
    g = torch.Generator().manual_seed(11)
    src = torch.randint(0, 256, (B, 810, 1080, 3), dtype=torch.uint8, generator=g).cuda()
    tgt = torch.randint(0, 256, (B, 810, 1080, 3), dtype=torch.uint8, generator=g).cuda()
    o = model.predict_correspondences_batched(src, tgt)
    f0 = o.flow.flow_output.clone()
    bad = sum(0 if torch.equal(model.predict_correspondences_batched(src, tgt).flow.flow_output, f0) else 1 for _ in range(REPS))
    print(f"{tag}: {bad} of {REPS} repeats differ", flush=True)
soak_default(m, 8, "UFM-Base class-default 420x560 B=8 fast (halo convolution on 120x160 / 60x80 maps)")
soak_default(m, 3, "UFM-Base class-default 420x560 B=3 fast")
