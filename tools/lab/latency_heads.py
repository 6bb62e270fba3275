#!/usr/bin/env python3
"""One-pair (and two-pair) p50 latency, UFM-Base 518^2, hipGraph replay and eager: A/B of Engine.group_heads (the two DPT heads as
one grouped launch per layer vs two launch sequences on two streams) or, with --splitk, of Engine.conv_splitk; one process, interleaved."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ufm_amd
from ufm_amd.modules import init_weights_
m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config()).eval()
init_weights_(m, 0)
m = m.to("cuda")
g = torch.Generator().manual_seed(1)
def p50(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort(); return ts[len(ts) // 2]
for B in (1, 2, 3):
    src = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
    tgt = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
    outs = {}
    for rep in range(2):
        for gh in (True, False):
            if '--splitk' in sys.argv:
                m.engine().conv_splitk = gh
            else:
                m.engine().group_heads = gh
            outs[gh] = m.predict_correspondences_batched(src, tgt).flow.flow_output.clone()
            e = p50(lambda: m.predict_correspondences_batched(src, tgt))
            gp = ufm_amd.GraphedPredictor(m, src, tgt)
            r = p50(lambda: gp(src, tgt))
            print(f"B={B} {'conv_splitk' if '--splitk' in sys.argv else 'group_heads'}={int(gh)}: eager p50 {e:.2f} ms, graph replay p50 {r:.2f} ms", flush=True)
    print("bitwise equal:", torch.equal(outs[True], outs[False]), "max abs diff", (outs[True] - outs[False]).abs().max().item())
