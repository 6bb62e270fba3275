#!/usr/bin/env python3
"""One- / two-pair p50 latency (UFM-Base 518^2), eager and hipGraph replay: A/B of Engine.level_streams (the DPT heads' four level
chains on side streams / graph branches) in one process, interleaved; stderr of a failing capture is kept."""
import os, sys, time, traceback, torch
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
for B in (1, 2):
    src = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
    tgt = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
    outs = {}
    for rep in range(2):
        for on in (True, False):
            m.engine().level_streams = on
            m.engine().level_streams_max_images = 4
            outs[on] = m.predict_correspondences_batched(src, tgt).flow.flow_output.clone()
            e = p50(lambda: m.predict_correspondences_batched(src, tgt))
            try:
                gp = ufm_amd.GraphedPredictor(m, src, tgt)
                r = p50(lambda: gp(src, tgt))
                same = torch.equal(gp(src, tgt).flow.flow_output, outs[on])
                print(f"B={B} level_streams={int(on)}: eager p50 {e:.2f} ms, graph replay p50 {r:.2f} ms (graph == eager: {same})", flush=True)
            except Exception:  # noqa: BLE001
                print(f"B={B} level_streams={int(on)}: eager p50 {e:.2f} ms, GRAPH CAPTURE FAILED:", flush=True)
                traceback.print_exc()
    print("bitwise equal on/off:", torch.equal(outs[True], outs[False]))
