#!/usr/bin/env python3
"""Interleaved same-process A/B of Engine attributes on the bench workload (UFM-Base, B=8, 518^2), rule 24 of the
programming guide: N variants x M rounds in ONE process, median and min per variant.
   python tools/lab/ab_engine.py defer_residual=0 defer_residual=1
   NUMERICS=precise MB=1 python tools/lab/ab_engine.py x=0 ...
A variant is a comma-separated list of attr=value pairs set on model.engine() (ints are cast; 'lib:NAME=V' calls
hip.lib().ufm_debug_set_NAME(V) instead)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ufm_amd
from ufm_amd import hip
from ufm_amd.modules import init_weights_

B = int(os.environ.get("B", "8"))
m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config()).eval()
init_weights_(m, 0)
m = m.to("cuda").set_numerics(os.environ.get("NUMERICS", "fast"))
g = torch.Generator().manual_seed(1)
src = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
tgt = torch.randint(0, 256, (B, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
if os.environ.get("MB"):
    m.engine().micro_batches = int(os.environ["MB"])
variants = sys.argv[1:] or ["defer_residual=0", "defer_residual=1"]


def apply(v):
    for kv in v.split(","):
        k, val = kv.split("=")
        if k.startswith("lib:"):
            getattr(hip.lib(), "ufm_debug_set_" + k[4:])(int(val))
        else:
            setattr(m.engine(), k, int(val) if val.lstrip("-").isdigit() else val)


times = {v: [] for v in variants}
rounds, per = int(os.environ.get("ROUNDS", "6")), int(os.environ.get("STEPS", "5"))
for v in variants:  # warm every variant's buffers
    apply(v)
    for _ in range(2):
        m.predict_correspondences_batched(src, tgt)
torch.cuda.synchronize()
for r in range(rounds):
    for v in variants:
        apply(v)
        m.predict_correspondences_batched(src, tgt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(per):
            m.predict_correspondences_batched(src, tgt)
        torch.cuda.synchronize()
        times[v].append((time.perf_counter() - t0) * 1e3 / per)
for v in variants:
    ts = sorted(times[v])
    print(f"{v:40s} median {ts[len(ts)//2]:7.2f} ms  min {ts[0]:7.2f}  ({B*1e3/ts[len(ts)//2]:.1f} pairs/s)", flush=True)
