#!/usr/bin/env python3
"""Step time of the bench workload (UFM-Base, B=8, 518^2, two micro-batch streams) for same-box A/B of two builds:
   python tools/lab/step_time.py            (the in-tree library)
   OLD=1 python tools/lab/step_time.py      (after copying an older libufm_hip.so in place: unknown new symbols are skipped)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ufm_amd import hip
if os.environ.get("OLD") == "1":
    import ctypes
    probe = ctypes.CDLL(hip.LIB_PATH)
    for k in list(hip.SIGNATURES):
        if not hasattr(probe, k):
            hip.SIGNATURES.pop(k)
if os.environ.get("VARIANT"):
    hip.lib().ufm_debug_set_gemm_variant(int(os.environ["VARIANT"]))
if os.environ.get("ROWS"):
    hip.lib().ufm_debug_set_gemm_tile_rows(int(os.environ["ROWS"]))
import ufm_amd
from ufm_amd.modules import init_weights_
m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config()).eval()
init_weights_(m, 0)
m = m.to("cuda")
g = torch.Generator().manual_seed(1)
src = torch.randint(0, 256, (8, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
tgt = torch.randint(0, 256, (8, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
if os.environ.get("MB"):
    m.engine().micro_batches = int(os.environ["MB"])
if os.environ.get("CH"):
    m.engine().concurrent_heads = os.environ["CH"] == "1"
for _ in range(3):
    m.predict_correspondences_batched(src, tgt)
torch.cuda.synchronize()
ts = []
for _ in range(int(os.environ.get("STEPS", "15"))):
    t0 = time.perf_counter()
    m.predict_correspondences_batched(src, tgt)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print(f"{os.environ.get('TAG', 'lib')}: median {ts[len(ts)//2]:.2f} ms  min {ts[0]:.2f}  ({8e3/ts[len(ts)//2]:.1f} pairs/s)", flush=True)
