import collections, os, sys, torch
sys.path.insert(0, "/root/repo")
import ufm_amd
from ufm_amd import hip
from ufm_amd.modules import init_weights_
m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config()).eval()
init_weights_(m, 0); m = m.to("cuda")
src = torch.randint(0, 256, (8, 518, 518, 3), dtype=torch.uint8).cuda(); tgt = torch.randint(0, 256, (8, 518, 518, 3), dtype=torch.uint8).cuda()
for _ in range(2): m.predict_correspondences_batched(src, tgt)
for flags in (0, 1 << 28, 7 << 24, (1 << 28) | (7 << 24), 0):
    hip.lib().ufm_debug_set_gemm_flags(flags)
    acc = collections.defaultdict(list)
    for rep in range(3):
        hip.TIMER = hip.KernelTimer()
        m.predict_correspondences_batched(src, tgt); torch.cuda.synchronize()
        recs = hip.TIMER.records; hip.TIMER = None
        for name, e0, e1, meta in recs:
            if name == "ufm_gemm_bf16": acc[meta[1]].append(e0.elapsed_time(e1) * 1e3)
    print(f"flags {flags:#x}: " + "  ".join(f"{k.split(' ',1)[1][:28]}: {sorted(v)[len(v)//2]:.1f}" for k, v in acc.items() if k.startswith("M21920")), flush=True)
hip.lib().ufm_debug_set_gemm_flags(0)
# per-call values of one step (call order): a mean far above the median comes from a few calls
hip.TIMER = hip.KernelTimer()
m.predict_correspondences_batched(src, tgt); torch.cuda.synchronize()
recs = hip.TIMER.records; hip.TIMER = None
prev = None
for name, e0, e1, meta in recs:
    if name == "ufm_gemm_bf16" and meta[1].startswith("M21920 N3072"):
        print(f"QKV {e0.elapsed_time(e1) * 1e3:7.1f} us   (previous launch: {prev})")
    prev = name
