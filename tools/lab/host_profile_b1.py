#!/usr/bin/env python3
"""Host-side profile of one-pair forwards (cProfile, 30 calls): where the Python side of predict_correspondences_batched spends
its time before / between the ~700 kernel launches."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ufm_amd
from ufm_amd.modules import init_weights_
m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_base_config()).eval()
init_weights_(m, 0)
m = m.to("cuda")
g = torch.Generator().manual_seed(1)
src = torch.randint(0, 256, (1, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
tgt = torch.randint(0, 256, (1, 518, 518, 3), dtype=torch.uint8, generator=g).cuda()
for _ in range(5):
    m.predict_correspondences_batched(src, tgt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    m.predict_correspondences_batched(src, tgt)
t_host = (time.perf_counter() - t0) / 30 * 1e3
torch.cuda.synchronize()
print(f"host time to ENQUEUE one forward: {t_host:.2f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    m.predict_correspondences_batched(src, tgt)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)

# sanity of ufm_amd.hip._stream(): the raw accessor returns the handle of torch's current stream, also inside a stream context
from ufm_amd import hip as _hip
assert _hip._stream() == torch.cuda.current_stream().cuda_stream
_s = torch.cuda.Stream()
with torch.cuda.stream(_s):
    assert _hip._stream() == _s.cuda_stream == torch.cuda.current_stream().cuda_stream
print("raw stream accessor agrees with torch.cuda.current_stream()")
