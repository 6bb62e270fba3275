#!/usr/bin/env python3
"""Golden vectors for the `ufm infer` post-processing.  RUN ONLY IN THE BUILD CONTAINER.

Imports the reference's own ``uniflowmatch/utils/viz.py`` (``cv2`` registered as an empty stub module: it is only
used by ``visualize_flow``, not by the function driven here) and stores inputs + outputs of ``warp_image_with_flow``
(viz.py:11-59) as ``viz_warp.npz``.  Data only; no reference source is copied."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.modules["cv2"] = types.ModuleType("cv2")
sys.path.insert(0, "/root/reference")
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_viz", "/root/reference/uniflowmatch/utils/viz.py")
ref_viz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_viz)

rng = np.random.default_rng(7)
cases = {}
for name, (H, W, Ht, Wt) in {"same": (40, 52, 40, 52), "differ": (33, 47, 41, 38)}.items():
    src = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    tgt = rng.integers(0, 256, (Ht, Wt, 3), dtype=np.uint8)
    flow = (rng.normal(size=(H, W, 2)) * 6).astype(np.float32)
    flow[:3] += 80.0   # far outside: exercises the clip to the last row/column
    flow[-2:] -= 90.0
    mask = (rng.random((H, W, 1)) > 0.3).astype(np.float32)
    cases[f"{name}_src"], cases[f"{name}_tgt"], cases[f"{name}_flow"], cases[f"{name}_mask"] = src, tgt, flow, mask
    cases[f"{name}_warped"] = ref_viz.warp_image_with_flow(src, None, tgt, flow).astype(np.float32)
    cases[f"{name}_warped_masked"] = ref_viz.warp_image_with_flow(src, mask, tgt, flow).astype(np.float32)
path = os.path.join(HERE, "viz_warp.npz")
np.savez_compressed(path, **cases)
print("wrote", path, os.path.getsize(path) // 1024, "KiB")
