"""Post-processing of the ``ufm infer`` runner (reference ``utils/viz.py`` and ``cli.py:85-156``).

* ``warp_image_with_flow`` -- same signature and semantics as the reference (viz.py:11-59): numpy in, numpy out;
  the resampling itself runs on the GPU (``ufm_warp_bilinear``), there is no CPU fallback.
* ``flow_to_color`` -- the reference calls the third-party ``flow_vis.flow_to_color`` (cli.py:128), which is not in
  this image and not in ``/root/reference``: this is a restatement of its published algorithm (Middlebury colour
  wheel of Baker et al., ICCV 2007; flow_vis 0.1) -- **parity unpinned**, checked only against the wheel's known
  anchor colours.  Host numpy: it formats a PNG, it is not on the hot path.
"""

from __future__ import annotations

import numpy as np
import torch

from . import hip


def warp_image_with_flow(source_image, source_mask, target_image, flow) -> np.ndarray:
    """viz.py:11-59.  source_image (H, W, 3) [only its shape is used], target_image (Ht, Wt, 3) uint8 or float,
    flow (H, W, 2) displacement source -> target, source_mask (H, W[, 1]) or None.  Returns float32 (H, W, 3)."""
    flow = np.asarray(flow)
    assert flow.shape[-1] == 2
    H, W = np.asarray(source_image).shape[:2]
    assert flow.shape[:2] == (H, W), "flow must live in the source frame"
    if not torch.cuda.is_available():
        raise RuntimeError("ufm_amd.viz.warp_image_with_flow runs on an AMD GPU only (no CPU fallback)")
    dev = torch.device("cuda", torch.cuda.current_device())
    tgt = torch.from_numpy(np.ascontiguousarray(target_image))
    tgt = (tgt if tgt.dtype == torch.uint8 else tgt.float()).to(dev).contiguous()
    fl = torch.from_numpy(np.ascontiguousarray(flow, dtype=np.float32)).to(dev).permute(2, 0, 1).contiguous()
    out = torch.empty((H, W, 3), device=dev, dtype=torch.float32)
    mask = None
    if source_mask is not None:
        mask = torch.from_numpy(np.ascontiguousarray(source_mask, dtype=np.float32)).to(dev).reshape(H, W).contiguous()
    hip.warp_bilinear(tgt, fl, out, mask=mask, mask_mode=1 if mask is not None else 0)
    return out.cpu().numpy()


def _color_wheel() -> np.ndarray:
    ry, yg, gc, cb, bm, mr = 15, 6, 4, 11, 13, 6
    wheel = np.zeros((ry + yg + gc + cb + bm + mr, 3))
    col = 0
    wheel[0:ry, 0], wheel[0:ry, 1] = 255, np.floor(255 * np.arange(ry) / ry)
    col += ry
    wheel[col : col + yg, 0], wheel[col : col + yg, 1] = 255 - np.floor(255 * np.arange(yg) / yg), 255
    col += yg
    wheel[col : col + gc, 1], wheel[col : col + gc, 2] = 255, np.floor(255 * np.arange(gc) / gc)
    col += gc
    wheel[col : col + cb, 1], wheel[col : col + cb, 2] = 255 - np.floor(255 * np.arange(cb) / cb), 255
    col += cb
    wheel[col : col + bm, 2], wheel[col : col + bm, 0] = 255, np.floor(255 * np.arange(bm) / bm)
    col += bm
    wheel[col : col + mr, 2], wheel[col : col + mr, 0] = 255 - np.floor(255 * np.arange(mr) / mr), 255
    return wheel


def flow_to_color(flow_uv: np.ndarray, clip_flow: float | None = None) -> np.ndarray:
    """(H, W, 2) flow -> (H, W, 3) uint8 RGB: hue = direction on the Middlebury wheel, saturation = magnitude relative
    to the largest magnitude in the field (restated flow_vis.flow_to_color; see the module docstring)."""
    flow_uv = np.asarray(flow_uv, dtype=np.float64)
    assert flow_uv.ndim == 3 and flow_uv.shape[2] == 2
    if clip_flow is not None:
        flow_uv = np.clip(flow_uv, 0, clip_flow)
    u, v = flow_uv[..., 0], flow_uv[..., 1]
    rad_max = np.max(np.sqrt(u * u + v * v))
    u, v = u / (rad_max + 1e-5), v / (rad_max + 1e-5)
    wheel = _color_wheel()
    ncols = wheel.shape[0]
    rad = np.sqrt(u * u + v * v)
    fk = (np.arctan2(-v, -u) / np.pi + 1) / 2 * (ncols - 1)
    k0 = np.floor(fk).astype(np.int32)
    k1 = k0 + 1
    k1[k1 == ncols] = 0
    f = fk - k0
    img = np.zeros(flow_uv.shape[:2] + (3,), dtype=np.uint8)
    for c in range(3):
        col = (1 - f) * wheel[k0, c] / 255.0 + f * wheel[k1, c] / 255.0
        inside = rad <= 1
        col[inside] = 1 - rad[inside] * (1 - col[inside])
        col[~inside] = col[~inside] * 0.75
        img[..., c] = np.floor(255 * col)
    return img


def save_png(path, array) -> None:
    from PIL import Image

    a = np.asarray(array)
    if a.dtype != np.uint8:
        # cv2.imwrite (cli.py:140-146) converts float images with saturate_cast: round to nearest, then clamp
        a = np.clip(np.rint(a), 0, 255).astype(np.uint8)
    Image.fromarray(a).save(str(path))


def load_rgb(path) -> np.ndarray:
    from PIL import Image

    return np.array(Image.open(str(path)).convert("RGB"))  # a writable copy (torch.from_numpy warns otherwise)
