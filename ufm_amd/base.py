"""Public model API base: mirrors ``uniflowmatch/models/base.py`` (same names, argument meaning,
error behaviour) with every tensor operation executed by HIP kernels.

Reference map
  UFMFlowFieldOutput / UFMMaskFieldOutput / UFMClassificationRefinementOutput / UFMOutputInterface
      <- models/base.py:11-72
  UniFlowMatchModelsBase.__init__                       <- models/base.py:86-100
  predict_correspondences_batched                       <- models/base.py:137-234
  _predict_correspondences_batched                      <- models/base.py:236-334
  resolution selection / region bookkeeping             <- utils/flow_resizing.py:276-354, 667-744
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import hip
from .modules import IMAGE_NORMALIZATION


@dataclass
class UFMFlowFieldOutput:
    flow_output: torch.Tensor
    flow_covariance: Optional[torch.Tensor] = None
    flow_covariance_inv: Optional[torch.Tensor] = None
    flow_covariance_log_det: Optional[torch.Tensor] = None


@dataclass
class UFMMaskFieldOutput:
    mask: torch.Tensor
    logits: torch.Tensor


@dataclass
class UFMClassificationRefinementOutput:
    regression_flow_output: torch.Tensor
    residual: torch.Tensor
    log_softmax: torch.Tensor
    feature_map_0: torch.Tensor
    feature_map_1: torch.Tensor


@dataclass
class UFMOutputInterface:
    flow: Optional[UFMFlowFieldOutput] = None
    classification_refinement: Optional[UFMClassificationRefinementOutput] = None
    covisibility: Optional[UFMMaskFieldOutput] = None


def closest_aspect_resolution(resolutions_wh: Sequence[Tuple[int, int]], h0: int, w0: int, h1: int, w1: int) -> Tuple[int, int]:
    """(H, W) of the trained resolution whose aspect is closest to both inputs
    (flow_resizing.py:676-692; every candidate is a fixed-size resize so all are runnable)."""
    cands = [(int(r[1]), int(r[0])) for r in resolutions_wh]
    if len(cands) == 0:
        raise ValueError("No valid shape found for the given resolution.")
    return min(cands, key=lambda s: abs(s[0] / s[1] - h0 / w0) + abs(s[0] / s[1] - h1 / w1))


def representation_region(target_hw: Tuple[int, int], h: int, w: int) -> List[int]:
    """[0,h,0,w] scaled into the network frame: float32 multiply, truncate to int (flow_resizing.py:332-345)."""
    th, tw = target_hw
    mh, mw = np.float32(th / h), np.float32(tw / w)
    return [0, int(np.float32(mh * np.float32(h))), 0, int(np.float32(mw * np.float32(w)))]


class UniFlowMatchModelsBase(torch.nn.Module):
    def __init__(self, inference_resolution: Optional[Union[List[Tuple[int, int]], Tuple[int, int]]] = None):
        super().__init__()
        if inference_resolution is None:
            inference_resolution = [(560, 420)]
        if isinstance(inference_resolution[0], int):  # single (W, H)
            inference_resolution = [inference_resolution]
        self.inference_resolution = [tuple(int(v) for v in r) for r in inference_resolution]
        self._engine = None
        self.numerics = "fast"

    # --- engine plumbing -------------------------------------------------------------------
    def set_numerics(self, numerics: str) -> "UniFlowMatchModelsBase":
        """"fast": bf16 MFMA trunk (the reference's GPU autocast policy) + bf16x3 split-precision heads;
        "precise": bf16x3 split precision for the trunk too (fp32-class accuracy on the bf16 matrix cores: the mode that
        meets the 1e-3 px tolerance without the exact-fp32 MFMA's 1/16 rate);
        "parity": fp32 MFMA everywhere; "parity_x3heads": fp32 MFMA trunk + the bf16x3 heads of "fast"
        (isolates the one choice of "fast" that is narrower than the reference's fp32 head island, ufm.py:635)."""
        if numerics not in ("fast", "precise", "parity", "parity_x3heads"):
            raise ValueError("numerics must be 'fast', 'precise', 'parity' or 'parity_x3heads'")
        if numerics != self.numerics:
            self.numerics, self._engine = numerics, None
        return self

    def engine(self):
        from .engine import Engine

        if self._engine is None:
            self._engine = Engine(self, self.numerics)
        return self._engine

    def forward(self, view1, view2) -> UFMOutputInterface:
        raise NotImplementedError("Implement this method in derived classes")

    def get_parameter_groups(self) -> Dict[str, torch.nn.ParameterList]:
        raise NotImplementedError("Implement this method in derived classes")

    # --- public entry point ----------------------------------------------------------------
    def predict_correspondences_batched(
        self,
        source_image: torch.Tensor,
        target_image: torch.Tensor,
        data_norm_type: Optional[str] = None,
    ) -> UFMOutputInterface:
        """Dense correspondences source -> target.

        source_image / target_image: BCHW / BHWC / CHW / HWC, uint8 (auto-normalised) or float32
        (then ``data_norm_type`` is mandatory).  Returns flow (B,2,H,W) in source pixels and
        covisibility (B,H,W) in [0,1]; unbatched inputs come back with a batch dim of 1, as in the
        reference (base.py:166-171 never squeezes).
        """
        assert isinstance(source_image, torch.Tensor) and isinstance(
            target_image, torch.Tensor
        ), "source_image and target_image must be torch.Tensors"
        assert source_image.dim() in [3, 4], "source_image must have dimensions 3 or 4"
        assert target_image.dim() in [3, 4], "target_image must have dimensions 3 or 4"
        if source_image.dim() == 3:
            source_image = source_image.unsqueeze(0)
            target_image = target_image.unsqueeze(0)
        # channel layout: BCHW wins when both readings are possible (base.py:174-181)
        if source_image.shape[1] == 3 and target_image.shape[1] == 3:
            layout = 1
            hs, ws, ht, wt = source_image.shape[2], source_image.shape[3], target_image.shape[2], target_image.shape[3]
        elif source_image.shape[-1] == 3 and target_image.shape[-1] == 3:
            layout = 0
            hs, ws, ht, wt = source_image.shape[1], source_image.shape[2], target_image.shape[1], target_image.shape[2]
        else:
            raise ValueError("source_image and target_image must have 3 channels in either BCHW or BHWC format")

        required = self.encoder.data_norm_type
        mean, std = IMAGE_NORMALIZATION[required]["mean"], IMAGE_NORMALIZATION[required]["std"]
        if source_image.dtype == torch.float32:
            assert data_norm_type is not None, "data_norm_type must be provided for float32 images"
            assert data_norm_type in IMAGE_NORMALIZATION, f"data_norm_type must be one of {list(IMAGE_NORMALIZATION.keys())}"
            assert target_image.dtype == torch.float32, "Image types must match"
            if data_norm_type != required:  # x*(ps/s) + (pm-m)/s in float32, base.py:212-213
                pm, ps = IMAGE_NORMALIZATION[data_norm_type]["mean"], IMAGE_NORMALIZATION[data_norm_type]["std"]
                scale3 = [float(np.float32(ps[c]) / np.float32(std[c])) for c in range(3)]
                shift3 = [float((np.float32(pm[c]) - np.float32(mean[c])) / np.float32(std[c])) for c in range(3)]
            else:
                scale3, shift3 = [1.0, 1.0, 1.0], [0.0, 0.0, 0.0]
        elif source_image.dtype == torch.uint8:
            assert target_image.dtype == torch.uint8, "Image types must match"
            scale3, shift3 = list(std), list(mean)  # kernel evaluates (x/255 - mean)/std, base.py:228-229
        else:
            raise ValueError("source_image and target_image must be of type torch.float32 or torch.uint8")
        return self._predict(source_image, target_image, layout, scale3, shift3, (hs, ws), (ht, wt))

    def _predict(self, src, tgt, layout, scale3, shift3, src_hw, tgt_hw) -> UFMOutputInterface:
        """base.py:236-334: resize -> forward -> un-map, all on device."""
        if not (src.is_cuda and tgt.is_cuda):
            raise RuntimeError("ufm_amd runs on an AMD GPU only: pass device tensors (there is no CPU fallback)")
        hs, ws = int(src_hw[0]), int(src_hw[1])
        ht, wt = int(tgt_hw[0]), int(tgt_hw[1])
        H, W = closest_aspect_resolution(self.inference_resolution, hs, ws, ht, wt)
        reg0_src, reg1_src = [0, hs, 0, ws], [0, ht, 0, wt]
        reg0_rep = representation_region((H, W), hs, ws)
        with torch.cuda.device(src.device):
            res = self._forward_device(src.contiguous(), tgt.contiguous(), layout, scale3, shift3, H, W, hs, ws, ht, wt)
            B = src.shape[0]
            out = UFMOutputInterface()
            flow = res.flow.flow_output.contiguous()
            flow_un = torch.empty((B, 2, hs, ws), device=src.device, dtype=torch.float32)
            hip.unmap_flow(flow, B, H, W, reg0_rep, reg0_src, reg1_src, hs, ws, flow_un)
            out.flow = UFMFlowFieldOutput(flow_output=flow_un)
            if res.flow.flow_covariance is not None:  # base.py:295-319: un-map, then [wr^2, hr^2, wr*hr]
                cov = res.flow.flow_covariance.contiguous()
                cov_un = torch.empty((B, 3, hs, ws), device=src.device, dtype=torch.float32)
                wr, hr = ws / W, hs / H
                hip.unmap_channels(cov, B, 3, H, W, reg0_rep, reg0_src, hs, ws, cov_un, chan_scale=[wr**2, hr**2, wr * hr])
                out.flow.flow_covariance = cov_un
            if res.covisibility is not None:
                m = res.covisibility.mask.contiguous()
                m_un = torch.empty((B, 1, hs, ws), device=src.device, dtype=torch.float32)
                hip.unmap_channels(m, B, 1, H, W, reg0_rep, reg0_src, hs, ws, m_un)
                out.covisibility = UFMMaskFieldOutput(mask=m_un.squeeze(1), logits=None)  # base.py:331-332
        return out

    def _forward_device(self, src, tgt, layout, scale3, shift3, H, W, hs, ws, ht, wt) -> UFMOutputInterface:
        raise NotImplementedError
