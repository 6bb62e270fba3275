"""fp32 CPU restatement of the reference's own hot-path code (glue + wiring).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Parity for this file is
PINNED: ``tests/test_oracle_vs_reference_goldens.py`` checks it against golden
vectors produced by the reference's own functions run in the build container
(``tests/golden/make_goldens.py``).

Each function cites the reference lines it follows (paths relative to
``/root/reference/uniflowmatch``).  Region 4-vectors are
``[top, bottom, left, right]`` Python ints.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn

from . import uniception_ref as U

# --------------------------------------------------------------------------- #
# output containers (models/base.py:11-72)
# --------------------------------------------------------------------------- #


@dataclass
class FlowOut:
    flow_output: torch.Tensor
    flow_covariance: Optional[torch.Tensor] = None
    flow_covariance_inv: Optional[torch.Tensor] = None
    flow_covariance_log_det: Optional[torch.Tensor] = None


@dataclass
class MaskOut:
    mask: torch.Tensor
    logits: Optional[torch.Tensor]


@dataclass
class RefineOut:
    regression_flow_output: torch.Tensor
    residual: torch.Tensor
    log_softmax: torch.Tensor
    feature_map_0: torch.Tensor
    feature_map_1: torch.Tensor


@dataclass
class Out:
    flow: Optional[FlowOut] = None
    classification_refinement: Optional[RefineOut] = None
    covisibility: Optional[MaskOut] = None


# --------------------------------------------------------------------------- #
# pre-processing: models/base.py:137-234
# --------------------------------------------------------------------------- #


def to_bchw_normalised(
    source: torch.Tensor, target: torch.Tensor, required_norm: str, data_norm_type: Optional[str]
) -> Tuple[torch.Tensor, torch.Tensor]:
    """base.py:160-231: dims, layout, dtype and normalisation handling."""
    assert isinstance(source, torch.Tensor) and isinstance(target, torch.Tensor)
    assert source.dim() in (3, 4) and target.dim() in (3, 4)
    if source.dim() == 3:  # base.py:166-171 (batch dim is never squeezed back)
        source, target = source.unsqueeze(0), target.unsqueeze(0)
    if source.shape[1] == 3 and target.shape[1] == 3:  # base.py:174-175: BCHW wins ties
        pass
    elif source.shape[-1] == 3 and target.shape[-1] == 3:  # base.py:176-179
        source, target = source.permute(0, 3, 1, 2), target.permute(0, 3, 1, 2)
    else:
        raise ValueError("source_image and target_image must have 3 channels in either BCHW or BHWC format")
    table = U.IMAGE_NORMALIZATION_DICT
    mean = table[required_norm].mean.view(1, 3, 1, 1)
    std = table[required_norm].std.view(1, 3, 1, 1)
    if source.dtype == torch.float32:  # base.py:187-213
        assert data_norm_type is not None, "data_norm_type must be provided for float32 images"
        assert data_norm_type in table
        if data_norm_type != required_norm:
            pm = table[data_norm_type].mean.view(1, 3, 1, 1)
            ps = table[data_norm_type].std.view(1, 3, 1, 1)
            source = source * (ps / std) + (pm - mean) / std
            target = target * (ps / std) + (pm - mean) / std
    elif source.dtype == torch.uint8:  # base.py:215-229
        source = (source.float() / 255.0 - mean) / std
        target = (target.float() / 255.0 - mean) / std
    else:
        raise ValueError("source_image and target_image must be of type torch.float32 or torch.uint8")
    return source, target


def select_resolution(resolutions_wh: Sequence[Tuple[int, int]], h0: int, w0: int, h1: int, w1: int) -> Tuple[int, int]:
    """utils/flow_resizing.py:667-694 (pairs branch; the single-image branch is dead code)
    with base.py:97-100: each (W, H) resolution is a fixed-size target (H, W) for both views."""
    cands = [(r[1], r[0]) for r in resolutions_wh]
    if not cands:
        raise ValueError("No valid shape found for the given resolution.")
    return min(cands, key=lambda s: abs(s[0] / s[1] - h0 / w0) + abs(s[0] / s[1] - h1 / w1))


def resize_pair(
    img0_bchw: torch.Tensor, img1_bchw: torch.Tensor, target_hw: Tuple[int, int]
) -> Tuple[torch.Tensor, torch.Tensor, List[int], List[int], List[int], List[int]]:
    """flow_resizing.py:276-354 + :724-744 on float BCHW input (base.py:263-266 permutes cancel)."""
    th, tw = target_hw
    _, _, h0, w0 = img0_bchw.shape
    _, _, h1, w1 = img1_bchw.shape
    r0 = F.interpolate(img0_bchw.float(), size=(th, tw), mode="bilinear", align_corners=False, antialias=True)
    r1 = F.interpolate(img1_bchw.float(), size=(th, tw), mode="bilinear", align_corners=False, antialias=True)
    reg0_src, reg1_src = [0, h0, 0, w0], [0, h1, 0, w1]
    # flow_resizing.py:332-345: float32 multiplier * int64 region, truncated to int64
    m0 = torch.tensor([th / h0, th / h0, tw / w0, tw / w0])
    m1 = torch.tensor([th / h1, th / h1, tw / w1, tw / w1])
    reg0_rep = (m0 * torch.tensor(reg0_src)).to(torch.int64).tolist()
    reg1_rep = (m1 * torch.tensor(reg1_src)).to(torch.int64).tolist()
    return r0, r1, reg0_src, reg1_src, reg0_rep, reg1_rep


# --------------------------------------------------------------------------- #
# post-processing: utils/flow_resizing.py:749-877, :955-1010
# --------------------------------------------------------------------------- #


def unmap_flow(
    flow: torch.Tensor,
    reg0_rep: Sequence[int],
    reg0_src: Sequence[int],
    reg1_src: Sequence[int],
    src_shape_hw: Tuple[int, int],
) -> Tuple[torch.Tensor, torch.Tensor]:
    """flow_resizing.py:749-877 (img1_region_representation and img1_source_shape are unused there)."""
    b = flow.shape[0]
    roi = flow[..., reg0_rep[0] : reg0_rep[1], reg0_rep[2] : reg0_rep[3]]
    rh, rw = roi.shape[2], roi.shape[3]
    xs = torch.arange(0, rw) + 0.5
    ys = torch.arange(0, rh) + 0.5
    gx, gy = torch.meshgrid(xs, ys, indexing="xy")
    src_coords = torch.stack((gx, gy), dim=0).unsqueeze(0).float().to(flow.device)  # 1,2,rh,rw (x, y)
    sh, sw = reg0_src[1] - reg0_src[0], reg0_src[3] - reg0_src[2]
    th, tw = reg1_src[1] - reg1_src[0], reg1_src[3] - reg1_src[2]
    src_valid = F.interpolate(src_coords, size=[sh, sw], mode="bilinear", align_corners=False)
    tgt_valid = F.interpolate(roi.float(), size=[sh, sw], mode="nearest") + src_valid
    rep_w, rep_h = reg0_rep[3] - reg0_rep[2], reg0_rep[1] - reg0_rep[0]
    # flow_resizing.py:832-853: ratios are formed from int64 tensors -> float32 true division
    s_scale = torch.tensor([torch.tensor(sw) / torch.tensor(rep_w), torch.tensor(sh) / torch.tensor(rep_h)])
    t_scale = torch.tensor([torch.tensor(tw) / torch.tensor(rep_w), torch.tensor(th) / torch.tensor(rep_h)])
    src_valid = src_valid * s_scale.view(1, 2, 1, 1).to(flow.device)
    tgt_valid = tgt_valid * t_scale.view(1, 2, 1, 1).to(flow.device)
    src_valid = src_valid + torch.tensor([reg0_src[2], reg0_src[0]]).view(1, 2, 1, 1).to(flow.device)
    tgt_valid = tgt_valid + torch.tensor([reg1_src[2], reg1_src[0]]).view(1, 2, 1, 1).to(flow.device)
    out = torch.zeros((b, 2, src_shape_hw[0], src_shape_hw[1]), dtype=flow.dtype, device=flow.device)
    out[..., reg0_src[0] : reg0_src[1], reg0_src[2] : reg0_src[3]] = tgt_valid - src_valid
    valid = torch.zeros((b, src_shape_hw[0], src_shape_hw[1]), dtype=torch.bool, device=flow.device)
    valid[..., reg0_src[0] : reg0_src[1], reg0_src[2] : reg0_src[3]] = True
    return out, valid


def unmap_channels(
    chan: torch.Tensor, reg0_rep: Sequence[int], reg0_src: Sequence[int], src_shape_hw: Tuple[int, int]
) -> Tuple[torch.Tensor, torch.Tensor]:
    """flow_resizing.py:955-1010: ROI crop -> legacy-nearest resize -> embed in zeros."""
    b, c = chan.shape[:2]
    roi = chan[..., reg0_rep[0] : reg0_rep[1], reg0_rep[2] : reg0_rep[3]]
    sh, sw = reg0_src[1] - reg0_src[0], reg0_src[3] - reg0_src[2]
    res = F.interpolate(roi, size=[sh, sw], mode="nearest")
    out = torch.zeros((b, c, src_shape_hw[0], src_shape_hw[1]), dtype=chan.dtype, device=chan.device)
    out[..., reg0_src[0] : reg0_src[1], reg0_src[2] : reg0_src[3]] = res
    valid = torch.zeros((b, src_shape_hw[0], src_shape_hw[1]), dtype=torch.bool, device=chan.device)
    valid[..., reg0_src[0] : reg0_src[1], reg0_src[2] : reg0_src[3]] = True
    return out, valid


# --------------------------------------------------------------------------- #
# classification refinement: models/ufm.py:1012-1178
# --------------------------------------------------------------------------- #


def neighborhood_features(flow: torch.Tensor, other: torch.Tensor, patch: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """ufm.py:1112-1178.  Returns (B,H,W,P,P,C) samples and (1,1,1,P,P,2) xy offsets."""
    assert patch % 2 == 1
    r = (patch - 1) // 2
    b, c, h, w = other.shape
    di, dj = torch.meshgrid(torch.arange(-r, r + 1), torch.arange(-r, r + 1), indexing="ij")  # di: rows (y), dj: cols (x)
    u, v = torch.meshgrid(torch.arange(w).float(), torch.arange(h).float(), indexing="xy")  # models/utils.py:10-16
    tx = flow[:, 0] + u  # B,H,W
    ty = flow[:, 1] + v
    sx = tx.view(b, h, w, 1, 1) + dj.view(1, 1, 1, patch, patch)
    sy = ty.view(b, h, w, 1, 1) + di.view(1, 1, 1, patch, patch)
    grid = torch.stack((sx, sy), dim=-1).reshape(b, h, w * patch * patch, 2)
    grid = (grid + 0.5) / torch.tensor([w, h]).view(1, 1, 1, 2)
    grid = grid * 2 - 1
    samp = F.grid_sample(other, grid=grid, mode="bicubic", padding_mode="zeros", align_corners=False)
    samp = samp.view(b, c, h, w, patch, patch).permute(0, 2, 3, 4, 5, 1)
    offs = torch.stack((dj, di), dim=-1).view(1, 1, 1, patch, patch, 2).float()
    return samp, offs


def refinement_attention(
    feat1: torch.Tensor, neigh: torch.Tensor, offs: torch.Tensor, temperature: float, bias: torch.Tensor, patch: int
) -> Tuple[torch.Tensor, torch.Tensor]:
    """ufm.py:1041-1095."""
    b, c, h, w = feat1.shape
    q = feat1.permute(0, 2, 3, 1).reshape(b * h * w, 1, c)
    k = neigh.reshape(b * h * w, patch * patch, c)
    vals = offs.reshape(-1, patch * patch, 2)
    score = torch.matmul(q, k.permute(0, 2, 1)) / temperature + bias
    attn = F.softmax(score, dim=-1)
    logp = F.log_softmax(score, dim=-1)
    res = torch.matmul(attn, vals).reshape(b, h, w, 2).permute(0, 3, 1, 2)
    return res, logp.reshape(b, h, w, patch, patch)


def classification_refinement(
    flow: torch.Tensor, feats_2b: torch.Tensor, patch: int, temperature: float, bias: torch.Tensor
) -> Tuple[torch.Tensor, torch.Tensor]:
    """ufm.py:1012-1039."""
    f1, f2 = feats_2b.chunk(2, dim=0)
    neigh, offs = neighborhood_features(flow, f2, patch)
    return refinement_attention(f1, neigh, offs, temperature, bias, patch)


# --------------------------------------------------------------------------- #
# model wiring: models/ufm.py:120-707 (+ :710-1009 for the refinement class)
# --------------------------------------------------------------------------- #

_ADAPTORS = {
    "FlowWithConfidenceAdaptor": U.FlowWithConfidenceAdaptor,
    "FlowAdaptor": U.FlowAdaptor,
    "MaskAdaptor": U.MaskAdaptor,
    "Covariance2DAdaptor": U.Covariance2DAdaptor,
    "ConfidenceAdaptor": U.ConfidenceAdaptor,
}


# --------------------------------------------------------------------------- #
# UNet fine-feature encoder: models/unet_encoder.py:10-71 (PINNED: tests/golden/unet_*.npz are outputs of the
# reference's own class, which loads standalone)
# --------------------------------------------------------------------------- #


class DoubleConvRef(nn.Module):
    """unet_encoder.py:10-23: (Conv3x3 pad 1 -> ReLU) x 2."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(cin, cout, kernel_size=3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(cout, cout, kernel_size=3, padding=1), nn.ReLU(inplace=True)
        )

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.conv(x)


class UNetRef(nn.Module):
    """unet_encoder.py:26-71.  Same attribute names, hence the same state-dict keys (downs.N.conv.{0,2}, ups.{2k} =
    ConvTranspose2d(k=s=2), ups.{2k+1} = DoubleConv, bottleneck, final_conv)."""

    def __init__(self, in_channels: int, out_channels: int, features: Sequence[int] = (64, 128, 256, 512)):
        super().__init__()
        self.downs, self.ups = nn.ModuleList(), nn.ModuleList()
        c = in_channels
        for f in features:
            self.downs.append(DoubleConvRef(c, f))
            c = f
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2)
        self.bottleneck = DoubleConvRef(features[-1], features[-1] * 2)
        for f in reversed(features):
            self.ups.append(nn.ConvTranspose2d(f * 2, f, kernel_size=2, stride=2))
            self.ups.append(DoubleConvRef(f * 2, f))
        self.final_conv = nn.Conv2d(features[0], out_channels, kernel_size=1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        skips = []
        for down in self.downs:  # :53-57
            x = down(x)
            skips.append(x)
            x = self.pool(x)
        x = self.bottleneck(x)
        skips = skips[::-1]
        for i in range(0, len(self.ups), 2):  # :62-69
            x = self.ups[i](x)
            skip = skips[i // 2]
            if x.shape != skip.shape:  # odd sizes: legacy "nearest" resize to the skip's size (:66-67)
                x = F.interpolate(x, size=skip.shape[2:])
            x = torch.cat((skip, x), dim=1)
            x = self.ups[i + 1](x)
        return self.final_conv(x)


def _make_head(head_type: str, feature_head_kwargs: Dict[str, Any], adaptors_kwargs: Dict[str, Any]) -> nn.Module:
    """ufm.py:243-289."""
    if head_type == "moge_conv":  # ufm.py:266-267
        feat = U.MoGeConvFeature(**feature_head_kwargs)
    else:
        assert head_type == "dpt", f"head_type {head_type!r} not supported"
        feat = nn.Sequential(
            U.DPTFeature(**feature_head_kwargs["dpt_feature"]), U.DPTRegressionProcessor(**feature_head_kwargs["dpt_processor"])
        )
    adaptors = [_ADAPTORS[cfg["class"]](**cfg["kwargs"]) for cfg in adaptors_kwargs.values()]
    return nn.Sequential(feat, U.AdaptorMap(*adaptors))


class UFMRef(nn.Module):
    """``UniFlowMatchConfidence`` (ufm.py:474-707); with ``classification_head_kwargs``
    also ``UniFlowMatchClassificationRefinement`` without the UNet option (ufm.py:710-1009)."""

    def __init__(
        self,
        encoder_str: str,
        encoder_kwargs: Dict[str, Any],
        info_sharing_str: str = "global_attention",
        info_sharing_kwargs: Dict[str, Any] = {},
        head_type: str = "dpt",
        feature_head_kwargs: Dict[str, Any] = {},
        adaptors_kwargs: Dict[str, Any] = {},
        uncertainty_head_type: str = "dpt",
        uncertainty_head_kwargs: Dict[str, Any] = {},
        uncertainty_adaptors_kwargs: Dict[str, Any] = {},
        classification_head_kwargs: Optional[Dict[str, Any]] = None,
        temperature: float = 4.0,
        refinement_range: int = 5,
        inference_resolution: Any = (560, 420),
        use_unet_feature: bool = False,
        feature_combine_method: str = "conv",
        **_: Any,
    ):
        super().__init__()
        res = inference_resolution if inference_resolution is not None else [(560, 420)]
        if isinstance(res[0], int):  # base.py:92-93
            res = [res]
        self.inference_resolution = [tuple(r) for r in res]
        self.encoder = U.feature_returner_encoder_factory(encoder_str, **encoder_kwargs)
        self.info_sharing = U.INFO_SHARING_CLASSES[info_sharing_str][1](**info_sharing_kwargs)
        self.head1 = _make_head(head_type, feature_head_kwargs, adaptors_kwargs)
        if len(uncertainty_head_kwargs) > 0:
            self.uncertainty_head = _make_head(uncertainty_head_type, uncertainty_head_kwargs, uncertainty_adaptors_kwargs)
        self.refine = classification_head_kwargs is not None
        if self.refine:
            self.classification_head = U.MLPFeature(**classification_head_kwargs)
            self.refinement_range = refinement_range
            self.temperature = temperature
            self.use_unet_feature = use_unet_feature
            self.feature_combine_method = feature_combine_method
            if use_unet_feature:  # ufm.py:816-825
                self.unet_feature = UNetRef(in_channels=3, out_channels=16, features=[64, 128, 256, 512])
                self.conv1 = nn.Conv2d(32, 32, kernel_size=1)
                if feature_combine_method == "conv":
                    self.conv2 = nn.Conv2d(32, 16, kernel_size=1)
                elif feature_combine_method == "modulate":
                    self.conv2 = nn.Conv2d(16, 16, kernel_size=1)
            self.classification_bias = nn.Parameter(torch.zeros(refinement_range * refinement_range))

    @torch.no_grad()
    def forward(self, img1: torch.Tensor, img2: torch.Tensor, symmetrized: bool = False) -> Out:
        """ufm.py:562-662 / :843-1009 with symmetrized=False; pure fp32 (the reference's CPU
        path is fp32: its autocast targets "cuda" only, base.py:273)."""
        if img1.shape[-2:] != img2.shape[-2:]:
            raise NotImplementedError("Unequal Image sizes are not supported now")  # ufm.py:316-317
        shape1 = (int(img1.shape[2]), int(img1.shape[3]))
        # ``autocast_bf16``: emulate the reference's GPU precision policy on the CPU -- trunk under bf16 autocast
        # (base.py:273), heads and refinement in the fp32 island (ufm.py:635).  Used only to measure how far the
        # reference's OWN bf16 policy moves the outputs (tests/test_model_gpu.py); the default is the fp32 CPU path.
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=bool(getattr(self, "autocast_bf16", False))):
            if symmetrized:  # ufm.py:336-352: pairs (a,b),(b,a): encode a's and b's once, interleave (ufm.py:69-82)
                enc = self.encoder(U.ViTEncoderInput(image=torch.cat((img1[::2], img2[::2]), dim=0), data_norm_type=self.encoder.data_norm_type))
                fa = [e.features.chunk(2, dim=0)[0] for e in enc]
                fb = [e.features.chunk(2, dim=0)[1] for e in enc]
                f1 = [torch.stack((a, b), dim=1).flatten(0, 1) for a, b in zip(fa, fb)]
                f2 = [torch.stack((b, a), dim=1).flatten(0, 1) for a, b in zip(fa, fb)]
            else:
                enc = self.encoder(U.ViTEncoderInput(image=torch.cat((img1, img2), dim=0), data_norm_type=self.encoder.data_norm_type))
                f1 = [e.features.chunk(2, dim=0)[0] for e in enc]
                f2 = [e.features.chunk(2, dim=0)[1] for e in enc]
            final, inter = self.info_sharing(U.MultiViewTransformerInput(features=[f1[-1], f2[-1]]))
        # ufm.py:602-608: only the view-1 pyramid is ever decoded (:637-641, :698-700)
        pyr1 = [f1[-1].float(), inter[0].features[0].float(), inter[1].features[0].float(), final.features[0].float()]
        out = Out()
        head_in = U.PredictionHeadLayeredInput(list_features=pyr1, target_output_shape=shape1)
        h1 = self.head1(head_in)
        flow = h1["flow"].value
        out.flow = FlowOut(flow_output=flow)
        if hasattr(self, "uncertainty_head"):
            hu = self.uncertainty_head(head_in)
            if "flow_cov" in hu:
                out.flow.flow_covariance = hu["flow_cov"].covariance
                out.flow.flow_covariance_inv = hu["flow_cov"].inv_covariance
                out.flow.flow_covariance_log_det = hu["flow_cov"].log_det
            if "keypoint_confidence" in hu:  # ufm.py:653-654 (attached dynamically)
                out.keypoint_confidence = hu["keypoint_confidence"].value.squeeze(1)
            if "non_occluded_mask" in hu:
                out.covisibility = MaskOut(mask=hu["non_occluded_mask"].mask, logits=hu["non_occluded_mask"].logits)
        if self.refine:  # ufm.py:949-1007
            c1 = torch.cat([f1[0].float(), pyr1[-1]], dim=1)
            c2 = torch.cat([f2[0].float(), final.features[1].float()], dim=1)
            cf = self.classification_head(U.PredictionHeadInput(torch.cat([c1, c2], dim=0))).decoded_channels
            if getattr(self, "use_unet_feature", False):  # ufm.py:915-917, :967-983
                un = torch.cat([self.unet_feature(img1), self.unet_feature(img2)], dim=0)
                if self.feature_combine_method == "conv":
                    cf = self.conv2(F.relu(self.conv1(torch.cat([cf, un], dim=1))))
                elif self.feature_combine_method == "modulate":
                    cf = self.conv2(cf * torch.tanh(un))
            residual, logp = classification_refinement(flow, cf, self.refinement_range, self.temperature, self.classification_bias)
            flow = flow + residual
            out.flow.flow_output = flow
            cf0, cf1 = cf.chunk(2, dim=0)
            out.classification_refinement = RefineOut(flow, residual, logp, cf0, cf1)
        return out

    @torch.no_grad()
    def predict_correspondences_batched(
        self, source_image: torch.Tensor, target_image: torch.Tensor, data_norm_type: Optional[str] = None
    ) -> Out:
        """base.py:137-334."""
        src, tgt = to_bchw_normalised(source_image, target_image, self.encoder.data_norm_type, data_norm_type)
        src_hw, tgt_hw = tuple(src.shape[2:]), tuple(tgt.shape[2:])
        target_hw = select_resolution(self.inference_resolution, src_hw[0], src_hw[1], tgt_hw[0], tgt_hw[1])
        s0, s1, reg0_src, reg1_src, reg0_rep, reg1_rep = resize_pair(src, tgt, target_hw)
        res = self.forward(s0, s1)
        out = Out()
        flow, _ = unmap_flow(res.flow.flow_output, reg0_rep, reg0_src, reg1_src, src_hw)
        out.flow = FlowOut(flow_output=flow)
        if res.flow.flow_covariance is not None:  # base.py:295-319
            cov, _ = unmap_channels(res.flow.flow_covariance, reg0_rep, reg0_src, src_hw)
            wr, hr = src_hw[1] / s0.shape[3], src_hw[0] / s0.shape[2]
            out.flow.flow_covariance = cov * torch.tensor([wr**2, hr**2, wr * hr]).view(1, 3, 1, 1)
        if res.covisibility is not None:  # base.py:322-332
            m, _ = unmap_channels(res.covisibility.mask, reg0_rep, reg0_src, src_hw)
            out.covisibility = MaskOut(mask=m.squeeze(1), logits=None)
        return out


# --------------------------------------------------------------------------- #
# configs (every dimension is a constructor kwarg; UFM-Base values are the
# "assumed UFM-Base" of SURVEY 8(d) -- not pinned by any file in the reference)
# --------------------------------------------------------------------------- #


def make_config(
    *,
    enc_dim: int = 1024,
    enc_depth: int = 24,
    enc_heads: int = 16,
    info_dim: int = 768,
    info_depth: int = 12,
    info_heads: int = 12,
    info_indices: Optional[List[int]] = None,
    enc_indices: Optional[List[int]] = None,
    layer_dims: Sequence[int] = (96, 192, 384, 768),
    feature_dim: int = 256,
    resolution_wh: Tuple[int, int] = (518, 518),
    native_img_size: int = 518,
    refine: bool = False,
    refine_dim: int = 16,
    enc_init_values: Optional[float] = 1.0,
    use_unet_feature: bool = False,
    feature_combine_method: str = "conv",
) -> Dict[str, Any]:
    dpt = dict(
        dpt_feature=dict(
            patch_size=14,
            hooks=[0, 1, 2, 3],
            input_feature_dims=[enc_dim, info_dim, info_dim, info_dim],
            layer_dims=list(layer_dims),
            feature_dim=feature_dim,
        ),
        dpt_processor=dict(input_feature_dim=feature_dim, output_dim=2),
    )
    unc = dict(dpt_feature=dict(dpt["dpt_feature"]), dpt_processor=dict(input_feature_dim=feature_dim, output_dim=1))
    cfg: Dict[str, Any] = dict(
        encoder_str="dinov2",
        encoder_kwargs=dict(
            name="dinov2",
            data_norm_type="dinov2",
            patch_size=14,
            size="large",
            embed_dim=enc_dim,
            depth=enc_depth,
            num_heads=enc_heads,
            img_size=native_img_size,
            indices=enc_indices,
            init_values=enc_init_values,
        ),
        info_sharing_str="global_attention",
        info_sharing_kwargs=dict(
            name="info_sharing",
            input_embed_dim=enc_dim,
            max_num_views=2,
            depth=info_depth,
            dim=info_dim,
            num_heads=info_heads,
            indices=info_indices,
        ),
        head_type="dpt",
        feature_head_kwargs=dpt,
        adaptors_kwargs=dict(flow=dict(**{"class": "FlowAdaptor"}, kwargs=dict(name="flow"))),
        uncertainty_head_type="dpt",
        uncertainty_head_kwargs=unc,
        uncertainty_adaptors_kwargs=dict(
            non_occluded_mask=dict(**{"class": "MaskAdaptor"}, kwargs=dict(name="non_occluded_mask"))
        ),
        inference_resolution=tuple(resolution_wh),
    )
    if refine:
        n_first = enc_dim
        cfg["classification_head_kwargs"] = dict(
            input_feature_dim=n_first + info_dim, patch_size=14, output_dim=refine_dim, mlp_ratio=1.0
        )
        cfg["temperature"] = 4.0
        cfg["refinement_range"] = 5
        if use_unet_feature:
            cfg["use_unet_feature"] = True
            cfg["feature_combine_method"] = feature_combine_method
    return cfg


def ufm_base_config(resolution_wh: Tuple[int, int] = (518, 518)) -> Dict[str, Any]:
    return make_config(resolution_wh=resolution_wh)


def ufm_tiny_config(resolution_wh: Tuple[int, int] = (56, 56), refine: bool = False, **kw: Any) -> Dict[str, Any]:
    """Small enough for second-scale CPU tests; same topology (all dims multiples of 64 for the kernels)."""
    return make_config(
        **kw,
        enc_dim=128,
        enc_depth=3,
        enc_heads=2,
        info_dim=128,
        info_depth=4,
        info_heads=2,
        layer_dims=(32, 32, 64, 64),
        feature_dim=64,
        resolution_wh=resolution_wh,
        native_img_size=56,
        refine=refine,
        enc_indices=[0, 2] if refine else None,
    )


def init_weights_(model: nn.Module, seed: int = 0) -> nn.Module:
    """Deterministic CPU random init shared by oracle and product (SURVEY 8(d) "Weights").

    Values are O(1)-scaled so that every code path (pos-embed, cls, LayerScale, biases)
    contributes visibly to the output; generated in state-dict key order from one generator.
    """
    import re

    g = torch.Generator().manual_seed(seed)
    convt = re.compile(r"act_(1|2)_postprocess\.1\.weight$")
    group_norm_weights = {f"{mn}.weight" for mn, m in model.named_modules() if isinstance(m, nn.GroupNorm)}  # (moge_conv head)
    with torch.no_grad():
        # named_parameters() de-duplicates the DPT act_N_postprocess / act_postprocess[N] aliases
        for name, p in sorted(model.named_parameters(), key=lambda kv: kv[0]):
            if p.dim() >= 2 and "pos_embed" not in name and "cls_token" not in name:
                fan_in = p.shape[0] if convt.search(name) else p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / fan_in**0.5))
            elif name.endswith("gamma"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif ("norm" in name and name.endswith("weight")) or name in group_norm_weights:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith("classification_bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            elif "pos_embed" in name or "cls_token" in name:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
    return model
