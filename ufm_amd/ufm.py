"""UFM model classes on HIP kernels: drop-in for ``uniflowmatch/models/ufm.py``.

  UniFlowMatch                          <- ufm.py:120-471
  UniFlowMatchConfidence  (UFM-Base)    <- ufm.py:474-707
  UniFlowMatchClassificationRefinement  (UFM-Refine) <- ufm.py:710-1238

Same constructor kwargs (they are what ``PyTorchModelHubMixin`` serialises into ``config.json``),
same ``from_pretrained`` / ``from_pretrained_ckpt`` / ``forward(view1, view2)`` /
``predict_correspondences_batched`` surface, same state-dict namespace.  The forward pass is
``ufm_amd.engine.Engine`` (hand-written gfx950 kernels); nothing here computes with torch ops.
"""

from __future__ import annotations

import os
from typing import Any, Dict, List, Optional, Tuple

import torch
from huggingface_hub import PyTorchModelHubMixin
from torch import nn

from . import modules as M
from .base import (
    UFMClassificationRefinementOutput,
    UFMFlowFieldOutput,
    UFMMaskFieldOutput,
    UFMOutputInterface,
    UniFlowMatchModelsBase,
)


from .keymap import load_checked, modify_state_dict  # noqa: E402,F401  (modify_state_dict: reference name, ufm.py:85)


def _package_uncertainty(result: UFMOutputInterface, hu: Dict[str, Any]) -> None:
    """The uncertainty head's named outputs (ufm.py:644-660): covariance triple, keypoint confidence, covisibility."""
    if "flow_cov" in hu:
        result.flow.flow_covariance = hu["flow_cov"]["covariance"]
        result.flow.flow_covariance_inv = hu["flow_cov"]["inv_covariance"]
        result.flow.flow_covariance_log_det = hu["flow_cov"]["log_det"]
    if "keypoint_confidence" in hu:
        result.keypoint_confidence = hu["keypoint_confidence"]["value"].squeeze(1)
    if "non_occluded_mask" in hu:
        result.covisibility = UFMMaskFieldOutput(mask=hu["non_occluded_mask"]["value"], logits=hu["non_occluded_mask"]["logits"])


class UniFlowMatch(UniFlowMatchModelsBase, PyTorchModelHubMixin):
    def __init__(
        self,
        encoder_str: str,
        encoder_kwargs: Dict[str, Any],
        info_sharing_and_head_structure: str = "dual+single",
        info_sharing_str: str = "global_attention",
        info_sharing_kwargs: Dict[str, Any] = {},
        encoder_skip_connection: Optional[List[int]] = None,
        info_sharing_skip_connection: Optional[List[int]] = None,
        head_type: str = "dpt",
        feature_head_kwargs: Dict[str, Any] = {},
        adaptors_kwargs: Dict[str, Any] = {},
        pretrained_checkpoint_path: Optional[str] = None,
        inference_resolution: Optional[Tuple[int, int]] = (560, 420),  # (W, H)
        *args,
        **kwargs,
    ):
        UniFlowMatchModelsBase.__init__(self, inference_resolution=inference_resolution)
        assert info_sharing_and_head_structure == "dual+single", "Only dual+single is supported now"
        assert head_type != "linear", "Linear head is not supported, because it have major disadvantage to DPTs"
        self.encoder_skip_connection = encoder_skip_connection
        self.info_sharing_skip_connection = info_sharing_skip_connection
        if encoder_str != "dinov2":
            raise NotImplementedError(f"encoder {encoder_str!r}: only the DINOv2 ViT encoder is built")
        if info_sharing_str not in M.INFO_SHARING_CLASSES:
            raise NotImplementedError(f"info_sharing_str {info_sharing_str!r}: built are {sorted(M.INFO_SHARING_CLASSES)}")
        self.encoder: nn.Module = M.DINOv2Encoder(**encoder_kwargs)
        self.head_type = head_type
        self.info_sharing: nn.Module = M.INFO_SHARING_CLASSES[info_sharing_str][1](**info_sharing_kwargs)
        self.head1: nn.Module = M.make_head(head_type, feature_head_kwargs, adaptors_kwargs)
        if pretrained_checkpoint_path is not None:  # ufm.py:198-217
            ckpt = torch.load(pretrained_checkpoint_path, map_location="cpu", weights_only=True)
            if "state_dict" in ckpt:  # Lightning training checkpoint: "model." prefix, dropped keys (keymap.py)
                load_checked(self, ckpt["state_dict"], lightning=True, what=pretrained_checkpoint_path)
            else:
                load_checked(self, ckpt["model"], what=pretrained_checkpoint_path)

    @classmethod
    def from_pretrained_ckpt(cls, pretrained_model_name_or_path, strict=True, **kw):
        """ufm.py:219-241 (``weights_only=True``: nothing from the file is executed)."""
        if os.path.isfile(pretrained_model_name_or_path):
            ckpt = torch.load(pretrained_model_name_or_path, map_location="cpu", weights_only=True)
            model = cls(**ckpt["model_args"])
            if strict:
                model.load_state_dict(ckpt["model"], strict=True)
            else:  # never silently: missing parameters raise, extra keys must be on keymap.ALLOWED_UNEXPECTED
                load_checked(model, ckpt["model"], what=pretrained_model_name_or_path)
            return model
        raise ValueError(f"Pretrained model {pretrained_model_name_or_path} not found.")

    # PyTorchModelHubMixin.from_pretrained ends here.  The mixin's default is load(strict=False) with no check at all:
    # a parameter whose name differs would silently keep its random init.  Both loaders go through
    # keymap.load_checked instead -- missing parameters raise, as the reference asserts at ufm.py:216-217.
    @classmethod
    def _load_as_safetensor(cls, model, model_file: str, map_location: str, strict: bool):
        from safetensors.torch import load_file

        load_checked(model, load_file(model_file, device="cpu"), what=model_file)
        if map_location != "cpu":
            model.to(map_location)
        model.eval()
        return model

    @classmethod
    def _load_as_pickle(cls, model, model_file: str, map_location: str, strict: bool):
        sd = torch.load(model_file, map_location="cpu", weights_only=True)  # nothing from the file is executed
        load_checked(model, sd, what=model_file)
        if map_location != "cpu":
            model.to(map_location)
        model.eval()
        return model

    # ------------------------------------------------------------------ forward
    def _check_views(self, view1, view2):
        img1, img2 = view1["img"], view2["img"]
        if img1.shape[-2:] != img2.shape[-2:]:
            raise NotImplementedError("Unequal Image sizes are not supported now")  # ufm.py:316-317
        want = self.encoder.data_norm_type
        if view1.get("data_norm_type", want) != want:
            raise AssertionError(f"encoder expects data_norm_type {want!r}")
        assert img1.dtype == torch.float32 and img1.dim() == 4 and img1.shape[1] == 3
        return img1, img2

    def forward(self, view1, view2) -> UFMOutputInterface:
        """Lower-level call of the reference (ufm.py:356-433): views carry normalised float32 BCHW
        images at network resolution; outputs are at that resolution."""
        img1, img2 = self._check_views(view1, view2)
        if not img1.is_cuda:
            raise RuntimeError("ufm_amd runs on an AMD GPU only (no CPU fallback)")
        H, W = int(img1.shape[2]), int(img1.shape[3])
        # symmetrized (ufm.py:336-352): pairs come as (a,b),(b,a): a's and b's are encoded once and the features interleaved
        sym = bool(view1.get("symmetrized", False))
        with torch.cuda.device(img1.device):
            return self._forward_device(img1.contiguous(), img2.contiguous(), 1, [1.0] * 3, [0.0] * 3, H, W, H, W, H, W, symmetrized=sym)

    def _forward_device(self, src, tgt, layout, scale3, shift3, H, W, hs, ws, ht, wt, symmetrized: bool = False) -> UFMOutputInterface:
        raw = self.engine().forward(src, tgt, layout=layout, scale3=scale3, shift3=shift3, H=H, W=W, Hs=hs, Ws=ws, Ht=ht, Wt=wt, symmetrized=symmetrized)
        return self._package(raw)

    def _package(self, raw: Dict[str, Any]) -> UFMOutputInterface:
        result = UFMOutputInterface()
        h1 = raw["head1"]
        if "flow" in h1:
            result.flow = UFMFlowFieldOutput(flow_output=h1["flow"]["value"])
        if "non_occluded_mask" in h1:
            result.covisibility = UFMMaskFieldOutput(mask=h1["non_occluded_mask"]["value"], logits=h1["non_occluded_mask"]["logits"])
        return result

    def get_parameter_groups(self) -> Dict[str, torch.nn.ParameterList]:
        return {
            "encoder": torch.nn.ParameterList(self.encoder.parameters()),
            "info_sharing": torch.nn.ParameterList(self.info_sharing.parameters()),
            "output_head": torch.nn.ParameterList(self.head1.parameters()),
        }


class UniFlowMatchConfidence(UniFlowMatch, PyTorchModelHubMixin):
    """UFM-Base: flow head + covisibility ("uncertainty") head."""

    def __init__(
        self,
        encoder_str: str,
        encoder_kwargs: Dict[str, Any],
        info_sharing_and_head_structure: str = "dual+single",
        info_sharing_str: str = "global_attention",
        info_sharing_kwargs: Dict[str, Any] = {},
        head_type: str = "dpt",
        feature_head_kwargs: Dict[str, Any] = {},
        adaptors_kwargs: Dict[str, Any] = {},
        detach_uncertainty_head: bool = True,
        uncertainty_head_type: str = "dpt",
        uncertainty_head_kwargs: Dict[str, Any] = {},
        uncertainty_adaptors_kwargs: Dict[str, Any] = {},
        pretrained_backbone_checkpoint_path: Optional[str] = None,
        pretrained_checkpoint_path: Optional[str] = None,
        inference_resolution: Optional[Tuple[int, int]] = (560, 420),  # WH
        *args,
        **kwargs,
    ):
        UniFlowMatch.__init__(
            self,
            encoder_str=encoder_str,
            encoder_kwargs=encoder_kwargs,
            info_sharing_and_head_structure=info_sharing_and_head_structure,
            info_sharing_str=info_sharing_str,
            info_sharing_kwargs=info_sharing_kwargs,
            head_type=head_type,
            feature_head_kwargs=feature_head_kwargs,
            adaptors_kwargs=adaptors_kwargs,
            pretrained_checkpoint_path=pretrained_backbone_checkpoint_path,
            inference_resolution=inference_resolution,
        )
        assert uncertainty_head_type == "dpt", "Only DPT is supported for uncertainty head now"
        self.uncertainty_head = M.make_head(uncertainty_head_type, uncertainty_head_kwargs, uncertainty_adaptors_kwargs)
        assert pretrained_checkpoint_path is None, "Pretrained weights are not supported for now"
        self.detach_uncertainty_head = detach_uncertainty_head

    def _package(self, raw: Dict[str, Any]) -> UFMOutputInterface:
        result = UFMOutputInterface()
        result.flow = UFMFlowFieldOutput(flow_output=raw["head1"]["flow"]["value"])
        _package_uncertainty(result, raw["uncertainty_head"])
        return result

    def get_parameter_groups(self) -> Dict[str, torch.nn.ParameterList]:
        groups = UniFlowMatch.get_parameter_groups(self)
        groups["uncertainty_head"] = torch.nn.ParameterList(self.uncertainty_head.parameters())
        return groups


class UniFlowMatchClassificationRefinement(UniFlowMatch, PyTorchModelHubMixin):
    """UFM-Refine: regression flow + local classification refinement (ufm.py:1012-1178)."""

    def __init__(
        self,
        encoder_str: str,
        encoder_kwargs: Dict[str, Any],
        info_sharing_and_head_structure: str = "dual+single",
        info_sharing_str: str = "global_attention",
        info_sharing_kwargs: Dict[str, Any] = {},
        head_type: str = "dpt",
        feature_head_kwargs: Dict[str, Any] = {},
        adaptors_kwargs: Dict[str, Any] = {},
        detach_uncertainty_head: bool = True,
        uncertainty_head_type: str = "dpt",
        uncertainty_head_kwargs: Dict[str, Any] = {},
        uncertainty_adaptors_kwargs: Dict[str, Any] = {},
        temperature: float = 4,
        use_unet_feature: bool = False,
        classification_head_type: str = "patch_mlp",
        classification_head_kwargs: Dict[str, Any] = {},
        feature_combine_method: str = "conv",
        refinement_range: int = 5,
        pretrained_backbone_checkpoint_path: Optional[str] = None,
        pretrained_checkpoint_path: Optional[str] = None,
        inference_resolution: Optional[Tuple[int, int]] = (560, 420),  # WH
        *args,
        **kwargs,
    ):
        UniFlowMatch.__init__(
            self,
            encoder_str=encoder_str,
            encoder_kwargs=encoder_kwargs,
            info_sharing_and_head_structure=info_sharing_and_head_structure,
            info_sharing_str=info_sharing_str,
            info_sharing_kwargs=info_sharing_kwargs,
            head_type=head_type,
            feature_head_kwargs=feature_head_kwargs,
            adaptors_kwargs=adaptors_kwargs,
            pretrained_checkpoint_path=pretrained_backbone_checkpoint_path,
            inference_resolution=inference_resolution,
        )
        assert classification_head_type == "patch_mlp", "Only DPT is supported for uncertainty head now"
        self.classification_head_type = classification_head_type
        self.classification_head = M.MLPFeatureParams(**classification_head_kwargs)
        self.refinement_range = refinement_range
        self.temperature = temperature
        assert pretrained_checkpoint_path is None, "Pretrained weights are not supported for now"
        self.use_unet_feature = use_unet_feature
        self.feature_combine_method = feature_combine_method
        if use_unet_feature:  # ufm.py:816-825 (parameter containers; Engine._unet / ufm_unet_combine compute)
            if feature_combine_method not in ("conv", "modulate"):
                raise ValueError(f"feature_combine_method {feature_combine_method!r}: the reference defines 'conv' and 'modulate' (ufm.py:821-825)")
            self.unet_feature = M.UNetParams(in_channels=3, out_channels=16, features=[64, 128, 256, 512])
            self.conv1 = nn.Conv2d(32, 32, kernel_size=1, stride=1, padding=0)
            self.conv2 = nn.Conv2d(32 if feature_combine_method == "conv" else 16, 16, kernel_size=1, stride=1, padding=0)
        self.classification_bias = nn.Parameter(torch.zeros(refinement_range * refinement_range))
        if len(uncertainty_head_kwargs) > 0:
            assert uncertainty_head_type == "dpt", "Only DPT is supported for uncertainty head now"
            self.uncertainty_head = M.make_head(uncertainty_head_type, uncertainty_head_kwargs, uncertainty_adaptors_kwargs)
            self.detach_uncertainty_head = detach_uncertainty_head

    def _package(self, raw: Dict[str, Any]) -> UFMOutputInterface:
        result = UFMOutputInterface()
        ref = raw["refine"]
        B = ref["flow"].shape[0]
        result.flow = UFMFlowFieldOutput(flow_output=ref["flow"])
        if "uncertainty_head" in raw:
            _package_uncertainty(result, raw["uncertainty_head"])
        # quirk kept from the reference: regression_flow_output is the REFINED flow (ufm.py:991,1002)
        result.classification_refinement = UFMClassificationRefinementOutput(
            regression_flow_output=ref["flow"],
            residual=ref["residual"],
            log_softmax=ref["log_softmax"],
            feature_map_0=ref["feats"][:B],
            feature_map_1=ref["feats"][B:],
        )
        return result

    def get_parameter_groups(self) -> Dict[str, torch.nn.ParameterList]:
        groups = UniFlowMatch.get_parameter_groups(self)
        groups["classification_head"] = torch.nn.ParameterList(self.classification_head.parameters())
        if hasattr(self, "uncertainty_head"):
            groups["uncertainty_head"] = torch.nn.ParameterList(self.uncertainty_head.parameters())
        return groups
