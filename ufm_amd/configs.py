"""Constructor-kwarg builders.  The real UFM-Base / UFM-Refine hyper-parameters live only in the
Hugging Face Hub ``config.json`` (not in the reference tree); these are the
"assumed UFM-Base" values of SURVEY 8(d), every one of them an ordinary constructor kwarg."""

from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple


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
    refine_mlp_ratio: float = 1.0,
    enc_init_values: Optional[float] = 1.0,
    use_unet_feature: bool = False,
    feature_combine_method: str = "conv",
) -> Dict[str, Any]:
    dpt_feature = dict(
        patch_size=14,
        hooks=[0, 1, 2, 3],
        input_feature_dims=[enc_dim, info_dim, info_dim, info_dim],
        layer_dims=list(layer_dims),
        feature_dim=feature_dim,
    )
    cfg: Dict[str, Any] = dict(
        encoder_str="dinov2",
        encoder_kwargs=dict(
            name="dinov2", data_norm_type="dinov2", patch_size=14, size="large", embed_dim=enc_dim, depth=enc_depth,
            num_heads=enc_heads, img_size=native_img_size, indices=enc_indices, init_values=enc_init_values,
        ),
        info_sharing_str="global_attention",
        info_sharing_kwargs=dict(
            name="info_sharing", input_embed_dim=enc_dim, max_num_views=2, depth=info_depth, dim=info_dim,
            num_heads=info_heads, indices=info_indices,
        ),
        head_type="dpt",
        feature_head_kwargs=dict(dpt_feature=dict(dpt_feature), dpt_processor=dict(input_feature_dim=feature_dim, output_dim=2)),
        adaptors_kwargs=dict(flow={"class": "FlowAdaptor", "kwargs": dict(name="flow")}),
        uncertainty_head_type="dpt",
        uncertainty_head_kwargs=dict(dpt_feature=dict(dpt_feature), dpt_processor=dict(input_feature_dim=feature_dim, output_dim=1)),
        uncertainty_adaptors_kwargs=dict(non_occluded_mask={"class": "MaskAdaptor", "kwargs": dict(name="non_occluded_mask")}),
        inference_resolution=tuple(resolution_wh),
    )
    if refine:
        cfg["classification_head_kwargs"] = dict(
            input_feature_dim=enc_dim + info_dim, patch_size=14, output_dim=refine_dim, mlp_ratio=refine_mlp_ratio
        )
        cfg["temperature"] = 4.0
        cfg["refinement_range"] = 5
        if use_unet_feature:  # ufm.py:816-825: UNet fine features combined with the patch-MLP features
            cfg["use_unet_feature"] = True
            cfg["feature_combine_method"] = feature_combine_method
    return cfg


def ufm_base_config(resolution_wh: Tuple[int, int] = (518, 518)) -> Dict[str, Any]:
    """UniFlowMatchConfidence kwargs: DINOv2 ViT-L/14 + 12x768 joint attention + two DPT heads."""
    return make_config(resolution_wh=resolution_wh)


def ufm_refine_config(resolution_wh: Tuple[int, int] = (518, 518), **kw: Any) -> Dict[str, Any]:
    """UniFlowMatchClassificationRefinement kwargs (first + last encoder features returned)."""
    return make_config(resolution_wh=resolution_wh, refine=True, enc_indices=[5, 23], **kw)


def ufm_tiny_config(resolution_wh: Tuple[int, int] = (56, 56), refine: bool = False, **kw: Any) -> Dict[str, Any]:
    """Same topology at test size; every channel count satisfies the kernels' tile multiples."""
    return make_config(
        **kw,
        enc_dim=128, enc_depth=3, enc_heads=2, info_dim=128, info_depth=4, info_heads=2, layer_dims=(32, 32, 64, 64),
        feature_dim=64, resolution_wh=resolution_wh, native_img_size=56, refine=refine, enc_indices=[0, 2] if refine else None,
    )
