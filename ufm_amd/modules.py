"""Parameter containers for the UFM hot path.

These classes own the weights under the same state-dict names the reference's model tree uses
(``encoder.model.*`` per ``uniflowmatch/models/ufm.py:208-210``, ``info_sharing.*`` per ``:193``,
``head1.0.0.* / head1.0.1.*`` per ``:262-273``, ``uncertainty_head.*`` per ``:553``,
``classification_head.*`` per ``:805``).  The third-party ``uniception`` module attribute names
inside those prefixes are recalled from the upstream project (not present in the reference
mount) and are isolated in this one file.

They are NOT compute modules: ``forward`` raises.  All arithmetic happens in
``ufm_amd.engine`` through the C-ABI HIP kernels; there is no PyTorch fallback path.
"""

from __future__ import annotations

import math
from typing import Any, Dict, List, Optional, Sequence, Union

import torch
from torch import nn

IMAGE_NORMALIZATION: Dict[str, Dict[str, Sequence[float]]] = {
    # name -> mean/std; the table the reference reads at models/base.py:75,183-229
    "dummy": dict(mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0)),
    "identity": dict(mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0)),
    "croco": dict(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)),
    "dinov2": dict(mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)),
    "dust3r": dict(mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)),
    "patch_embedder": dict(mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)),
}


class _Holder(nn.Module):
    """A module that only holds parameters."""

    def forward(self, *a: Any, **k: Any):  # pragma: no cover - guard
        raise RuntimeError(
            f"{type(self).__name__} is a parameter container; compute runs in ufm_amd.engine on HIP kernels "
            "(there is no PyTorch fallback)"
        )


class _Gamma(_Holder):
    def __init__(self, dim: int, init_values: float):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))


class _AttnParams(_Holder):
    def __init__(self, dim: int, num_heads: int, qkv_bias: bool = True):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, 3 * dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=True)


class _MlpParams(_Holder):
    def __init__(self, dim: int, hidden: int, out: Optional[int] = None):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, out or dim)


class BlockParams(_Holder):
    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0, qkv_bias: bool = True, init_values: Optional[float] = None):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _AttnParams(dim, num_heads, qkv_bias)
        self.ls1 = _Gamma(dim, init_values) if init_values is not None else nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _MlpParams(dim, int(dim * mlp_ratio))
        self.ls2 = _Gamma(dim, init_values) if init_values is not None else nn.Identity()


class _PatchEmbedParams(_Holder):
    def __init__(self, patch: int, dim: int):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)


class DinoViTParams(_Holder):
    def __init__(self, img_size: int, patch_size: int, embed_dim: int, depth: int, num_heads: int, mlp_ratio: float, init_values: Optional[float]):
        super().__init__()
        self.patch_size, self.embed_dim, self.num_heads = patch_size, embed_dim, num_heads
        self.interpolate_offset = 0.1
        self.patch_embed = _PatchEmbedParams(patch_size, embed_dim)
        g = img_size // patch_size
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, g * g + 1, embed_dim))
        self.blocks = nn.ModuleList([BlockParams(embed_dim, num_heads, mlp_ratio, True, init_values) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)


_DINOV2_SIZES = {
    "small": dict(embed_dim=384, depth=12, num_heads=6),
    "base": dict(embed_dim=768, depth=12, num_heads=12),
    "large": dict(embed_dim=1024, depth=24, num_heads=16),
}


class DINOv2Encoder(_Holder):
    """What ``feature_returner_encoder_factory("dinov2", **encoder_kwargs)`` builds (ufm.py:187)."""

    def __init__(
        self,
        name: str = "dinov2",
        data_norm_type: str = "dinov2",
        patch_size: int = 14,
        size: str = "large",
        with_registers: bool = False,
        indices: Optional[Union[int, List[int]]] = None,
        norm_intermediate: bool = True,
        keep_first_n_layers: Optional[int] = None,
        **kw: Any,
    ):
        super().__init__()
        if with_registers:
            raise NotImplementedError("DINOv2 register tokens are outside the UFM hot path")
        if not norm_intermediate:
            raise NotImplementedError("norm_intermediate=False is not used by UFM")
        self.name, self.data_norm_type, self.patch_size = name, data_norm_type, patch_size
        dims = dict(_DINOV2_SIZES[size])
        for k in ("embed_dim", "depth", "num_heads"):
            if k in kw:
                dims[k] = kw[k]
        self.model = DinoViTParams(kw.get("img_size", 518), patch_size, mlp_ratio=kw.get("mlp_ratio", 4.0), init_values=kw.get("init_values", 1.0), **dims)
        if keep_first_n_layers is not None:
            self.model.blocks = self.model.blocks[:keep_first_n_layers]
        depth = len(self.model.blocks)
        if indices is None:
            indices = [depth - 1]
        if isinstance(indices, int):
            indices = list(range(depth - indices, depth))
        self.indices = [i % depth for i in indices]
        self.enc_embed_dim = dims["embed_dim"]


def sinusoid_view_table(n_position: int, dim: int, base: float = 10000.0) -> torch.Tensor:
    """1-D sin/cos view-index encoding ([U] info-sharing view positional table)."""
    tab = torch.zeros(n_position, dim, dtype=torch.float64)
    for pos in range(n_position):
        for i in range(dim):
            ang = pos / math.pow(base, 2.0 * (i // 2) / dim)
            tab[pos, i] = math.sin(ang) if i % 2 == 0 else math.cos(ang)
    return tab.float()


class GlobalAttentionInfoSharing(_Holder):
    """``INFO_SHARING_CLASSES["global_attention"][1](**info_sharing_kwargs)`` (ufm.py:193)."""

    def __init__(
        self,
        name: str = "global_attention",
        input_embed_dim: int = 1024,
        max_num_views: int = 2,
        use_rand_idx_pe_for_non_reference_views: bool = False,
        size: Optional[str] = None,
        depth: int = 12,
        dim: int = 768,
        num_heads: int = 12,
        mlp_ratio: float = 4.0,
        qkv_bias: bool = True,
        init_values: Optional[float] = None,
        indices: Optional[List[int]] = None,
        norm_intermediate: bool = True,
        **_: Any,
    ):
        super().__init__()
        if size is not None:
            depth, dim, num_heads = {"base": (12, 768, 12), "large": (24, 1024, 16)}[size]
        if not norm_intermediate:
            raise NotImplementedError("norm_intermediate=False is not used by UFM")
        self.name, self.input_embed_dim, self.max_num_views = name, input_embed_dim, max_num_views
        self.depth, self.dim, self.num_heads = depth, dim, num_heads
        self.proj_embed = nn.Linear(input_embed_dim, dim, bias=True) if input_embed_dim != dim else nn.Identity()
        self.self_attention_blocks = nn.ModuleList([BlockParams(dim, num_heads, mlp_ratio, qkv_bias, init_values) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.register_buffer("view_pos_table", sinusoid_view_table(max_num_views, dim), persistent=False)
        self.indices = list(indices) if indices is not None else [depth // 2 - 1, (3 * depth) // 4 - 1]


class _CrossAttnParams(_Holder):
    def __init__(self, dim: int, num_heads: int, qkv_bias: bool = True):
        super().__init__()
        self.num_heads = num_heads
        self.projq = nn.Linear(dim, dim, bias=qkv_bias)
        self.projk = nn.Linear(dim, dim, bias=qkv_bias)
        self.projv = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class CrossBlockParams(_Holder):
    """[U] CrossAttentionBlock (CroCo decoder block): self-attention, cross-attention to the other view, MLP."""

    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0, qkv_bias: bool = True, init_values: Optional[float] = None, norm_cross_tokens: bool = True):
        super().__init__()
        ls = (lambda: _Gamma(dim, init_values)) if init_values is not None else (lambda: nn.Identity())
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _AttnParams(dim, num_heads, qkv_bias)
        self.ls1 = ls()
        self.norm_y = nn.LayerNorm(dim, eps=1e-6) if norm_cross_tokens else nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.cross_attn = _CrossAttnParams(dim, num_heads, qkv_bias)
        self.ls2 = ls()
        self.norm3 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _MlpParams(dim, int(dim * mlp_ratio))
        self.ls3 = ls()


class CrossAttentionInfoSharing(_Holder):
    """``INFO_SHARING_CLASSES["cross_attention"][1](**info_sharing_kwargs)`` (ufm.py:193): one branch of CroCo-style
    decoder blocks per view, each block attending to its own view and to the other view's tokens of the previous layer;
    ``rope_freq`` turns on RoPE-2D on q / k (upstream: a callable `custom_positional_encoding`).  The third-party class is
    absent from the reference: restated, **parity unpinned** (oracle/uniception_ref.py)."""

    def __init__(
        self,
        name: str = "cross_attention",
        input_embed_dim: int = 1024,
        num_views: int = 2,
        size: Optional[str] = None,
        depth: int = 12,
        dim: int = 768,
        num_heads: int = 12,
        mlp_ratio: float = 4.0,
        qkv_bias: bool = True,
        init_values: Optional[float] = None,
        indices: Optional[List[int]] = None,
        norm_intermediate: bool = True,
        norm_cross_tokens: bool = True,
        rope_freq: Optional[float] = None,
        **_: Any,
    ):
        super().__init__()
        if size is not None:
            depth, dim, num_heads = {"base": (12, 768, 12), "large": (24, 1024, 16)}[size]
        if not norm_intermediate:
            raise NotImplementedError("norm_intermediate=False is not used by UFM")
        if num_views != 2:
            raise NotImplementedError("UFM shares information between exactly two views (ufm.py:390)")
        if not norm_cross_tokens:
            raise NotImplementedError("norm_cross_tokens=False is not built")
        self.name, self.input_embed_dim, self.num_views = name, input_embed_dim, num_views
        self.depth, self.dim, self.num_heads = depth, dim, num_heads
        self.rope_freq = float(rope_freq) if rope_freq else None
        self.proj_embed = nn.Linear(input_embed_dim, dim, bias=True) if input_embed_dim != dim else nn.Identity()
        self.multi_view_branches = nn.ModuleList(
            [nn.ModuleList([CrossBlockParams(dim, num_heads, mlp_ratio, qkv_bias, init_values, norm_cross_tokens) for _ in range(depth)]) for _ in range(num_views)]
        )
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.indices = list(indices) if indices is not None else [depth // 2 - 1, (3 * depth) // 4 - 1]


INFO_SHARING_CLASSES = {"global_attention": (None, GlobalAttentionInfoSharing), "cross_attention": (None, CrossAttentionInfoSharing)}


class _RCUParams(_Holder):
    def __init__(self, f: int):
        super().__init__()
        self.conv1 = nn.Conv2d(f, f, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(f, f, 3, 1, 1, bias=True)


class _FusionParams(_Holder):
    def __init__(self, f: int):
        super().__init__()
        self.out_conv = nn.Conv2d(f, f, 1, 1, 0, bias=True)
        self.resConfUnit1 = _RCUParams(f)
        self.resConfUnit2 = _RCUParams(f)


class DPTFeatureParams(_Holder):
    def __init__(
        self,
        patch_size: int = 14,
        main_tasks: Sequence[str] = ("rgb",),
        hooks: Sequence[int] = (0, 1, 2, 3),
        input_feature_dims: Union[int, Sequence[int]] = 768,
        layer_dims: Sequence[int] = (96, 192, 384, 768),
        feature_dim: int = 256,
        use_bn: bool = False,
        output_width_ratio: float = 1,
        **_: Any,
    ):
        super().__init__()
        if use_bn or output_width_ratio != 1:
            raise NotImplementedError("DPT use_bn / output_width_ratio variants are not used by UFM")
        if isinstance(input_feature_dims, int):
            input_feature_dims = [input_feature_dims] * 4
        self.patch_size, self.hooks = patch_size, list(hooks)
        self.input_feature_dims, self.layer_dims, self.feature_dim = list(input_feature_dims), list(layer_dims), feature_dim
        ld, idim = self.layer_dims, self.input_feature_dims
        self.scratch = _Holder()
        for i in range(4):
            setattr(self.scratch, f"layer{i + 1}_rn", nn.Conv2d(ld[i], feature_dim, 3, 1, 1, bias=False))
            setattr(self.scratch, f"refinenet{i + 1}", _FusionParams(feature_dim))
        self.act_1_postprocess = nn.Sequential(nn.Conv2d(idim[0], ld[0], 1), nn.ConvTranspose2d(ld[0], ld[0], 4, 4, 0, bias=True))
        self.act_2_postprocess = nn.Sequential(nn.Conv2d(idim[1], ld[1], 1), nn.ConvTranspose2d(ld[1], ld[1], 2, 2, 0, bias=True))
        self.act_3_postprocess = nn.Sequential(nn.Conv2d(idim[2], ld[2], 1))
        self.act_4_postprocess = nn.Sequential(nn.Conv2d(idim[3], ld[3], 1), nn.Conv2d(ld[3], ld[3], 3, 2, 1))
        self.act_postprocess = nn.ModuleList([self.act_1_postprocess, self.act_2_postprocess, self.act_3_postprocess, self.act_4_postprocess])


class DPTProcessorParams(_Holder):
    def __init__(self, input_feature_dim: int = 256, output_dim: int = 2, hidden_dims: Optional[Sequence[int]] = None, **_: Any):
        super().__init__()
        if hidden_dims is None:
            hidden_dims = [input_feature_dim // 2, 32]
        self.output_dim = output_dim
        self.conv1 = nn.Conv2d(input_feature_dim, hidden_dims[0], 3, 1, 1)
        self.conv2 = nn.Sequential(nn.Conv2d(hidden_dims[0], hidden_dims[1], 3, 1, 1), nn.ReLU(True), nn.Conv2d(hidden_dims[1], output_dim, 1, 1, 0))


class MLPFeatureParams(_Holder):
    def __init__(self, input_feature_dim: int, patch_size: int, output_dim: int, mlp_ratio: float = 4.0, **_: Any):
        super().__init__()
        self.patch_size, self.output_dim = patch_size, output_dim
        self.mlp = _MlpParams(input_feature_dim, int(mlp_ratio * input_feature_dim), output_dim * patch_size * patch_size)


class _ResConvBlockParams(_Holder):
    """[U] MoGe ResidualConvBlock: GroupNorm -> ReLU -> conv3x3 (replicate) -> GroupNorm -> ReLU -> conv3x3 (replicate) + skip.
    torch modules hold the parameters under the upstream names (layers.0/2/3/5, skip_connection)."""

    def __init__(self, cin: int, cout: int, hidden: int, norm: str = "group_norm"):
        super().__init__()
        groups = (lambda ch: max(ch // 32, 1)) if norm == "group_norm" else (lambda ch: 1)
        self.layers = nn.Sequential(
            nn.GroupNorm(groups(cin), cin), nn.ReLU(), nn.Conv2d(cin, hidden, 3, padding=1, padding_mode="replicate"),
            nn.GroupNorm(groups(hidden), hidden), nn.ReLU(), nn.Conv2d(hidden, cout, 3, padding=1, padding_mode="replicate"),
        )
        self.skip_connection = nn.Conv2d(cin, cout, 1) if cin != cout else nn.Identity()


class MoGeConvParams(_Holder):
    """``head_type="moge_conv"`` (ufm.py:266-267): parameters of [U] MoGeConvFeature under the upstream attribute names
    (projects.N, upsample_blocks.K.0.{0,1}, upsample_blocks.K.{1..}.layers.*, output_block.J.*).  The forward pass is
    Engine._head_moge.  **Parity unpinned** (the class is absent from the reference; oracle/uniception_ref.py restates it)."""

    def __init__(
        self,
        input_feature_dims: Union[int, Sequence[int]] = 768,
        dim_out: Union[int, Sequence[int]] = 2,
        num_features: int = 4,
        dim_proj: int = 512,
        dim_upsample: Sequence[int] = (256, 128, 128),
        dim_times_res_block_hidden: int = 1,
        num_res_blocks: int = 1,
        res_block_norm: str = "group_norm",
        last_res_blocks: int = 0,
        last_conv_channels: int = 32,
        last_conv_size: int = 1,
        patch_size: int = 14,
        **_: Any,
    ):
        super().__init__()
        if isinstance(input_feature_dims, int):
            input_feature_dims = [input_feature_dims] * num_features
        if isinstance(dim_out, int):
            dim_out = [dim_out]
        if len(dim_out) != 1 or last_res_blocks != 0 or last_conv_size != 1:
            raise NotImplementedError("moge_conv: built for one output block, last_res_blocks=0, last_conv_size=1 (MoGe's defaults)")
        for c in [dim_proj, last_conv_channels] + list(dim_upsample):
            if c % 32 != 0:
                raise NotImplementedError(f"moge_conv: channel count {c} must be a multiple of 32 (conv kernel tiles)")
        self.patch_size, self.dim_out, self.output_dim = patch_size, list(dim_out), sum(dim_out)
        self.dim_proj, self.dim_upsample, self.last_conv_channels = dim_proj, list(dim_upsample), last_conv_channels
        self.projects = nn.ModuleList([nn.Conv2d(c, dim_proj, 1) for c in input_feature_dims])
        dims_in = [dim_proj] + list(dim_upsample[:-1])
        self.upsample_blocks = nn.ModuleList(
            [
                nn.Sequential(
                    nn.Sequential(nn.ConvTranspose2d(cin + 2, cout, 2, 2), nn.Conv2d(cout, cout, 3, 1, 1, padding_mode="replicate")),
                    *[_ResConvBlockParams(cout, cout, dim_times_res_block_hidden * cout, res_block_norm) for _ in range(num_res_blocks)],
                )
                for cin, cout in zip(dims_in, dim_upsample)
            ]
        )
        self.output_block = nn.ModuleList(
            [
                nn.Sequential(
                    nn.Conv2d(dim_upsample[-1] + 2, last_conv_channels, 3, 1, 1, padding_mode="replicate"),
                    nn.ReLU(inplace=True),
                    nn.Conv2d(last_conv_channels, d, 1, 1, 0, padding_mode="replicate"),
                )
                for d in self.dim_out
            ]
        )


class _DoubleConvParams(_Holder):
    """unet_encoder.py:10-23: parameters of (Conv3x3 pad 1 -> ReLU) x 2 under the reference's names conv.0 / conv.2."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.conv = nn.ModuleList([nn.Conv2d(cin, cout, kernel_size=3, padding=1), nn.Identity(), nn.Conv2d(cout, cout, kernel_size=3, padding=1), nn.Identity()])


class UNetParams(_Holder):
    """Parameters of unet_encoder.py:26-71 with the reference's attribute names (state-dict keys ``downs.N.conv.{0,2}.*``,
    ``ups.{2k}.*`` = ConvTranspose2d(k=s=2), ``ups.{2k+1}.conv.{0,2}.*``, ``bottleneck.conv.{0,2}.*``, ``final_conv.*``).
    The forward pass is Engine._unet (implicit-GEMM conv kernels + ufm_maxpool2x2_nhwc / ufm_resize_nearest_nhwc)."""

    def __init__(self, in_channels: int, out_channels: int, features=(64, 128, 256, 512)):
        super().__init__()
        self.in_channels, self.out_channels, self.features = in_channels, out_channels, list(features)
        self.downs, self.ups = nn.ModuleList(), nn.ModuleList()
        c = in_channels
        for f in features:
            self.downs.append(_DoubleConvParams(c, f))
            c = f
        self.bottleneck = _DoubleConvParams(features[-1], features[-1] * 2)
        for f in reversed(features):
            self.ups.append(nn.ConvTranspose2d(f * 2, f, kernel_size=2, stride=2))
            self.ups.append(_DoubleConvParams(f * 2, f))
        self.final_conv = nn.Conv2d(features[0], out_channels, kernel_size=1)


class AdaptorSpec(_Holder):
    """Parameter-free output map; ``kind``/``scale``/``shift`` are consumed by ufm_head_tail."""

    def __init__(self, cls_name: str, name: str, **kw: Any):
        super().__init__()
        self.cls_name, self.name = cls_name, name
        if cls_name == "FlowAdaptor":
            self.required_channels = 2
            self.kinds = [0, 0]
            self.scale = [float(v) for v in kw.get("flow_std", (1.0, 1.0))]
            self.shift = [float(v) for v in kw.get("flow_mean", (0.0, 0.0))]
        elif cls_name == "MaskAdaptor":
            self.required_channels = 1
            self.kinds, self.scale, self.shift = [1], [1.0], [0.0]
        elif cls_name == "FlowWithConfidenceAdaptor":
            # selectable in the reference's adaptor table (ufm.py:38); its forward reads only ``["flow"].value`` (ufm.py:420, 645):
            # two flow channels through the tail's affine map, the third finished by ufm_adaptor_confidence ([U], parity unpinned)
            self.required_channels = 3
            self.kinds = [0, 0, 0]
            self.scale = [float(v) for v in kw.get("flow_std", (1.0, 1.0))] + [1.0]
            self.shift = [float(v) for v in kw.get("flow_mean", (0.0, 0.0))] + [0.0]
            ctype = kw.get("confidence_type", "exp")
            if ctype not in ("exp", "sigmoid", "identity"):
                raise ValueError(f"FlowWithConfidenceAdaptor: unknown confidence_type {ctype!r} (known: 'exp', 'sigmoid', 'identity')")
            self.confidence_type = {"exp": 0, "sigmoid": 1, "identity": 2}[ctype]
            self.vmin, self.vmax = float(kw.get("vmin", 1.0)), float(kw.get("vmax", float("inf")))
        elif cls_name == "Covariance2DAdaptor":
            # raw channels out of the tail kernel (kind 0, a = 1, d = 0); ufm_adaptor_covariance2d finishes them
            self.required_channels = 3
            self.kinds, self.scale, self.shift = [0, 0, 0], [1.0] * 3, [0.0] * 3
        elif cls_name == "ConfidenceAdaptor":
            self.required_channels = 1
            self.kinds, self.scale, self.shift = [0], [1.0], [0.0]
            ctype = kw.get("confidence_type", "exp")
            if ctype not in ("exp", "sigmoid", "identity"):
                raise ValueError(f"ConfidenceAdaptor: unknown confidence_type {ctype!r} (known: 'exp', 'sigmoid', 'identity')")
            self.confidence_type = {"exp": 0, "sigmoid": 1, "identity": 2}[ctype]
            self.vmin, self.vmax = float(kw.get("vmin", 1.0)), float(kw.get("vmax", float("inf")))
        else:
            raise NotImplementedError(f"adaptor {cls_name} has no call site on the UFM inference path (ufm.py:644-660)")


class AdaptorMap(_Holder):
    def __init__(self, *adaptors: AdaptorSpec):
        super().__init__()
        self.adaptors = nn.ModuleList(adaptors)


def make_head(head_type: str, feature_head_kwargs: Dict[str, Any], adaptors_kwargs: Dict[str, Any]) -> nn.Module:
    """ufm.py:243-289 -- Sequential(Sequential(DPTFeature, DPTRegressionProcessor), AdaptorMap(...))."""
    adaptors = [AdaptorSpec(cfg["class"], **cfg["kwargs"]) for cfg in adaptors_kwargs.values()]
    total = sum(a.required_channels for a in adaptors)
    if head_type == "moge_conv":
        feat = MoGeConvParams(**feature_head_kwargs)
        out_dim = feat.output_dim
    elif head_type == "dpt":
        feat = nn.Sequential(DPTFeatureParams(**feature_head_kwargs["dpt_feature"]), DPTProcessorParams(**feature_head_kwargs["dpt_processor"]))
        out_dim = feat[1].output_dim
    else:
        raise ValueError(f"Head type {head_type} not supported.")  # ufm.py:268-269
    if total != out_dim:
        raise ValueError(f"head produces {out_dim} channels, adaptors need {total}")
    return nn.Sequential(feat, AdaptorMap(*adaptors))


def init_weights_(model: nn.Module, seed: int = 0) -> nn.Module:
    """Deterministic CPU random init (SURVEY 8(d) "Weights"): O(1)-scaled so every path contributes.
    Same rule and same generator order as the oracle's initialiser, so both trees get identical
    weights from the same seed without sharing code."""
    import re

    g = torch.Generator().manual_seed(seed)
    convt = re.compile(r"act_(1|2)_postprocess\.1\.weight$")
    group_norm_weights = {f"{mn}.weight" for mn, m in model.named_modules() if isinstance(m, nn.GroupNorm)}  # (moge_conv head)
    with torch.no_grad():
        for name, p in sorted(model.named_parameters(), key=lambda kv: kv[0]):
            cpu = torch.empty(p.shape, dtype=torch.float32)
            if p.dim() >= 2 and "pos_embed" not in name and "cls_token" not in name:
                fan_in = p.shape[0] if convt.search(name) else p[0].numel()
                cpu = torch.randn(p.shape, generator=g) * (1.0 / fan_in**0.5)
            elif name.endswith("gamma"):
                cpu = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
            elif ("norm" in name and name.endswith("weight")) or name in group_norm_weights:
                cpu = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
            elif name.endswith("classification_bias"):
                cpu = 0.1 * torch.randn(p.shape, generator=g)
            elif "pos_embed" in name or "cls_token" in name:
                cpu = 0.5 * torch.randn(p.shape, generator=g)
            else:
                cpu = 0.02 * torch.randn(p.shape, generator=g)
            p.copy_(cpu.to(p.device))
    return model
