"""fp32 CPU restatement of the third-party ``uniception`` blocks UFM is built from.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  **Parity unpinned** for
everything in this file: the ``uniception`` package (github castacks/UniCeption,
a git submodule of the reference, ``/root/reference/.gitmodules:1-3``, version
NOT pinned anywhere in the reference) is absent from ``/root/reference``.  The
blocks are restated from the published architectures they implement (DINOv2
ViT, DUSt3R/MultiMAE DPT head, timm-style pre-LN transformer block).  What
pins this file is (i) the reference's own call sites, which fix every
constructor/IO name used here:

  * ``uniflowmatch/models/ufm.py:13-25``   imported names
  * ``ufm.py:187``        ``feature_returner_encoder_factory(encoder_str, **kw)``
  * ``ufm.py:193``        ``INFO_SHARING_CLASSES[name][1](**kw)`` (index 1 = the
                          intermediate-feature-returner variant)
  * ``ufm.py:262-273``    ``Sequential(Sequential(DPTFeature, DPTRegressionProcessor), AdaptorMap(*adaptors))``
  * ``ufm.py:308-315``    ``ViTEncoderInput(image=, data_norm_type=)`` -> list of objects with ``.features`` (BCHW)
  * ``ufm.py:390-409``    ``MultiViewTransformerInput(features=[f1, f2])`` -> ``(final, [inter0, inter1])`` each ``.features[view]``
  * ``ufm.py:449-453``    ``PredictionHeadLayeredInput(list_features=, target_output_shape=)``
  * ``ufm.py:645-659``    head output dict keyed by adaptor name with ``.value`` / ``.mask`` / ``.logits`` /
                          ``.covariance`` / ``.inv_covariance`` / ``.log_det``
  * ``ufm.py:961-965``    ``MLPFeature(...)(PredictionHeadInput(x)).decoded_channels``
  * ``base.py:75,183-229`` ``IMAGE_NORMALIZATION_DICT[name].mean/.std`` and ``encoder.data_norm_type``

and (ii) a cross-check of the encoder against ``transformers.Dinov2Model``
(``tests/test_oracle_encoder_vs_hf.py``).

State-dict key names follow the upstream module attribute names as recalled
(``encoder.model.*`` is implied by ``ufm.py:208-210``) so that a real checkpoint
would map one-to-one; this is documented, not verified.
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from functools import partial
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn.functional as F
from torch import nn

# --------------------------------------------------------------------------- #
# image normalisation table (base.py:75, :183-229 use .mean / .std 3-vectors)
# --------------------------------------------------------------------------- #


@dataclass
class ImageNormalization:
    mean: torch.Tensor
    std: torch.Tensor


_IMAGENET_MEAN = (0.485, 0.456, 0.406)
_IMAGENET_STD = (0.229, 0.224, 0.225)

IMAGE_NORMALIZATION_DICT: Dict[str, ImageNormalization] = {
    "dummy": ImageNormalization(torch.tensor([0.0, 0.0, 0.0]), torch.tensor([1.0, 1.0, 1.0])),
    "identity": ImageNormalization(torch.tensor([0.0, 0.0, 0.0]), torch.tensor([1.0, 1.0, 1.0])),
    "croco": ImageNormalization(torch.tensor(_IMAGENET_MEAN), torch.tensor(_IMAGENET_STD)),
    "dinov2": ImageNormalization(torch.tensor(_IMAGENET_MEAN), torch.tensor(_IMAGENET_STD)),
    "dust3r": ImageNormalization(torch.tensor([0.5, 0.5, 0.5]), torch.tensor([0.5, 0.5, 0.5])),
    "patch_embedder": ImageNormalization(torch.tensor([0.5, 0.5, 0.5]), torch.tensor([0.5, 0.5, 0.5])),
}


# --------------------------------------------------------------------------- #
# IO dataclasses (field names fixed by the reference call sites listed above)
# --------------------------------------------------------------------------- #


@dataclass
class ViTEncoderInput:
    image: torch.Tensor
    data_norm_type: Optional[str] = None


@dataclass
class ViTEncoderOutput:
    features: torch.Tensor


@dataclass
class MultiViewTransformerInput:
    features: List[torch.Tensor]
    additional_input_tokens: Optional[torch.Tensor] = None


@dataclass
class MultiViewTransformerOutput:
    features: List[torch.Tensor]
    additional_token_features: Optional[torch.Tensor] = None


@dataclass
class PredictionHeadInput:
    last_feature: torch.Tensor


@dataclass
class PredictionHeadLayeredInput:
    list_features: List[torch.Tensor]
    target_output_shape: Tuple[int, int]


@dataclass
class PixelTaskOutput:
    decoded_channels: torch.Tensor


@dataclass
class RegressionAdaptorOutput:
    value: torch.Tensor


@dataclass
class RegressionWithConfidenceAdaptorOutput:
    value: torch.Tensor
    confidence: torch.Tensor


@dataclass
class MaskAdaptorOutput:
    logits: torch.Tensor
    mask: torch.Tensor


@dataclass
class Covariance2DAdaptorOutput:
    covariance: torch.Tensor
    log_det: torch.Tensor
    inv_covariance: torch.Tensor


# --------------------------------------------------------------------------- #
# transformer building blocks (timm-style pre-LN block; DINOv2 == same block
# with LayerScale gamma).  LayerNorm eps 1e-6, exact-erf GELU, scale 1/sqrt(d).
# --------------------------------------------------------------------------- #


class RoPE2D(nn.Module):
    """CroCo-style 2-D rotary embedding ([UPSTREAM-RECALL] croco/models/pos_embed.py RoPE2D, the `custom_positional_encoding`
    of the cross-attention info-sharing variant): the head dim is split in two halves, the first rotated by the token's
    y index and the second by its x index; inside a half of width D2 the pairs are (i, i + D2/2) with
    theta_i = pos / freq^(2 i / D2)   (rotate_half form: out = t * cos + rotate_half(t) * sin).  Parity unpinned."""

    def __init__(self, freq: float = 100.0):
        super().__init__()
        self.base = float(freq)

    def tables(self, d2: int, positions_1d: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        inv_freq = 1.0 / (self.base ** (torch.arange(0, d2, 2, dtype=torch.float64) / d2))
        f = positions_1d.double().unsqueeze(-1) * inv_freq  # (..., d2/2)
        f = torch.cat((f, f), dim=-1)
        return f.cos().float(), f.sin().float()

    @staticmethod
    def rotate_half(x: torch.Tensor) -> torch.Tensor:
        x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2 :]
        return torch.cat((-x2, x1), dim=-1)

    def forward(self, tokens: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        """tokens (B, H, N, D); positions (B, N, 2) integer (y, x)."""
        d2 = tokens.shape[-1] // 2
        out = []
        for half, axis in ((tokens[..., :d2], 0), (tokens[..., d2:], 1)):
            cos, sin = self.tables(d2, positions[..., axis])  # (B, N, d2)
            cos, sin = cos[:, None].to(tokens.dtype), sin[:, None].to(tokens.dtype)
            out.append(half * cos + self.rotate_half(half) * sin)
        return torch.cat(out, dim=-1)


def grid_positions(b: int, h: int, w: int) -> torch.Tensor:
    """(B, h*w, 2) integer (y, x) of every token of a row-major h x w grid ([UPSTREAM-RECALL] PositionGetter)."""
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    return torch.stack((ys, xs), dim=-1).reshape(1, h * w, 2).expand(b, -1, -1)


class Attention(nn.Module):
    def __init__(self, dim: int, num_heads: int, qkv_bias: bool = True, proj_bias: bool = True, rope: Optional[nn.Module] = None):
        super().__init__()
        assert dim % num_heads == 0
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim**-0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)
        self.rope = rope

    def forward(self, x: torch.Tensor, xpos: Optional[torch.Tensor] = None) -> torch.Tensor:
        b, n, c = x.shape
        qkv = self.qkv(x).reshape(b, n, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        if self.rope is not None:
            q, k = self.rope(q, xpos), self.rope(k, xpos)
        # explicit softmax(q k^T * scale) v ; kept explicit (not SDPA) so the
        # oracle has no dependence on a fused kernel's internal choices.
        if n <= 4096:
            attn = (q * self.scale) @ k.transpose(-2, -1)
            attn = attn.softmax(dim=-1)
            out = attn @ v
        else:  # long sequences: SDPA's math is the same contraction, tiled
            out = F.scaled_dot_product_attention(q, k, v, scale=self.scale)
        out = out.transpose(1, 2).reshape(b, n, c)
        return self.proj(out)


class LayerScale(nn.Module):
    def __init__(self, dim: int, init_values: float = 1e-5):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x * self.gamma


class Mlp(nn.Module):
    def __init__(self, in_features: int, hidden_features: int, out_features: Optional[int] = None, bias: bool = True):
        super().__init__()
        out_features = out_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = nn.GELU()  # exact erf form
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.fc2(self.act(self.fc1(x)))


class Block(nn.Module):
    """x += ls1(attn(norm1(x))) ; x += ls2(mlp(norm2(x)))"""

    def __init__(
        self,
        dim: int,
        num_heads: int,
        mlp_ratio: float = 4.0,
        qkv_bias: bool = True,
        init_values: Optional[float] = None,
        eps: float = 1e-6,
    ):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads, qkv_bias=qkv_bias)
        self.ls1 = LayerScale(dim, init_values) if init_values is not None else nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.ls2 = LayerScale(dim, init_values) if init_values is not None else nn.Identity()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x + self.ls1(self.attn(self.norm1(x)))
        x = x + self.ls2(self.mlp(self.norm2(x)))
        return x


# --------------------------------------------------------------------------- #
# DINOv2 ViT (hub ``DinoVisionTransformer`` layout, no registers, mask_token
# deleted as ufm.py:208-210 implies).
# --------------------------------------------------------------------------- #

_DINOV2_SIZES = {
    "small": dict(embed_dim=384, depth=12, num_heads=6),
    "base": dict(embed_dim=768, depth=12, num_heads=12),
    "large": dict(embed_dim=1024, depth=24, num_heads=16),
}


class PatchEmbed(nn.Module):
    def __init__(self, patch_size: int, in_chans: int, embed_dim: int):
        super().__init__()
        self.patch_size = patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.proj(x)  # B D h w
        return x.flatten(2).transpose(1, 2)  # B hw D


class DinoVisionTransformerRef(nn.Module):
    """Attribute names mirror the hub model so state-dict keys are ``blocks.{i}.…``."""

    def __init__(
        self,
        img_size: int = 518,
        patch_size: int = 14,
        embed_dim: int = 1024,
        depth: int = 24,
        num_heads: int = 16,
        mlp_ratio: float = 4.0,
        init_values: Optional[float] = 1.0,
        interpolate_offset: float = 0.1,
    ):
        super().__init__()
        self.patch_size = patch_size
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.interpolate_offset = interpolate_offset
        self.patch_embed = PatchEmbed(patch_size, 3, embed_dim)
        g = img_size // patch_size
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, g * g + 1, embed_dim))
        self.blocks = nn.ModuleList(
            [Block(embed_dim, num_heads, mlp_ratio, True, init_values) for _ in range(depth)]
        )
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)

    # hub semantics: identity at the native grid, otherwise bicubic with the
    # "+offset" scale-factor form (the build picks hub semantics).
    def interpolate_pos_encoding(self, npatch: int, h: int, w: int) -> torch.Tensor:
        n_native = self.pos_embed.shape[1] - 1
        if npatch == n_native and h == w:
            return self.pos_embed
        pe = self.pos_embed.float()
        cls_pe, patch_pe = pe[:, :1], pe[:, 1:]
        dim = pe.shape[-1]
        h0, w0 = h // self.patch_size, w // self.patch_size
        m = int(round(math.sqrt(n_native)))
        assert m * m == n_native
        kwargs: Dict[str, Any] = {}
        if self.interpolate_offset:
            sx = float(w0 + self.interpolate_offset) / m
            sy = float(h0 + self.interpolate_offset) / m
            kwargs["scale_factor"] = (sy, sx)
        else:
            kwargs["size"] = (h0, w0)
        patch_pe = F.interpolate(
            patch_pe.reshape(1, m, m, dim).permute(0, 3, 1, 2), mode="bicubic", antialias=False, **kwargs
        )
        assert (h0, w0) == tuple(patch_pe.shape[-2:])
        patch_pe = patch_pe.permute(0, 2, 3, 1).reshape(1, -1, dim)
        return torch.cat((cls_pe, patch_pe), dim=1)

    def prepare_tokens(self, x: torch.Tensor) -> torch.Tensor:
        b, _, h, w = x.shape
        x = self.patch_embed(x)
        x = torch.cat((self.cls_token.expand(b, -1, -1), x), dim=1)
        return x + self.interpolate_pos_encoding(x.shape[1] - 1, h, w)

    def get_intermediate_layers(self, x: torch.Tensor, indices: Sequence[int], norm: bool = True) -> List[torch.Tensor]:
        b, _, h, w = x.shape
        x = self.prepare_tokens(x)
        outs = []
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if i in indices:
                outs.append(x)
        assert len(outs) == len(indices)
        if norm:
            outs = [self.norm(o) for o in outs]
        outs = [o[:, 1:] for o in outs]  # drop cls
        gh, gw = h // self.patch_size, w // self.patch_size
        return [o.reshape(b, gh, gw, -1).permute(0, 3, 1, 2).contiguous() for o in outs]


class DINOv2IntermediateFeatureReturner(nn.Module):
    """``feature_returner_encoder_factory("dinov2", **kw)`` (ufm.py:187).

    ``indices``: block indices whose (normed, cls-dropped, BCHW) outputs are
    returned, in order; ``[-1]`` of the returned list feeds info-sharing
    (ufm.py:596), ``[0]`` feeds the Refine classification head (ufm.py:954-959).
    Non-upstream keys accepted for tiny test configs: ``embed_dim``, ``depth``,
    ``num_heads``, ``mlp_ratio``, ``init_values``, ``img_size``.
    """

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
        assert not with_registers, "register tokens are not on the UFM hot path"
        self.name = name
        self.data_norm_type = data_norm_type
        self.patch_size = patch_size
        dims = dict(_DINOV2_SIZES[size])
        for k in ("embed_dim", "depth", "num_heads"):
            if k in kw:
                dims[k] = kw[k]
        self.model = DinoVisionTransformerRef(
            img_size=kw.get("img_size", 518),
            patch_size=patch_size,
            mlp_ratio=kw.get("mlp_ratio", 4.0),
            init_values=kw.get("init_values", 1.0),
            **dims,
        )
        if keep_first_n_layers is not None:
            self.model.blocks = self.model.blocks[:keep_first_n_layers]
        depth = len(self.model.blocks)
        if indices is None:
            indices = [depth - 1]
        if isinstance(indices, int):
            indices = list(range(depth - indices, depth))
        self.indices = [i % depth for i in indices]
        self.norm_intermediate = norm_intermediate
        self.enc_embed_dim = dims["embed_dim"]

    def forward(self, encoder_input: ViTEncoderInput) -> List[ViTEncoderOutput]:
        assert encoder_input.data_norm_type == self.data_norm_type, (
            f"encoder expects {self.data_norm_type}, got {encoder_input.data_norm_type}"
        )
        img = encoder_input.image
        assert img.shape[-2] % self.patch_size == 0 and img.shape[-1] % self.patch_size == 0
        feats = self.model.get_intermediate_layers(img, self.indices, norm=self.norm_intermediate)
        return [ViTEncoderOutput(features=f) for f in feats]


def feature_returner_encoder_factory(encoder_str: str, **kwargs: Any) -> nn.Module:
    if encoder_str != "dinov2":
        raise ValueError(f"oracle only restates the dinov2 encoder, got {encoder_str!r}")
    return DINOv2IntermediateFeatureReturner(**kwargs)


# --------------------------------------------------------------------------- #
# multi-view global-attention transformer ("global_attention", ufm.py:491)
# --------------------------------------------------------------------------- #


def sinusoid_view_table(n_position: int, dim: int, base: float = 10000.0) -> torch.Tensor:
    pos = torch.arange(n_position, dtype=torch.float64).unsqueeze(1)
    i = torch.arange(dim, dtype=torch.float64).unsqueeze(0)
    angle = pos / torch.pow(torch.tensor(base, dtype=torch.float64), 2.0 * torch.div(i, 2, rounding_mode="floor") / dim)
    table = torch.zeros(n_position, dim, dtype=torch.float64)
    table[:, 0::2] = torch.sin(angle[:, 0::2])
    table[:, 1::2] = torch.cos(angle[:, 1::2])
    return table.float()


class MultiViewGlobalAttentionTransformerIFR(nn.Module):
    """All views' tokens concatenated, ``depth`` joint self-attention blocks.

    Returns ``(final, [intermediates at self.indices])``; every output is
    ``MultiViewTransformerOutput(features=[BCHW per view])`` (ufm.py:598-615).
    """

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
            preset = {"base": (12, 768, 12), "large": (24, 1024, 16)}[size]
            depth, dim, num_heads = preset
        self.name = name
        self.input_embed_dim = input_embed_dim
        self.max_num_views = max_num_views
        self.depth = depth
        self.dim = dim
        self.num_heads = num_heads
        self.proj_embed = nn.Linear(input_embed_dim, dim, bias=True) if input_embed_dim != dim else nn.Identity()
        self.self_attention_blocks = nn.ModuleList(
            [Block(dim, num_heads, mlp_ratio, qkv_bias, init_values) for _ in range(depth)]
        )
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.register_buffer("view_pos_table", sinusoid_view_table(max_num_views, dim), persistent=False)
        if indices is None:
            indices = [depth // 2 - 1, (3 * depth) // 4 - 1]
        self.indices = list(indices)
        self.norm_intermediate = norm_intermediate

    def _split(self, x: torch.Tensor, b: int, v: int, h: int, w: int) -> List[torch.Tensor]:
        x = x.reshape(b, v, h, w, self.dim).permute(1, 0, 4, 2, 3)
        return [x[i].contiguous() for i in range(v)]

    def forward(self, model_input: MultiViewTransformerInput):
        feats = model_input.features
        v = len(feats)
        assert v <= self.max_num_views
        b, c, h, w = feats[0].shape
        assert all(f.shape == feats[0].shape for f in feats)
        assert c == self.input_embed_dim
        x = torch.stack(feats, dim=1).permute(0, 1, 3, 4, 2).reshape(b, v * h * w, c)
        x = self.proj_embed(x)
        pe = self.view_pos_table[:v].to(x.dtype)  # reference view = index 0, others 1..v-1
        x = (x.reshape(b, v, h * w, self.dim) + pe.view(1, v, 1, self.dim)).reshape(b, v * h * w, self.dim)
        inter: List[MultiViewTransformerOutput] = []
        for i, blk in enumerate(self.self_attention_blocks):
            x = blk(x)
            if i in self.indices:
                xi = self.norm(x) if self.norm_intermediate else x
                inter.append(MultiViewTransformerOutput(features=self._split(xi, b, v, h, w)))
        final = MultiViewTransformerOutput(features=self._split(self.norm(x), b, v, h, w))
        return final, inter


# --------------------------------------------------------------------------- #
# multi-view cross-attention transformer ("cross_attention"): CroCo / DUSt3R decoder
# layout -- one branch of blocks per view; every block = self-attention, cross-attention
# to the OTHER views' tokens of the previous layer, MLP.  [UPSTREAM-RECALL]; parity unpinned.
# --------------------------------------------------------------------------- #


class CrossAttention(nn.Module):
    def __init__(self, dim: int, num_heads: int, qkv_bias: bool = True, rope: Optional[nn.Module] = None):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim**-0.5
        self.projq = nn.Linear(dim, dim, bias=qkv_bias)
        self.projk = nn.Linear(dim, dim, bias=qkv_bias)
        self.projv = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.rope = rope

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, qpos=None, kpos=None) -> torch.Tensor:
        b, nq, c = query.shape
        nk = key.shape[1]
        q = self.projq(query).reshape(b, nq, self.num_heads, self.head_dim).permute(0, 2, 1, 3)
        k = self.projk(key).reshape(b, nk, self.num_heads, self.head_dim).permute(0, 2, 1, 3)
        v = self.projv(value).reshape(b, nk, self.num_heads, self.head_dim).permute(0, 2, 1, 3)
        if self.rope is not None:
            q, k = self.rope(q, qpos), self.rope(k, kpos)
        attn = ((q * self.scale) @ k.transpose(-2, -1)).softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(b, nq, c))


class CrossAttentionBlock(nn.Module):
    """x += ls1(attn(norm1 x)); x += ls2(cross_attn(norm2 x, norm_y y, norm_y y)); x += ls3(mlp(norm3 x))"""

    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0, qkv_bias: bool = True, init_values: Optional[float] = None,
                 norm_cross_tokens: bool = True, rope: Optional[nn.Module] = None, eps: float = 1e-6):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads, qkv_bias=qkv_bias, rope=rope)
        self.ls1 = LayerScale(dim, init_values) if init_values is not None else nn.Identity()
        self.norm_y = nn.LayerNorm(dim, eps=eps) if norm_cross_tokens else nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.cross_attn = CrossAttention(dim, num_heads, qkv_bias=qkv_bias, rope=rope)
        self.ls2 = LayerScale(dim, init_values) if init_values is not None else nn.Identity()
        self.norm3 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.ls3 = LayerScale(dim, init_values) if init_values is not None else nn.Identity()

    def forward(self, x: torch.Tensor, y: torch.Tensor, xpos=None, ypos=None) -> torch.Tensor:
        x = x + self.ls1(self.attn(self.norm1(x), xpos))
        y_ = self.norm_y(y)
        x = x + self.ls2(self.cross_attn(self.norm2(x), y_, y_, xpos, ypos))
        x = x + self.ls3(self.mlp(self.norm3(x)))
        return x


class MultiViewCrossAttentionTransformerIFR(nn.Module):
    """``INFO_SHARING_CLASSES["cross_attention"][1]``.  Every view has its own branch of ``depth`` blocks; at layer l each
    view's tokens attend to themselves and to the concatenated layer-(l-1) tokens of all OTHER views (both views are
    updated from the previous layer's outputs, as in DUSt3R's two decoders).  ``rope_freq``: enable the CroCo RoPE-2D on
    q / k of both attentions (upstream passes a callable `custom_positional_encoding`, which a JSON config cannot hold).
    Same IO contract as the global-attention variant (ufm.py:598-615)."""

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
        self.name, self.input_embed_dim, self.num_views = name, input_embed_dim, num_views
        self.depth, self.dim, self.num_heads = depth, dim, num_heads
        self.proj_embed = nn.Linear(input_embed_dim, dim, bias=True) if input_embed_dim != dim else nn.Identity()
        self.rope = RoPE2D(rope_freq) if rope_freq else None
        self.multi_view_branches = nn.ModuleList(
            [
                nn.ModuleList([CrossAttentionBlock(dim, num_heads, mlp_ratio, qkv_bias, init_values, norm_cross_tokens, self.rope) for _ in range(depth)])
                for _ in range(num_views)
            ]
        )
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        if indices is None:
            indices = [depth // 2 - 1, (3 * depth) // 4 - 1]
        self.indices = list(indices)
        self.norm_intermediate = norm_intermediate

    def forward(self, model_input: MultiViewTransformerInput):
        feats = model_input.features
        v = len(feats)
        assert v == self.num_views
        b, c, h, w = feats[0].shape
        assert all(f.shape == feats[0].shape for f in feats) and c == self.input_embed_dim
        xs = [self.proj_embed(f.permute(0, 2, 3, 1).reshape(b, h * w, c)) for f in feats]
        pos = grid_positions(b, h, w) if self.rope is not None else None
        opos = torch.cat([pos] * (v - 1), dim=1) if pos is not None else None

        def to_maps(tokens):
            return [t.reshape(b, h, w, self.dim).permute(0, 3, 1, 2).contiguous() for t in tokens]

        inter: List[MultiViewTransformerOutput] = []
        for layer in range(self.depth):
            prev = xs
            xs = [
                self.multi_view_branches[i][layer](prev[i], torch.cat([prev[j] for j in range(v) if j != i], dim=1), pos, opos)
                for i in range(v)
            ]
            if layer in self.indices:
                inter.append(MultiViewTransformerOutput(features=to_maps([self.norm(t) if self.norm_intermediate else t for t in xs])))
        final = MultiViewTransformerOutput(features=to_maps([self.norm(t) for t in xs]))
        return final, inter


class _NoIFRVariant(nn.Module):
    def __init__(self, *a: Any, **k: Any):
        raise NotImplementedError("only the intermediate-feature-returner variant is on the UFM path (ufm.py:193)")


INFO_SHARING_CLASSES = {
    "global_attention": (_NoIFRVariant, MultiViewGlobalAttentionTransformerIFR),
    "cross_attention": (_NoIFRVariant, MultiViewCrossAttentionTransformerIFR),
}


# --------------------------------------------------------------------------- #
# DPT feature head + regression processor (DUSt3R / MultiMAE DPT layout)
# --------------------------------------------------------------------------- #


class ResidualConvUnit(nn.Module):
    def __init__(self, features: int):
        super().__init__()
        self.conv1 = nn.Conv2d(features, features, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(features, features, 3, 1, 1, bias=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        out = self.conv1(F.relu(x))
        out = self.conv2(F.relu(out))
        return out + x


class FeatureFusionBlock(nn.Module):
    def __init__(self, features: int):
        super().__init__()
        self.out_conv = nn.Conv2d(features, features, 1, 1, 0, bias=True)
        self.resConfUnit1 = ResidualConvUnit(features)
        self.resConfUnit2 = ResidualConvUnit(features)

    def forward(self, *xs: torch.Tensor) -> torch.Tensor:
        out = xs[0]
        if len(xs) == 2:
            out = out + self.resConfUnit1(xs[1])
        out = self.resConfUnit2(out)
        out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
        return self.out_conv(out)


class _Scratch(nn.Module):
    pass


class DPTFeature(nn.Module):
    """4 token grids -> (B, feature_dim, 8g, 8g) for patch 14 (stride-2 map)."""

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
        assert not use_bn and output_width_ratio == 1
        if isinstance(input_feature_dims, int):
            input_feature_dims = [input_feature_dims] * 4
        self.patch_size = patch_size
        self.hooks = list(hooks)
        ld = list(layer_dims)
        idim = list(input_feature_dims)
        self.scratch = _Scratch()
        for i in range(4):
            setattr(self.scratch, f"layer{i + 1}_rn", nn.Conv2d(ld[i], feature_dim, 3, 1, 1, bias=False))
            setattr(self.scratch, f"refinenet{i + 1}", FeatureFusionBlock(feature_dim))
        self.act_1_postprocess = nn.Sequential(
            nn.Conv2d(idim[0], ld[0], 1), nn.ConvTranspose2d(ld[0], ld[0], 4, 4, 0, bias=True)
        )
        self.act_2_postprocess = nn.Sequential(
            nn.Conv2d(idim[1], ld[1], 1), nn.ConvTranspose2d(ld[1], ld[1], 2, 2, 0, bias=True)
        )
        self.act_3_postprocess = nn.Sequential(nn.Conv2d(idim[2], ld[2], 1))
        self.act_4_postprocess = nn.Sequential(nn.Conv2d(idim[3], ld[3], 1), nn.Conv2d(ld[3], ld[3], 3, 2, 1))
        self.act_postprocess = nn.ModuleList(
            [self.act_1_postprocess, self.act_2_postprocess, self.act_3_postprocess, self.act_4_postprocess]
        )

    def forward(self, head_input: PredictionHeadLayeredInput) -> PredictionHeadLayeredInput:
        feats = [head_input.list_features[h] for h in self.hooks]
        layers = [self.act_postprocess[i](f) for i, f in enumerate(feats)]
        layers = [getattr(self.scratch, f"layer{i + 1}_rn")(l) for i, l in enumerate(layers)]
        p4 = self.scratch.refinenet4(layers[3])[:, :, : layers[2].shape[2], : layers[2].shape[3]]
        p3 = self.scratch.refinenet3(p4, layers[2])
        p2 = self.scratch.refinenet2(p3, layers[1])
        p1 = self.scratch.refinenet1(p2, layers[0])
        return PredictionHeadLayeredInput(list_features=[p1], target_output_shape=head_input.target_output_shape)


class DPTRegressionProcessor(nn.Module):
    def __init__(self, input_feature_dim: int = 256, output_dim: int = 2, hidden_dims: Optional[Sequence[int]] = None, **_: Any):
        super().__init__()
        if hidden_dims is None:
            hidden_dims = [input_feature_dim // 2, 32]
        self.conv1 = nn.Conv2d(input_feature_dim, hidden_dims[0], 3, 1, 1)
        self.conv2 = nn.Sequential(
            nn.Conv2d(hidden_dims[0], hidden_dims[1], 3, 1, 1),
            nn.ReLU(True),
            nn.Conv2d(hidden_dims[1], output_dim, 1, 1, 0),
        )

    def forward(self, head_input: PredictionHeadLayeredInput) -> PixelTaskOutput:
        x = self.conv1(head_input.list_features[0])
        x = F.interpolate(x, size=tuple(head_input.target_output_shape), mode="bilinear", align_corners=True)
        return PixelTaskOutput(decoded_channels=self.conv2(x))


class MLPFeature(nn.Module):
    """Per-token MLP -> patch_size**2 * output_dim -> pixel-shuffle to full res (ufm.py:961-965)."""

    def __init__(self, input_feature_dim: int, patch_size: int, output_dim: int, mlp_ratio: float = 4.0, **_: Any):
        super().__init__()
        self.patch_size = patch_size
        self.output_dim = output_dim
        self.mlp = Mlp(input_feature_dim, int(mlp_ratio * input_feature_dim), output_dim * patch_size * patch_size)

    def forward(self, head_input: PredictionHeadInput) -> PixelTaskOutput:
        x = head_input.last_feature  # B C h w
        x = self.mlp(x.permute(0, 2, 3, 1))  # B h w (C_out p p)
        x = F.pixel_shuffle(x.permute(0, 3, 1, 2), self.patch_size)
        return PixelTaskOutput(decoded_channels=x)


def normalized_view_plane_uv(width: int, height: int, aspect_ratio: float) -> torch.Tensor:
    """(H, W, 2) view-plane coordinates of pixel centres, the image diagonal spanning [-1, 1]^2-ish
    ([UPSTREAM-RECALL] MoGe utils.geometry_torch.normalized_view_plane_uv): u along x, v along y."""
    span_x = aspect_ratio / (1 + aspect_ratio**2) ** 0.5
    span_y = 1 / (1 + aspect_ratio**2) ** 0.5
    u = torch.linspace(-span_x * (width - 1) / width, span_x * (width - 1) / width, width, dtype=torch.float32)
    v = torch.linspace(-span_y * (height - 1) / height, span_y * (height - 1) / height, height, dtype=torch.float32)
    u, v = torch.meshgrid(u, v, indexing="xy")
    return torch.stack([u, v], dim=-1)


class ResidualConvBlock(nn.Module):
    """GroupNorm -> ReLU -> conv3x3 (replicate padding) -> GroupNorm -> ReLU -> conv3x3 (replicate) + skip
    ([UPSTREAM-RECALL] MoGe ResidualConvBlock; 'group_norm' = 32 channels per group, 'layer_norm' = one group)."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None, hidden_channels: Optional[int] = None, norm: str = "group_norm"):
        super().__init__()
        out_channels = out_channels or in_channels
        hidden_channels = hidden_channels or in_channels
        groups = (lambda ch: max(ch // 32, 1)) if norm == "group_norm" else (lambda ch: 1)
        self.layers = nn.Sequential(
            nn.GroupNorm(groups(in_channels), in_channels),
            nn.ReLU(),
            nn.Conv2d(in_channels, hidden_channels, 3, padding=1, padding_mode="replicate"),
            nn.GroupNorm(groups(hidden_channels), hidden_channels),
            nn.ReLU(),
            nn.Conv2d(hidden_channels, out_channels, 3, padding=1, padding_mode="replicate"),
        )
        self.skip_connection = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else nn.Identity()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.layers(x) + self.skip_connection(x)


class MoGeConvFeature(nn.Module):
    """``head_type="moge_conv"`` (ufm.py:266-267): the convolutional head of MoGe ([UPSTREAM-RECALL] MoGe v1 `Head`,
    wrapped by uniception as MoGeConvFeature).  Per-level 1x1 projections summed at the token grid; three x2 stages
    {concat view-plane uv, ConvTranspose2d(k=s=2), conv3x3 replicate, residual conv blocks}; bilinear
    (align_corners=False) to the target shape; concat uv; per-output block conv3x3 -> ReLU -> conv.  Parity unpinned."""

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
        assert len(input_feature_dims) == num_features
        self.patch_size = patch_size
        self.dim_out = list(dim_out)
        self.projects = nn.ModuleList([nn.Conv2d(c, dim_proj, 1) for c in input_feature_dims])
        dims_in = [dim_proj] + list(dim_upsample[:-1])
        self.upsample_blocks = nn.ModuleList(
            [
                nn.Sequential(
                    nn.Sequential(nn.ConvTranspose2d(cin + 2, cout, 2, 2), nn.Conv2d(cout, cout, 3, 1, 1, padding_mode="replicate")),
                    *[ResidualConvBlock(cout, cout, dim_times_res_block_hidden * cout, res_block_norm) for _ in range(num_res_blocks)],
                )
                for cin, cout in zip(dims_in, dim_upsample)
            ]
        )
        self.output_block = nn.ModuleList(
            [
                nn.Sequential(
                    nn.Conv2d(dim_upsample[-1] + 2, last_conv_channels, 3, 1, 1, padding_mode="replicate"),
                    *[ResidualConvBlock(last_conv_channels, last_conv_channels, dim_times_res_block_hidden * last_conv_channels, res_block_norm) for _ in range(last_res_blocks)],
                    nn.ReLU(inplace=True),
                    nn.Conv2d(last_conv_channels, d, last_conv_size, 1, last_conv_size // 2, padding_mode="replicate"),
                )
                for d in self.dim_out
            ]
        )

    def forward(self, head_input: PredictionHeadLayeredInput) -> PixelTaskOutput:
        img_h, img_w = head_input.target_output_shape
        x = torch.stack([proj(f) for proj, f in zip(self.projects, head_input.list_features)], dim=1).sum(dim=1)
        for block in self.upsample_blocks:
            uv = normalized_view_plane_uv(x.shape[-1], x.shape[-2], img_w / img_h).to(x)
            x = torch.cat([x, uv.permute(2, 0, 1).unsqueeze(0).expand(x.shape[0], -1, -1, -1)], dim=1)
            x = block(x)
        x = F.interpolate(x, (img_h, img_w), mode="bilinear", align_corners=False)
        uv = normalized_view_plane_uv(img_w, img_h, img_w / img_h).to(x)
        x = torch.cat([x, uv.permute(2, 0, 1).unsqueeze(0).expand(x.shape[0], -1, -1, -1)], dim=1)
        return PixelTaskOutput(decoded_channels=torch.cat([blk(x) for blk in self.output_block], dim=1))


# --------------------------------------------------------------------------- #
# adaptors (parameter-free output maps) and the AdaptorMap fan-out
# --------------------------------------------------------------------------- #


class FlowAdaptor(nn.Module):
    """2 channels -> flow in pixels: ``x * flow_std + flow_mean`` (defaults identity)."""

    def __init__(self, name: str, flow_mean: Sequence[float] = (0.0, 0.0), flow_std: Sequence[float] = (1.0, 1.0), **_: Any):
        super().__init__()
        self.name = name
        self.required_channels = 2
        self.register_buffer("flow_mean", torch.tensor(list(flow_mean), dtype=torch.float32).view(1, 2, 1, 1), persistent=False)
        self.register_buffer("flow_std", torch.tensor(list(flow_std), dtype=torch.float32).view(1, 2, 1, 1), persistent=False)

    def forward(self, x: torch.Tensor) -> RegressionAdaptorOutput:
        return RegressionAdaptorOutput(value=x * self.flow_std + self.flow_mean)


class MaskAdaptor(nn.Module):
    def __init__(self, name: str, **_: Any):
        super().__init__()
        self.name = name
        self.required_channels = 1

    def forward(self, x: torch.Tensor) -> MaskAdaptorOutput:
        return MaskAdaptorOutput(logits=x, mask=torch.sigmoid(x))


class ConfidenceAdaptor(nn.Module):
    def __init__(self, name: str, confidence_type: str = "exp", vmin: float = 1.0, vmax: float = float("inf"), **_: Any):
        super().__init__()
        self.name = name
        self.required_channels = 1
        self.confidence_type, self.vmin, self.vmax = confidence_type, vmin, vmax

    def forward(self, x: torch.Tensor) -> RegressionAdaptorOutput:
        if self.confidence_type == "exp":
            v = (self.vmin + x.exp()).clip(max=self.vmax)
        elif self.confidence_type == "sigmoid":
            v = (self.vmax - self.vmin) * torch.sigmoid(x) + self.vmin
        else:
            v = x
        return RegressionAdaptorOutput(value=v)


class Covariance2DAdaptor(nn.Module):
    """3 channels (log-sigma_x, log-sigma_y, atanh-rho) -> 2x2 covariance packed as [xx, yy, xy]."""

    def __init__(self, name: str, **_: Any):
        super().__init__()
        self.name = name
        self.required_channels = 3

    def forward(self, x: torch.Tensor) -> Covariance2DAdaptorOutput:
        sx, sy, rho = x[:, 0:1].exp(), x[:, 1:2].exp(), torch.tanh(x[:, 2:3]) * 0.99
        cxx, cyy, cxy = sx * sx, sy * sy, rho * sx * sy
        det = cxx * cyy - cxy * cxy
        inv = torch.cat([cyy / det, cxx / det, -cxy / det], dim=1)
        return Covariance2DAdaptorOutput(covariance=torch.cat([cxx, cyy, cxy], dim=1), log_det=det.log(), inv_covariance=inv)


class FlowWithConfidenceAdaptor(nn.Module):
    def __init__(self, name: str, **kw: Any):
        super().__init__()
        self.name = name
        self.required_channels = 3
        self.flow = FlowAdaptor(name, **{k: v for k, v in kw.items() if k.startswith("flow_")})
        self.conf = ConfidenceAdaptor(name, **{k: v for k, v in kw.items() if not k.startswith("flow_")})

    def forward(self, x: torch.Tensor) -> RegressionWithConfidenceAdaptorOutput:
        # the reference reads only ``.value`` of its "flow" entry (ufm.py:420, 645, 924); the confidence channel rides along
        return RegressionWithConfidenceAdaptorOutput(value=self.flow(x[:, :2]).value, confidence=self.conf(x[:, 2:3]).value)


class AdaptorMap(nn.Module):
    """Splits decoded channels between adaptors in order; returns {adaptor.name: output}."""

    def __init__(self, *adaptors: nn.Module):
        super().__init__()
        self.adaptors = nn.ModuleList(adaptors)

    def forward(self, head_output: PixelTaskOutput) -> Dict[str, Any]:
        x = head_output.decoded_channels
        total = sum(a.required_channels for a in self.adaptors)
        assert x.shape[1] == total, f"head produced {x.shape[1]} channels, adaptors need {total}"
        out, c0 = {}, 0
        for a in self.adaptors:
            out[a.name] = a(x[:, c0 : c0 + a.required_channels])
            c0 += a.required_channels
        return out
