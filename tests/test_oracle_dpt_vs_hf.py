"""Cross-check of the oracle's restated DPT head (``oracle/uniception_ref.py::DPTFeature`` + ``DPTRegressionProcessor``, the
classes the reference instantiates at ``/root/reference/uniflowmatch/models/ufm.py:243-289`` from the absent ``uniception``
package) against an INDEPENDENT implementation of the published DPT neck that is importable here:
``transformers.models.dpt.modeling_dpt`` (``DPTNeck`` = ``DPTReassembleStage`` + the 3x3 ``convs`` + ``DPTFeatureFusionStage``,
``DPTDepthEstimationHead``), built offline from a ``DPTConfig`` object -- no model name, no fetch.  Also one info-sharing
transformer block against ``Dinov2Layer``.  CPU only.

What matches structurally (and is therefore pinned by this test to <= 2e-5):
  * reassemble: 1x1 projection, then ConvTranspose(k = s = 4), ConvTranspose(k = s = 2), identity, 3x3 stride-2 conv
    (factors 4, 2, 1, 1/2);
  * one bias-free 3x3 conv per level into the fusion width (``scratch.layer{i}_rn``);
  * fusion, coarse to fine: pre-activation residual units (ReLU -> 3x3 -> ReLU -> 3x3, + input), the coarsest block using
    only its second unit, bilinear x2 with ``align_corners=True``, then the 1x1 ``out_conv``;
  * the regression processor: 3x3 (F -> F/2), bilinear to the target size with ``align_corners=True``, 3x3 (-> 32), ReLU,
    1x1.
Where the two differ, and what the test does about it:
  * readout: HF takes token sequences WITH a cls token and (``readout_type="ignore"``) drops it; the oracle's head is handed
    BCHW maps without cls (``ufm.py:602-630``).  The test prepends a dummy cls row for HF.
  * backbone width: HF uses ONE hidden size for all four levels; UFM-Base feeds [1024, 768, 768, 768].  The test uses equal
    widths (the per-level projection is an ordinary 1x1 conv either way).
  * odd grids: when a x2-upsampled map is one pixel larger than the next skip (37 -> 19 -> 38), the oracle crops the
    coarse path, HF re-interpolates the skip (``align_corners=False``).  The test uses an even grid (8 x 8 tokens), where
    neither path is taken; the crop convention itself stays "restated from recall".
  * HF's depth head ends in a ReLU and always has ONE output channel and a fixed x2 upsample: compared before that ReLU, with
    ``output_dim=1`` and a target of twice the fused map.
"""

import warnings

import pytest
import torch

from oracle import uniception_ref as U
from oracle.ufm_ref import init_weights_


def _neck_state_from_ref(ref: U.DPTFeature):
    r = ref.state_dict()
    sd = {}
    for i in range(4):
        a = f"act_{i + 1}_postprocess."
        q = f"reassemble_stage.layers.{i}."
        sd[q + "projection.weight"], sd[q + "projection.bias"] = r[a + "0.weight"], r[a + "0.bias"]
        if i != 2:  # factor 1 has no resize
            sd[q + "resize.weight"], sd[q + "resize.bias"] = r[a + "1.weight"], r[a + "1.bias"]
        sd[f"convs.{i}.weight"] = r[f"scratch.layer{i + 1}_rn.weight"]
        # fusion_stage.layers[j] handles level 3 - j (coarse to fine) = scratch.refinenet{4 - j}
        f, g = f"fusion_stage.layers.{3 - i}.", f"scratch.refinenet{i + 1}."
        sd[f + "projection.weight"], sd[f + "projection.bias"] = r[g + "out_conv.weight"], r[g + "out_conv.bias"]
        for hf_unit, our_unit in (("residual_layer1", "resConfUnit1"), ("residual_layer2", "resConfUnit2")):
            for hf_c, our_c in (("convolution1", "conv1"), ("convolution2", "conv2")):
                sd[f + f"{hf_unit}.{hf_c}.weight"] = r[g + f"{our_unit}.{our_c}.weight"]
                sd[f + f"{hf_unit}.{hf_c}.bias"] = r[g + f"{our_unit}.{our_c}.bias"]
    return sd


@pytest.mark.parametrize("dim,layer_dims,feature_dim", [(64, (16, 32, 48, 64), 32), (48, (24, 24, 40, 56), 64)])
def test_dpt_feature_matches_hf_dpt_neck(dim, layer_dims, feature_dim):
    transformers = pytest.importorskip("transformers")
    warnings.filterwarnings("ignore")
    from transformers.models.dpt import modeling_dpt as M

    head = U.DPTFeature(patch_size=14, hooks=(0, 1, 2, 3), input_feature_dims=dim, layer_dims=layer_dims, feature_dim=feature_dim).eval()
    init_weights_(head, seed=3)
    cfg = transformers.DPTConfig(
        hidden_size=dim, neck_hidden_sizes=list(layer_dims), reassemble_factors=[4, 2, 1, 0.5], fusion_hidden_size=feature_dim,
        readout_type="ignore", is_hybrid=False, use_batch_norm_in_fusion_residual=False, use_bias_in_fusion_residual=True,
        neck_ignore_stages=[], num_hidden_layers=4, num_attention_heads=2, intermediate_size=4 * dim, image_size=112, patch_size=14,
    )
    neck = M.DPTNeck(cfg).eval()
    neck.load_state_dict(_neck_state_from_ref(head), strict=True)

    g = torch.Generator().manual_seed(11)
    b, gh = 2, 8  # even token grid: no crop / re-interpolation of a skip (module docstring)
    feats = [torch.randn(b, dim, gh, gh, generator=g) for _ in range(4)]
    with torch.no_grad():
        ours = head(U.PredictionHeadLayeredInput(list_features=feats, target_output_shape=(16 * gh, 16 * gh))).list_features[0]
        tokens = [torch.cat([torch.zeros(b, 1, dim), f.flatten(2).transpose(1, 2)], dim=1) for f in feats]  # dummy cls row in front
        theirs = neck(tokens, gh, gh)
    assert ours.shape == theirs[-1].shape == (b, feature_dim, 8 * gh, 8 * gh)
    scale = float(theirs[-1].abs().max())
    assert (ours - theirs[-1]).abs().max() <= 2e-5 * max(1.0, scale)
    # the coarser fused maps are not returned by the oracle class; re-derive the second-finest one from its sub-modules
    layers = [getattr(head.scratch, f"layer{i + 1}_rn")(head.act_postprocess[i](f)) for i, f in enumerate(feats)]
    with torch.no_grad():
        p4 = head.scratch.refinenet4(layers[3])
        p3 = head.scratch.refinenet3(p4, layers[2])
        p2 = head.scratch.refinenet2(p3, layers[1])
    for mine, hf in ((p4, theirs[0]), (p3, theirs[1]), (p2, theirs[2])):
        assert (mine - hf).abs().max() <= 2e-5 * max(1.0, float(hf.abs().max()))


def test_dpt_regression_processor_matches_hf_depth_head():
    transformers = pytest.importorskip("transformers")
    warnings.filterwarnings("ignore")
    from transformers.models.dpt import modeling_dpt as M

    feature_dim = 64
    proc = U.DPTRegressionProcessor(input_feature_dim=feature_dim, output_dim=1).eval()
    init_weights_(proc, seed=5)
    cfg = transformers.DPTConfig(hidden_size=32, neck_hidden_sizes=[8, 8, 8, 8], fusion_hidden_size=feature_dim, head_in_index=-1,
                                 add_projection=False, num_hidden_layers=4, num_attention_heads=2, intermediate_size=64)
    hf = M.DPTDepthEstimationHead(cfg).eval()
    r = proc.state_dict()
    sd = {"head.0.weight": r["conv1.weight"], "head.0.bias": r["conv1.bias"], "head.2.weight": r["conv2.0.weight"], "head.2.bias": r["conv2.0.bias"],
          "head.4.weight": r["conv2.2.weight"], "head.4.bias": r["conv2.2.bias"]}
    hf.load_state_dict(sd, strict=True)
    x = torch.randn(2, feature_dim, 12, 10, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        ours = proc(U.PredictionHeadLayeredInput(list_features=[x], target_output_shape=(24, 20))).decoded_channels
        theirs = hf.head[:5](x)  # everything before HF's final ReLU (depth >= 0), which UFM's heads do not have
    assert ours.shape == theirs.shape == (2, 1, 24, 20)
    assert (ours - theirs).abs().max() <= 2e-5 * max(1.0, float(theirs.abs().max()))


def test_info_sharing_block_matches_hf_dinov2_layer():
    """One block of the global-attention info-sharing transformer (pre-LN, joint self-attention over both views' tokens, MLP with
    exact GELU, no LayerScale: ``init_values=None``) against ``Dinov2Layer`` with LayerScale = 1."""
    transformers = pytest.importorskip("transformers")
    warnings.filterwarnings("ignore")
    from transformers.models.dinov2 import modeling_dinov2 as D

    dim, heads = 96, 3
    blk = U.Block(dim, heads, 4.0, True, None).eval()
    init_weights_(blk, seed=6)
    cfg = transformers.Dinov2Config(hidden_size=dim, num_hidden_layers=1, num_attention_heads=heads, mlp_ratio=4, layerscale_value=1.0,
                                    layer_norm_eps=1e-6, hidden_act="gelu", qkv_bias=True, use_swiglu_ffn=False)
    hf = D.Dinov2Layer(cfg).eval()
    r = blk.state_dict()
    sd = {}
    for n in ("norm1", "norm2"):
        sd[n + ".weight"], sd[n + ".bias"] = r[n + ".weight"], r[n + ".bias"]
    wq, wk, wv = r["attn.qkv.weight"].chunk(3, dim=0)
    bq, bk, bv = r["attn.qkv.bias"].chunk(3, dim=0)
    for nm, w, b in (("query", wq, bq), ("key", wk, bk), ("value", wv, bv)):
        sd[f"attention.attention.{nm}.weight"], sd[f"attention.attention.{nm}.bias"] = w, b
    sd["attention.output.dense.weight"], sd["attention.output.dense.bias"] = r["attn.proj.weight"], r["attn.proj.bias"]
    sd["layer_scale1.lambda1"], sd["layer_scale2.lambda1"] = torch.ones(dim), torch.ones(dim)
    for n in ("fc1", "fc2"):
        sd[f"mlp.{n}.weight"], sd[f"mlp.{n}.bias"] = r[f"mlp.{n}.weight"], r[f"mlp.{n}.bias"]
    hf.load_state_dict(sd, strict=True)
    x = torch.randn(2, 2 * 16, dim, generator=torch.Generator().manual_seed(8))  # one pair's two 4x4 grids, concatenated
    with torch.no_grad():
        ours = blk(x)
        theirs = hf(x)
        theirs = theirs[0] if isinstance(theirs, tuple) else theirs
    assert (ours - theirs).abs().max() <= 2e-5 * max(1.0, float(theirs.abs().max()))
