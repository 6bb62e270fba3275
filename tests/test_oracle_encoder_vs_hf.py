"""Cross-check of the oracle's restated DINOv2 encoder against ``transformers.Dinov2Model`` built
offline from a config object (no name, no fetch) -- SURVEY 8(c)(3).  CPU only."""

import warnings

import pytest
import torch

from oracle import uniception_ref as U


def _hf_state_from_ref(ref: U.DinoVisionTransformerRef):
    sd = {}
    r = ref.state_dict()
    d = ref.embed_dim
    sd["embeddings.cls_token"] = r["cls_token"]
    sd["embeddings.mask_token"] = torch.zeros(1, d)
    sd["embeddings.position_embeddings"] = r["pos_embed"]
    sd["embeddings.patch_embeddings.projection.weight"] = r["patch_embed.proj.weight"]
    sd["embeddings.patch_embeddings.projection.bias"] = r["patch_embed.proj.bias"]
    for i in range(len(ref.blocks)):
        p, q = f"blocks.{i}.", f"encoder.layer.{i}."
        for n in ("norm1", "norm2"):
            sd[q + n + ".weight"], sd[q + n + ".bias"] = r[p + n + ".weight"], r[p + n + ".bias"]
        wq, wk, wv = r[p + "attn.qkv.weight"].chunk(3, dim=0)
        bq, bk, bv = r[p + "attn.qkv.bias"].chunk(3, dim=0)
        for nm, w, b in (("query", wq, bq), ("key", wk, bk), ("value", wv, bv)):
            sd[q + f"attention.attention.{nm}.weight"], sd[q + f"attention.attention.{nm}.bias"] = w, b
        sd[q + "attention.output.dense.weight"], sd[q + "attention.output.dense.bias"] = r[p + "attn.proj.weight"], r[p + "attn.proj.bias"]
        sd[q + "layer_scale1.lambda1"], sd[q + "layer_scale2.lambda1"] = r[p + "ls1.gamma"], r[p + "ls2.gamma"]
        for n in ("fc1", "fc2"):
            sd[q + f"mlp.{n}.weight"], sd[q + f"mlp.{n}.bias"] = r[p + f"mlp.{n}.weight"], r[p + f"mlp.{n}.bias"]
    sd["layernorm.weight"], sd["layernorm.bias"] = r["norm.weight"], r["norm.bias"]
    return sd


@pytest.mark.parametrize("dim,depth,heads", [(128, 3, 2), (192, 2, 3)])
def test_encoder_matches_hf_dinov2(dim, depth, heads):
    transformers = pytest.importorskip("transformers")
    warnings.filterwarnings("ignore")
    from oracle.ufm_ref import init_weights_

    enc = U.DINOv2IntermediateFeatureReturner(
        embed_dim=dim, depth=depth, num_heads=heads, img_size=56, indices=[0, depth - 1]
    ).eval()
    init_weights_(enc, seed=2)
    cfg = transformers.Dinov2Config(
        hidden_size=dim, num_hidden_layers=depth, num_attention_heads=heads, image_size=56, patch_size=14,
        mlp_ratio=4, layerscale_value=1.0, layer_norm_eps=1e-6, hidden_act="gelu", qkv_bias=True, use_swiglu_ffn=False,
    )
    hf = transformers.Dinov2Model(cfg).eval()
    missing = hf.load_state_dict(_hf_state_from_ref(enc.model), strict=True)
    img = torch.randn(2, 3, 56, 56, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        ours = enc(U.ViTEncoderInput(image=img, data_norm_type="dinov2"))
        theirs = hf(pixel_values=img, output_hidden_states=True)
    # last block, final LayerNorm, cls dropped, BCHW at the native 4x4 grid (no pos-embed interpolation)
    ref_last = theirs.last_hidden_state[:, 1:].reshape(2, 4, 4, dim).permute(0, 3, 1, 2)
    assert (ours[-1].features - ref_last).abs().max() <= 2e-5
    # first returned intermediate = output of block 0, normed with the final LayerNorm (norm_intermediate=True)
    h0 = hf.layernorm(theirs.hidden_states[1])[:, 1:].reshape(2, 4, 4, dim).permute(0, 3, 1, 2)
    assert (ours[0].features - h0).abs().max() <= 2e-5
