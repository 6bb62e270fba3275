"""CPU-only checks of the host-side mirror of the reference interface (no kernels run)."""

import numpy as np
import pytest
import torch

import ufm_amd
from ufm_amd.base import closest_aspect_resolution, representation_region


def test_import_paths_match_reference():
    import uniflowmatch
    import uniflowmatch.models
    import uniflowmatch.models.base as b
    import uniflowmatch.models.ufm as u

    assert uniflowmatch.UniFlowMatchConfidence is ufm_amd.UniFlowMatchConfidence is u.UniFlowMatchConfidence
    assert b.UFMOutputInterface is ufm_amd.UFMOutputInterface
    for name in ("UniFlowMatch", "UniFlowMatchClassificationRefinement", "UFMFlowFieldOutput", "UFMMaskFieldOutput", "UniFlowMatchModelsBase"):
        assert hasattr(uniflowmatch.models, name)


def test_constructor_and_state_dict_namespace():
    m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.make_config(
        enc_dim=256, enc_heads=4, enc_depth=1, info_dim=128, info_heads=2, info_depth=2, layer_dims=(32, 32, 64, 64),
        feature_dim=64, native_img_size=56, resolution_wh=(56, 56)))
    keys = set(m.state_dict().keys())
    for k in ("encoder.model.cls_token", "encoder.model.pos_embed", "encoder.model.patch_embed.proj.weight", "encoder.model.blocks.0.attn.qkv.weight",
              "encoder.model.blocks.0.ls1.gamma", "info_sharing.proj_embed.weight", "info_sharing.self_attention_blocks.0.mlp.fc1.weight",
              "head1.0.0.scratch.refinenet1.resConfUnit1.conv1.weight", "head1.0.0.act_postprocess.0.1.weight", "head1.0.1.conv2.2.bias",
              "uncertainty_head.0.1.conv1.weight"):
        assert k in keys, k
    assert "encoder.model.mask_token" not in keys  # dropped by the reference's loader (ufm.py:208-210)
    assert m.inference_resolution == [(56, 56)]
    assert ufm_amd.UniFlowMatch(encoder_str="dinov2", encoder_kwargs=ufm_amd.ufm_tiny_config()["encoder_kwargs"],
                                info_sharing_kwargs=ufm_amd.ufm_tiny_config()["info_sharing_kwargs"],
                                feature_head_kwargs=ufm_amd.ufm_tiny_config()["feature_head_kwargs"],
                                adaptors_kwargs=ufm_amd.ufm_tiny_config()["adaptors_kwargs"]).inference_resolution == [(560, 420)]
    g = m.get_parameter_groups()
    assert set(g) == {"encoder", "info_sharing", "output_head", "uncertainty_head"}


def test_save_and_from_pretrained_local_dir(tmp_path):
    m = ufm_amd.UniFlowMatchClassificationRefinement(**ufm_amd.ufm_tiny_config(refine=True))
    ufm_amd.modules.init_weights_(m, 1)
    m.save_pretrained(str(tmp_path / "d"))
    m2 = ufm_amd.UniFlowMatchClassificationRefinement.from_pretrained(str(tmp_path / "d"))
    for (k1, v1), (k2, v2) in zip(sorted(m.state_dict().items()), sorted(m2.state_dict().items())):
        assert k1 == k2 and torch.equal(v1, v2)
    assert m2.refinement_range == 5 and m2.temperature == 4.0


def test_modify_state_dict_semantics():
    sd = {"model.a.feature_matching_proj.w": 1, "encoder.model.mask_token": 2, "encoder.model.cls_token": 3, "x_old_y": 4}
    out = ufm_amd.ufm.modify_state_dict(sd, {"feature_matching_proj": None, "encoder.model.mask_token": None, "_old_": "_new_"})
    assert out == {"encoder.model.cls_token": 3, "x_new_y": 4}


def test_resolution_selection_and_regions_match_reference_semantics():
    assert closest_aspect_resolution([(512, 200), (200, 512)], 145, 256, 135, 256) == (200, 512)
    assert closest_aspect_resolution([(560, 420)], 1080, 1080, 607, 1080) == (420, 560)
    with pytest.raises(ValueError):
        closest_aspect_resolution([], 1, 1, 1, 1)
    # float32 multiply + truncation of flow_resizing.py:332-345 (golden: selfdemo.npz regions)
    assert representation_region((200, 512), 145, 256) == [0, 200, 0, 512]
    for h in (607, 810, 1080, 580, 33, 47):
        r = representation_region((518, 518), h, h + 7)
        t = (torch.tensor([518 / h, 518 / h, 518 / (h + 7), 518 / (h + 7)]) * torch.tensor([0, h, 0, h + 7])).to(torch.int64).tolist()
        assert r == t


def test_error_behaviour_without_gpu():
    m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_tiny_config())
    z = torch.zeros(1, 56, 56, 3, dtype=torch.uint8)
    with pytest.raises(AssertionError, match="torch.Tensors"):
        m.predict_correspondences_batched(np.zeros((56, 56, 3)), z)
    with pytest.raises(AssertionError, match="dimensions 3 or 4"):
        m.predict_correspondences_batched(torch.zeros(56, 56), torch.zeros(56, 56))
    with pytest.raises(ValueError, match="3 channels"):
        m.predict_correspondences_batched(torch.zeros(1, 5, 8, 8, dtype=torch.uint8), torch.zeros(1, 5, 8, 8, dtype=torch.uint8))
    with pytest.raises(AssertionError, match="data_norm_type must be provided"):
        m.predict_correspondences_batched(torch.zeros(1, 3, 56, 56), torch.zeros(1, 3, 56, 56))
    with pytest.raises(ValueError, match="float32 or torch.uint8"):
        m.predict_correspondences_batched(z.int(), z.int())
    with pytest.raises(RuntimeError, match="GPU only"):  # the product has no CPU path
        m.predict_correspondences_batched(z, z)
    with pytest.raises(NotImplementedError):
        ufm_amd.UniFlowMatchConfidence(**{**ufm_amd.ufm_tiny_config(), "info_sharing_str": "cross_attention"})
    with pytest.raises(RuntimeError, match="parameter container"):
        m.encoder(None)
