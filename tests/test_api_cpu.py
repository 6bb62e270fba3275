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
    with pytest.raises(NotImplementedError, match="built are"):  # an info-sharing variant that does not exist
        ufm_amd.UniFlowMatchConfidence(**{**ufm_amd.ufm_tiny_config(), "info_sharing_str": "alternating_attention"})
    xa = ufm_amd.UniFlowMatchConfidence(**{**ufm_amd.ufm_tiny_config(), "info_sharing_str": "cross_attention",
                                           "info_sharing_kwargs": dict(input_embed_dim=128, depth=2, dim=128, num_heads=2)})
    assert len(xa.info_sharing.multi_view_branches) == 2 and "info_sharing.multi_view_branches.1.1.cross_attn.projq.weight" in xa.state_dict()
    with pytest.raises(RuntimeError, match="parameter container"):
        m.encoder(None)


def test_checkpoint_loading_is_never_silent(tmp_path):
    """PyTorchModelHubMixin's default load is strict=False with no check -- a parameter whose key differs would
    keep its random init and `ufm infer --weights` would write plausible but wrong images.  ufm_amd routes every loader
    through keymap.load_checked: missing parameters raise (ufm.py:216-217), extra keys must be on the allow list."""
    from safetensors.torch import load_file, save_file

    from ufm_amd import keymap

    m = ufm_amd.UniFlowMatchConfidence(**ufm_amd.ufm_tiny_config())
    ufm_amd.modules.init_weights_(m, 2)
    d = tmp_path / "good"
    m.save_pretrained(str(d))
    sd = load_file(str(d / "model.safetensors"))

    def variant(name, edit):
        v = tmp_path / name
        v.mkdir()
        (v / "config.json").write_text((d / "config.json").read_text())
        s2 = dict(sd)
        edit(s2)
        save_file(s2, str(v / "model.safetensors"))
        return str(v)

    # a renamed key = one missing + one unexpected: must raise and name both
    def rename(s2):
        s2["encoder.model.pos_embedding"] = s2.pop("encoder.model.pos_embed")
    with pytest.raises(RuntimeError, match="pos_embed"):
        ufm_amd.UniFlowMatchConfidence.from_pretrained(variant("renamed", rename))
    # a deleted key must raise
    with pytest.raises(RuntimeError, match="missing from the file"):
        ufm_amd.UniFlowMatchConfidence.from_pretrained(variant("deleted", lambda s2: s2.pop("head1.0.1.conv1.weight")))
    # allowed extras (the DINOv2 mask token the reference drops, ufm.py:209) load fine and leave every parameter exact
    ok = ufm_amd.UniFlowMatchConfidence.from_pretrained(variant("extra", lambda s2: s2.update({"encoder.model.mask_token": torch.zeros(1, 128)})))
    for (k1, v1), (k2, v2) in zip(sorted(m.state_dict().items()), sorted(ok.state_dict().items())):
        assert k1 == k2 and torch.equal(v1, v2)
    # keys of encoder variants this build does not implement are refused by name (a DINOv2-with-registers checkpoint would
    # otherwise load into the plain encoder and run to wrong outputs); the reference's Lightning path is strict too (ufm.py:203-211)
    with pytest.raises(RuntimeError, match="register"):
        ufm_amd.UniFlowMatchConfidence.from_pretrained(variant("registers", lambda s2: s2.update({"encoder.model.register_tokens": torch.zeros(1, 4, 128)})))
    with pytest.raises(RuntimeError, match="does not have"):
        ufm_amd.UniFlowMatchConfidence.from_pretrained(variant("rope", lambda s2: s2.update({"info_sharing.rope.cache": torch.zeros(4)})))
    # an unknown extra key is an error too (it usually means a renamed parameter)
    with pytest.raises(RuntimeError, match="does not have"):
        ufm_amd.UniFlowMatchConfidence.from_pretrained(variant("junk", lambda s2: s2.update({"encoder.model.something_new": torch.zeros(1)})))

    # Lightning-style training checkpoint through pretrained_checkpoint_path (ufm.py:198-211): "model." prefix, dropped keys
    ck = {"state_dict": {**{"model." + k: v for k, v in sd.items() if not k.startswith("uncertainty_head")},
                         "model.feature_matching_proj.weight": torch.zeros(2, 2), "model.encoder.model.mask_token": torch.zeros(1, 128),
                         "loss.weight": torch.zeros(1)}}
    path = str(tmp_path / "train.ckpt")
    torch.save(ck, path)
    cfg = ufm_amd.ufm_tiny_config()
    base = ufm_amd.UniFlowMatch(encoder_str=cfg["encoder_str"], encoder_kwargs=cfg["encoder_kwargs"], info_sharing_kwargs=cfg["info_sharing_kwargs"],
                                feature_head_kwargs=cfg["feature_head_kwargs"], adaptors_kwargs=cfg["adaptors_kwargs"], pretrained_checkpoint_path=path)
    assert torch.equal(base.state_dict()["encoder.model.pos_embed"], sd["encoder.model.pos_embed"])
    ck["state_dict"].pop("model.info_sharing.self_attention_blocks.0.attn.qkv.weight")
    torch.save(ck, path)
    with pytest.raises(RuntimeError, match="info_sharing.self_attention_blocks.0.attn.qkv.weight"):
        ufm_amd.UniFlowMatch(encoder_str=cfg["encoder_str"], encoder_kwargs=cfg["encoder_kwargs"], info_sharing_kwargs=cfg["info_sharing_kwargs"],
                             feature_head_kwargs=cfg["feature_head_kwargs"], adaptors_kwargs=cfg["adaptors_kwargs"], pretrained_checkpoint_path=path)
    assert keymap.normalise({"model.a.b": 1, "other": 2}, lightning=True) == {"a.b": 1}


def test_uncertainty_head_with_covariance_and_keypoint_confidence_constructs():
    """(f)3: the optional uncertainty-head branches of ufm.py:648-654 are part of the head's adaptor map."""
    cfg = ufm_amd.ufm_tiny_config()
    cfg["uncertainty_head_kwargs"]["dpt_processor"]["output_dim"] = 5
    cfg["uncertainty_adaptors_kwargs"] = dict(
        non_occluded_mask={"class": "MaskAdaptor", "kwargs": dict(name="non_occluded_mask")},
        flow_cov={"class": "Covariance2DAdaptor", "kwargs": dict(name="flow_cov")},
        keypoint_confidence={"class": "ConfidenceAdaptor", "kwargs": dict(name="keypoint_confidence", confidence_type="sigmoid", vmin=0.0, vmax=1.0)},
    )
    m = ufm_amd.UniFlowMatchConfidence(**cfg)
    ad = m.uncertainty_head[1].adaptors
    assert [a.required_channels for a in ad] == [1, 3, 1] and m.uncertainty_head[0][1].output_dim == 5
    with pytest.raises(ValueError, match="adaptors need"):
        ufm_amd.UniFlowMatchConfidence(**{**cfg, "uncertainty_head_kwargs": {**cfg["uncertainty_head_kwargs"], "dpt_processor": dict(input_feature_dim=64, output_dim=4)}})


def test_confidence_adaptor_rejects_unknown_types():
    """an unknown confidence_type used to fall through to identity."""
    from ufm_amd.modules import AdaptorSpec

    assert AdaptorSpec("ConfidenceAdaptor", name="c", confidence_type="sigmoid", vmin=0.0, vmax=1.0).confidence_type == 1
    with pytest.raises(ValueError, match="confidence_type"):
        AdaptorSpec("ConfidenceAdaptor", name="c", confidence_type="softplus")


def test_bench_line_keeps_every_judged_scalar_in_its_last_2000_characters():
    """bench.py prints ONE JSON line of ~15 kB and the driver's record keeps its last 2 000 characters: `order_line` puts the bulky
    per-kernel tables first and a compact `summary` of every judged scalar last (round-3 review: precise_mode.value was cut off)."""
    import json
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    shapes = {f"M21920 N{n} K{k} {e}": {"launches": 24, "ms_per_step": 4.8, "avg_launch_us": 201.5, "tflops": 912.4, "frac": 0.365}
              for n, k, e in ((1024, 4096, "f32 += (read-modify-write)"), (4096, 1024, "bf16 out GELU"), (3072, 1024, "bf16 out"), (1024, 1024, "f32 += (read-modify-write)"),
                              (768, 3072, "f32 += (read-modify-write)"), (3072, 768, "bf16 out GELU"), (2304, 768, "bf16 out"), (768, 768, "f32 += (read-modify-write)"))}
    fam = lambda ms, gf: {"launches": 60, "ms_per_step": ms, "avg_launch_us": 100.0, "bound": "mfma", "algorithmic_gflop": gf, "achieved": 800.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.33,  # noqa: E731
                          "traffic": 3.9e8, "mfma_busy_frac_pmc": 0.37}
    kernels = {"ufm_gemm_bf16": dict(fam(19.9, 17023.0), per_shape=shapes), "ufm_attention_bf16": fam(6.7, 5163.0),
               "ufm_conv2d_nhwc_bf16x3": dict(fam(11.7, 3487.0), per_shape={f"B8 {s}x{s} 256->256 k3": shapes[next(iter(shapes))] for s in range(19, 39)}),
               "ufm_dpt_tail_fused": fam(1.2, 316.0), "ufm_layernorm": {"launches": 77, "ms_per_step": 1.7, "bound": "hbm", "frac": 0.66, "achieved": 5300.0, "peak": 8000.0}}
    line = {"metric": "image-pairs/sec, UFM-Base 518x518", "value": 216.2, "unit": "pairs/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 37.0, "p50_latency_ms": 36.9,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic", "config": {"workload": "x" * 300, "numerics": "y" * 200},
            "roofline": {"kernel": "ufm_gemm_bf16", "bound": "mfma", "achieved": 850.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.34, "traffic": 3.9e8, "traffic_source": {"file": "z" * 80}},
            "attention": {"achieved": 750.0, "peak": 2500.0, "frac": 0.30, "ms_per_step": 6.9}, "kernels": kernels, "instrumented_step_ms": 41.9,
            "cpu_baseline": {"value": 0.144, "unit": "pairs/s", "cores": 16, "kind": "port", "sample": "s" * 250},
            "check_vs_oracle": {"numerics": "fast", "flow_max_abs": 0.035, "flow_range": 3.6, "covis_max_abs": 0.005},
            "latency_b1_ms": {"eager_p50": 9.4, "iters": 20, "batch": 1, "graph_replay_p50": 9.0, "graph_bitwise_equals_eager": True},
            "precise_mode": {"value": 103.7, "unit": "pairs/s", "ms_per_step": 77.1, "flow_max_abs": 1.2e-4, "covis_max_abs": 2e-5, "numerics": "n" * 200, "kernels": {k: v for k, v in kernels.items()}},
            "parity_mode": {"value": 33.7, "unit": "pairs/s", "ms_per_step": 237.0, "flow_max_abs": 6.1e-5, "numerics": "p" * 100}}
    # round 5 / 6 legs: the side configurations, the in-kernel clock and the per-family table of the two-stream dispatch
    side = lambda v: {"workload": "w" * 120, "value": v, "unit": "pairs/s", "ms_per_step": 38.5, "steps": 6, "numerics": "fast",  # noqa: E731
                      "mfma_families": {"gemm_bf16": {"ms": 19.1, "frac": 0.35}, "attention_bf16": {"ms": 6.1, "frac": 0.31}, "conv2d_nhwc_bf16x3": {"ms": 10.2, "frac": 0.4}}}
    line.update(config4=side(207.6), config5=side(37.5), default_res=side(256.9))
    line["roofline"].update(clock_ghz=1.92, frac_at_clock=0.46, clock_source="c" * 300)
    line["pipeline_kernels"] = {"families": {k.replace("ufm_", ""): {"launches": 100, "sum_ms": 30.0, "frac_while_sharing": 0.2016} for k in kernels if k != "ufm_layernorm"},
                                "sum_of_launch_ms": 72.5, "wall_ms_per_step": 35.07, "overlap_factor": 2.067, "dispatch": "d" * 120, "how": "h" * 150}
    text = json.dumps(bench.order_line(line, 8))
    assert len(text) > 6000  # the test is only meaningful on a line longer than the tail
    tail = text[-2000:]
    for needle in ('"summary"', '"value": 216.2', '"roofline_frac": 0.34', '"attention_frac": 0.3', '"precise": {"pairs_per_s": 103.7, "flow_max_abs": 0.00012}', '"parity": {"pairs_per_s": 33.7',
                   '"fast_flow_max_abs": 0.035', '"graph_replay_p50": 9.0', '"end_to_end_frac"', '"attention_share_frac"', '"gemm_shape_frac"', '"cpu_pairs_per_s": 0.144',
                   '"config4": {"pairs_per_s": 207.6', '"config5": {"pairs_per_s": 37.5', '"default_res": {"pairs_per_s": 256.9', '"pipeline_overlap_factor": 2.067', '"clock_ghz": 1.92'):
        assert needle in tail, needle
    back = json.loads(text)  # still one valid JSON object with the contract's keys
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in back, key
    assert abs(back["end_to_end"]["frac"] - (17023.0 + 5163.0 + 3487.0 + 316.0) * (216.2 / 8) / 1e3 / 2500.0) < 1e-9


def test_param_epoch_moves_on_every_kind_of_registration():
    """the engines' cached Parameter list is re-walked only when engine.PARAM_EPOCH moved.  torch's
    parameter-registration hook alone misses a swapped-in SUB-MODULE and a new buffer; the module- and buffer-registration hooks
    (round 5) catch them.  CPU-only: the hooks are plain torch."""
    import copy

    import torch
    from ufm_amd import engine

    seq = torch.nn.Sequential(torch.nn.Linear(2, 2), torch.nn.ReLU())
    e = engine.PARAM_EPOCH[0]
    seq[0] = copy.deepcopy(seq[0])  # Sequential.__setitem__ -> setattr of a Module: no Parameter is registered anywhere
    assert engine.PARAM_EPOCH[0] > e
    e = engine.PARAM_EPOCH[0]
    seq.extra_head = torch.nn.Linear(2, 2)
    assert engine.PARAM_EPOCH[0] > e
    e = engine.PARAM_EPOCH[0]
    seq.add_module("third", torch.nn.Identity())
    assert engine.PARAM_EPOCH[0] > e
    e = engine.PARAM_EPOCH[0]
    seq.register_buffer("table", torch.zeros(3))
    assert engine.PARAM_EPOCH[0] > e
    e = engine.PARAM_EPOCH[0]
    seq[0].bias = torch.nn.Parameter(torch.zeros(2))
    assert engine.PARAM_EPOCH[0] > e


def test_bench_gpus_n_without_a_launcher_starts_its_own_ranks_and_never_touches_the_gpu(monkeypatch, capsys):
    """`python bench.py --gpus N` (N > 1, WORLD_SIZE unset) used to die on an assertion.  The parent now
    starts the driver's own command (torch.distributed.run, one process per GPU, 127.0.0.1) as a CHILD process, relays its
    stdout, returns its exit code and makes no GPU call itself.  Partition: /root/reference/uniflowmatch/models/ufm.py:306-318
    only ever cats / chunks on dim 0, so ranks split the pair batch."""
    import os
    import subprocess
    import sys

    import torch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], 29400)
    assert cmd[:9] == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port", "29400"]
    assert cmd[9] == os.path.join(root, "bench.py") and cmd[10:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]

    seen = {}

    class FakeProc:
        def __init__(self, cmd, env=None, **kw):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = iter(['{"metric": "x", "n_gpus": 2}\n'])

        def wait(self):
            return 7

    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = bench.self_launch(2, ["--gpus", "2", "--steps", "1"])
    out = capsys.readouterr()
    assert rc == 7 and out.out == '{"metric": "x", "n_gpus": 2}\n'
    assert "--nproc-per-node=2" in seen["cmd"] and seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "parent touched the GPU: False" in out.err and not torch.cuda.is_initialized()


def test_bench_gpus_2_end_to_end_on_a_box_without_a_gpu_fails_in_the_ranks_not_in_the_parent():
    """The real thing here (no GPU): the parent launches two ranks, each rank refuses to run without an MI355X, the launcher's
    non-zero exit code comes back as the parent's, and the parent reports that it never initialised the GPU."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    envv = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=600, env=envv, cwd=root)
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the two ranks would run the real benchmark (covered by the -m gpu rehearsal test)")
    assert r.returncode != 0
    assert "bench.py needs an MI355X" in r.stderr and "launch with torch.distributed.run" not in r.stderr
    assert "parent touched the GPU: False" in r.stderr


def test_interleave_split_layout_and_block_eligibility():
    """Round 6: UFM_BF16X2_IL = [rows][C / 32][hi 32 | lo 32] (hip.interleave_split: a plain re-layout of the (2, rows, C) planes), and the rule that decides
    which transformer blocks run their Linears on it in numerics "precise" (engine._il_ok: N % 256 == 0, K % 32 == 0, K >= 64 for all four)."""
    import torch
    from ufm_amd import engine, hip

    planes = torch.arange(2 * 3 * 64, dtype=torch.float32).view(2, 3, 64).to(torch.bfloat16)
    il = hip.interleave_split(planes)
    assert il.shape == (3, 2, 2, 32) and il.is_contiguous()
    for row in range(3):
        for chunk in range(2):
            assert torch.equal(il[row, chunk, 0], planes[0, row, 32 * chunk : 32 * chunk + 32])  # hi half of the chunk
            assert torch.equal(il[row, chunk, 1], planes[1, row, 32 * chunk : 32 * chunk + 32])  # lo half right behind it
    L = torch.nn.Linear
    assert engine._il_ok(L(1024, 3072), L(1024, 1024), L(1024, 4096), L(4096, 1024))       # the encoder's block
    assert engine._il_ok(L(768, 2304), L(768, 768), L(768, 3072), L(3072, 768))            # the info-sharing block
    assert not engine._il_ok(L(128, 384), L(128, 128), L(128, 512), L(512, 128))           # the tiny test model: planar path
    assert not engine._il_ok(L(32, 256))                                                   # K below two K-tiles
