"""Pins the CPU oracle (oracle/ufm_ref.py) against vectors produced by the reference's OWN
glue code (tests/golden/make_goldens.py, run in the build container).  CPU only."""

import glob
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import ufm_ref as R

TOL = 1e-5  # glue goldens must match to <= 1e-5


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


class _FakeOracle(R.UFMRef):
    """Oracle pre/post around an analytic forward (same fields as the generator's FakeModel)."""

    def __init__(self, res, seed):
        torch.nn.Module.__init__(self)
        if isinstance(res[0], int):
            res = [res]
        self.inference_resolution = [tuple(r) for r in res]
        self.encoder = SimpleNamespace(data_norm_type="dinov2")
        self.seed = seed
        self.seen = None
        self.cov_in = None

    def forward(self, a, b):
        from tests.golden.make_goldens import analytic_fields

        self.seen = (a.clone(), b.clone())
        fl, mask = analytic_fields(a.shape[0], a.shape[2], a.shape[3], self.seed)
        out = R.Out(flow=R.FlowOut(fl), covisibility=R.MaskOut(mask, mask * 0))
        if self.cov_in is not None:  # the generator's FakeModel returned this flow_covariance (base.py:295-319 branch)
            out.flow.flow_covariance = self.cov_in
        return out


@pytest.mark.parametrize(
    "name", sorted(os.path.basename(p) for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "glue_prepost_*.npz")))
)
def test_prepost_glue(golden_dir, name):
    g = _load(golden_dir, name)
    res = [tuple(int(v) for v in r) for r in g["resolutions"]]
    norm = str(g["norm"]) or None
    m = _FakeOracle(res, int(g["fake_seed"]))
    if "cov_in" in g:
        m.cov_in = torch.from_numpy(g["cov_in"])
    out = m.predict_correspondences_batched(torch.from_numpy(g["src"]), torch.from_numpy(g["tgt"]), data_norm_type=norm)
    assert m.seen[0].shape == g["seen1"].shape
    assert np.abs(m.seen[0].numpy() - g["seen1"]).max() <= TOL
    assert np.abs(m.seen[1].numpy() - g["seen2"]).max() <= TOL
    assert out.flow.flow_output.shape == g["flow"].shape
    assert np.abs(out.flow.flow_output.numpy() - g["flow"]).max() <= TOL
    assert out.covisibility.mask.shape == g["mask"].shape
    assert np.abs(out.covisibility.mask.numpy() - g["mask"]).max() <= TOL
    assert out.covisibility.logits is None
    if "cov_out" in g:
        assert out.flow.flow_covariance.shape == g["cov_out"].shape
        assert np.abs(out.flow.flow_covariance.numpy() - g["cov_out"]).max() <= TOL * max(1.0, float(np.abs(g["cov_out"]).max()))


@pytest.mark.parametrize("name", ["unmap_full.npz", "unmap_crop.npz"])
def test_unmap(golden_dir, name):
    g = _load(golden_dir, name)
    rep0, src0, src1 = g["rep0"].tolist(), g["src0"].tolist(), g["src1"].tolist()
    shp = tuple(int(v) for v in g["shape0"])
    fo, fv = R.unmap_flow(torch.from_numpy(g["flow_in"]), rep0, src0, src1, shp)
    co, cv = R.unmap_channels(torch.from_numpy(g["chan_in"]), rep0, src0, shp)
    assert np.abs(fo.numpy() - g["flow_out"]).max() <= TOL
    assert (fv.numpy() == g["flow_valid"]).all()
    assert np.array_equal(co.numpy(), g["chan_out"])
    assert (cv.numpy() == g["chan_valid"]).all()


@pytest.mark.parametrize("name", ["refine_p5.npz", "refine_p3.npz"])
def test_refinement(golden_dir, name):
    g = _load(golden_dir, name)
    p = int(g["patch"])
    flow, feats = torch.from_numpy(g["flow"]), torch.from_numpy(g["feats"])
    res, logp = R.classification_refinement(flow, feats, p, float(g["temperature"]), torch.from_numpy(g["bias"]))
    assert np.abs(res.numpy() - g["residual"]).max() <= TOL
    assert np.abs(logp.numpy() - g["log_softmax"]).max() <= TOL
    neigh, offs = R.neighborhood_features(flow, feats[flow.shape[0] :], p)
    assert np.array_equal(offs.numpy(), g["offsets"])
    assert np.abs(neigh[:, ::7, ::7].numpy() - g["neigh_sample"]).max() <= TOL
    # channel 0 (x) varies along the LAST patch dim, channel 1 (y) along the first
    r = (p - 1) // 2
    assert offs[0, 0, 0, 0, -1, 0] == r and offs[0, 0, 0, -1, 0, 1] == r


@pytest.mark.parametrize("name,refine", [("wiring_confidence.npz", False), ("wiring_refine.npz", True)])
def test_forward_wiring(golden_dir, name, refine):
    """The reference's real forward ran on the restated blocks; the oracle's own wiring must agree."""
    g = _load(golden_dir, name)
    m = R.UFMRef(**R.ufm_tiny_config(refine=refine)).eval()
    R.init_weights_(m, seed=int(g["seed"]))
    wsum = float(sum(p.double().abs().sum() for p in m.parameters()))
    assert abs(wsum - float(g["weight_abs_sum"])) <= 1e-6 * wsum, "CPU RNG stream differs from the generator's"
    assert sorted(m.state_dict().keys()) == [str(k) for k in g["keys"]], "state-dict namespace differs from the reference class"
    out = m.predict_correspondences_batched(torch.from_numpy(g["src"]), torch.from_numpy(g["tgt"]))
    assert np.abs(out.flow.flow_output.numpy() - g["flow"]).max() <= 2e-4
    assert np.abs(out.covisibility.mask.numpy() - g["mask"]).max() <= TOL * 10
    out2 = m.predict_correspondences_batched(torch.from_numpy(g["src2"]), torch.from_numpy(g["tgt2"]))
    assert out2.flow.flow_output.shape == g["flow2"].shape
    assert np.abs(out2.flow.flow_output.numpy() - g["flow2"]).max() <= 2e-4
    assert np.abs(out2.covisibility.mask.numpy() - g["mask2"]).max() <= TOL * 10


def test_selfdemo_regions(golden_dir):
    g = _load(golden_dir, "selfdemo.npz")
    hw = R.select_resolution([(512, 200), (200, 512)], 145, 256, 135, 256)
    assert hw == (200, 512) == tuple(g["shape0"][1:3])
    _, _, s0, s1, p0, p1 = R.resize_pair(torch.zeros(1, 3, 145, 256), torch.zeros(1, 3, 135, 256), hw)
    assert s0 == g["src0"].tolist() and s1 == g["src1"].tolist()
    assert p0 == g["rep0"].tolist() and p1 == g["rep1"].tolist()


def test_known_answer_identity():
    """SURVEY 8(c) known-answer facts: 518->518 resize is identity, channel unmap exact, flow unmap <= 3.05e-5."""
    x = torch.rand(1, 3, 70, 70)
    r0, _, s0, s1, p0, _ = R.resize_pair(x, x, (70, 70))
    assert torch.equal(r0, x)
    fl = torch.randn(1, 2, 70, 70) * 10
    fo, _ = R.unmap_flow(fl, p0, s0, s1, (70, 70))
    assert (fo - fl).abs().max() <= 3.1e-5
    co, _ = R.unmap_channels(fl, p0, s0, (70, 70))
    assert torch.equal(co, fl)


@pytest.mark.parametrize("name", ["unet_small_odd.npz", "unet_small_even.npz", "unet_full_odd.npz"])
def test_unet_restatement_is_pinned(golden_dir, name):
    """R5: oracle UNetRef vs the reference's own UNet (models/unet_encoder.py loads standalone; the generator seeds the
    reference class with init_weights_(seed), the same call re-creates identical weights here): same state-dict keys,
    same outputs, odd sizes exercising the nearest fix-up of unet_encoder.py:66-67."""
    g = _load(golden_dir, name)
    net = R.UNetRef(in_channels=3, out_channels=16, features=[int(f) for f in g["features"]]).eval()
    R.init_weights_(net, seed=int(g["seed"]))
    assert sorted(net.state_dict().keys()) == [str(k) for k in g["keys"]]
    wsum = float(sum(p.double().abs().sum() for p in net.parameters()))
    assert abs(wsum - float(g["weight_abs_sum"])) <= 1e-6 * wsum
    with torch.no_grad():
        y = net(torch.from_numpy(g["x"]))
    assert y.shape == g["y"].shape
    assert np.abs(y.numpy() - g["y"]).max() <= 2e-5 * max(1.0, float(np.abs(g["y"]).max()))


@pytest.mark.parametrize("method", ["conv", "modulate"])
def test_refine_with_unet_wiring_is_pinned(golden_dir, method):
    """The reference's real UFM-Refine forward with use_unet_feature=True (ufm.py:816-825, :915-917, :967-983) running on
    the restated third-party blocks: which tensors are concatenated / modulated, conv1 -> ReLU -> conv2, chunk order."""
    g = _load(golden_dir, f"wiring_refine_unet_{method}.npz")
    m = R.UFMRef(**R.ufm_tiny_config(refine=True, use_unet_feature=True, feature_combine_method=method)).eval()
    R.init_weights_(m, seed=int(g["seed"]))
    assert sorted(m.state_dict().keys()) == [str(k) for k in g["keys"]]
    out = m.predict_correspondences_batched(torch.from_numpy(g["src"]), torch.from_numpy(g["tgt"]))
    assert np.abs(out.flow.flow_output.numpy() - g["flow"]).max() <= 2e-4
    assert np.abs(out.covisibility.mask.numpy() - g["mask"]).max() <= 1e-5
    s_n, t_n = R.to_bchw_normalised(torch.from_numpy(g["src"]), torch.from_numpy(g["tgt"]), "dinov2", None)
    low = m.forward(s_n, t_n)
    assert np.abs(low.classification_refinement.feature_map_0.numpy() - g["feature_map_0"]).max() <= 1e-4 * max(1.0, float(np.abs(g["feature_map_0"]).max()))
    assert np.abs(low.classification_refinement.residual.numpy() - g["residual"]).max() <= 2e-4


@pytest.mark.parametrize("name,refine", [("confidence", False), ("refine", True)])
def test_symmetrized_encoding_is_pinned(golden_dir, name, refine):
    """ufm.py:336-352 + interleave (ufm.py:69-82) through the reference's real forward with symmetrized=True on
    NON-symmetric inputs: pins which image's features land in which pair and view."""
    g = _load(golden_dir, f"wiring_symmetrized_{name}.npz")
    m = R.UFMRef(**R.ufm_tiny_config(refine=refine)).eval()
    R.init_weights_(m, seed=int(g["seed"]))
    out = m.forward(torch.from_numpy(g["img1"]), torch.from_numpy(g["img2"]), symmetrized=True)
    assert np.abs(out.flow.flow_output.numpy() - g["flow"]).max() <= 2e-4
    assert np.abs(out.covisibility.mask.numpy() - g["mask"]).max() <= 1e-5
    if refine:
        assert np.abs(out.classification_refinement.feature_map_1.numpy() - g["feature_map_1"]).max() <= 1e-4 * max(1.0, float(np.abs(g["feature_map_1"]).max()))
    plain = m.forward(torch.from_numpy(g["img1"]), torch.from_numpy(g["img2"]), symmetrized=False)
    assert (plain.flow.flow_output - out.flow.flow_output).abs().max() > 1e-2  # the golden does exercise the interleave
