#!/usr/bin/env python3
"""Golden-vector generator.  RUN ONLY IN THE BUILD CONTAINER (needs /root/reference).

Imports the reference's own Python glue (``/root/reference/uniflowmatch``) with the
absent third-party modules (``uniception``, ``cv2``) registered as stub modules whose
attributes are the oracle's restatement classes, drives the reference's functions
on seeded inputs and stores inputs + outputs as small ``.npz`` fixtures next to this
script.  The fixtures are data only; no reference source is copied.

Families:
  glue_prepost_*.npz   base.py:137-334 + flow_resizing.py through a fake ``forward``
  unmap_*.npz          flow_resizing.py:749-877, :955-1010 called directly
  refine_*.npz         ufm.py:1012-1178 classification refinement
  wiring_*.npz         the reference's real ``forward`` / ``predict_correspondences_batched``
                       (ufm.py:562-662, :843-1009) running on top of the oracle's
                       restated blocks, weights from ``init_weights_(seed)``
  selfdemo.npz         flow_resizing.py:1013-1091 self-demo regions
"""

import os
import sys
import types
import warnings
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
REF = "/root/reference"

from oracle import ufm_ref as R  # noqa: E402
from oracle import uniception_ref as U  # noqa: E402


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    _mod("cv2")
    _mod("uniception")
    _mod("uniception.models")
    _mod(
        "uniception.models.encoders",
        ViTEncoderInput=U.ViTEncoderInput,
        feature_returner_encoder_factory=U.feature_returner_encoder_factory,
    )
    _mod("uniception.models.encoders.image_normalizations", IMAGE_NORMALIZATION_DICT=U.IMAGE_NORMALIZATION_DICT)
    _mod(
        "uniception.models.info_sharing",
        INFO_SHARING_CLASSES=U.INFO_SHARING_CLASSES,
        MultiViewTransformerInput=U.MultiViewTransformerInput,
    )
    _mod("uniception.models.prediction_heads")
    _mod(
        "uniception.models.prediction_heads.adaptors",
        ConfidenceAdaptor=U.ConfidenceAdaptor,
        Covariance2DAdaptor=U.Covariance2DAdaptor,
        FlowAdaptor=U.FlowAdaptor,
        FlowWithConfidenceAdaptor=U.FlowWithConfidenceAdaptor,
        MaskAdaptor=U.MaskAdaptor,
    )
    _mod(
        "uniception.models.prediction_heads.base",
        AdaptorMap=U.AdaptorMap,
        PredictionHeadInput=U.PredictionHeadInput,
        PredictionHeadLayeredInput=U.PredictionHeadLayeredInput,
    )
    _mod("uniception.models.prediction_heads.dpt", DPTFeature=U.DPTFeature, DPTRegressionProcessor=U.DPTRegressionProcessor)
    _mod("uniception.models.prediction_heads.mlp_feature", MLPFeature=U.MLPFeature)
    _mod("uniception.models.prediction_heads.moge_conv", MoGeConvFeature=U.MoGeConvFeature)
    sys.path.insert(0, REF)


def np_(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def analytic_fields(b, h, w, seed):
    """Smooth, non-trivial flow + mask at network resolution for the fake forward."""
    g = torch.Generator().manual_seed(seed)
    ys = torch.linspace(-1, 1, h).view(1, 1, h, 1)
    xs = torch.linspace(-1, 1, w).view(1, 1, 1, w)
    a = torch.rand(b, 2, 1, 1, generator=g) * 20 - 10
    fl = a * torch.cat([torch.sin(3 * xs + ys), torch.cos(2 * ys - xs)], dim=1) + torch.randn(b, 2, h, w, generator=g)
    mask = torch.sigmoid(4 * torch.sin(5 * xs * ys + a[:, :1]))
    return fl.float(), mask.float()


def main():
    warnings.filterwarnings("ignore")
    install_stubs()
    import uniflowmatch.models.base as ref_base
    import uniflowmatch.models.ufm as ref_ufm
    import uniflowmatch.utils.flow_resizing as ref_fr

    # ------------------------------------------------------------------ glue
    class FakeModel(ref_base.UniFlowMatchModelsBase):
        def __init__(self, res, seed, with_cov=False):
            super().__init__(inference_resolution=res)
            self.encoder = SimpleNamespace(data_norm_type="dinov2")
            self.seed = seed
            self.seen = None
            self.with_cov = with_cov

        def forward(self, view1, view2):
            self.seen = (view1["img"].clone(), view2["img"].clone())
            b, _, h, w = view1["img"].shape
            fl, mask = analytic_fields(b, h, w, self.seed)
            out = ref_base.UFMOutputInterface()
            out.flow = ref_base.UFMFlowFieldOutput(flow_output=fl)
            if self.with_cov:  # base.py:295-319: the covariance is un-mapped and rescaled by [wr^2, hr^2, wr*hr]
                gc = torch.Generator().manual_seed(self.seed + 1000)
                out.flow.flow_covariance = torch.rand(b, 3, h, w, generator=gc) + 0.1 * fl[:, :1].abs()
                self.cov_in = out.flow.flow_covariance.clone()
            out.covisibility = ref_base.UFMMaskFieldOutput(mask=mask, logits=mask * 0)
            return out

    cases = [
        # name, resolutions (W,H), src shape, tgt shape, layout, dtype, batched
        ("ident_u8_bhwc", (56, 56), (56, 56), (56, 56), "bhwc", "u8", True),
        ("ident_u8_hwc", (56, 56), (56, 56), (56, 56), "bhwc", "u8", False),
        ("down_u8_bhwc", (56, 42), (75, 100), (60, 90), "bhwc", "u8", True),
        ("up_f32_bchw", (70, 56), (30, 40), (33, 47), "bchw", "f32", True),
        ("multi_res", [(56, 42), (42, 56), (56, 56)], (120, 70), (110, 80), "bhwc", "u8", True),
        ("renorm_f32", (56, 56), (64, 48), (64, 48), "bchw", "f32_dust3r", True),
        ("cov_down_u8", (56, 42), (75, 100), (60, 90), "bhwc", "u8", True),   # flow_covariance branch, base.py:295-319
        ("cov_ident_u8", (56, 56), (56, 56), (56, 56), "bhwc", "u8", True),
    ]
    for i, (name, res, s_hw, t_hw, layout, dt, batched) in enumerate(cases):
        g = torch.Generator().manual_seed(100 + i)
        b = 2 if batched else 1
        if dt == "u8":
            src = torch.randint(0, 256, (b, s_hw[0], s_hw[1], 3), dtype=torch.uint8, generator=g)
            tgt = torch.randint(0, 256, (b, t_hw[0], t_hw[1], 3), dtype=torch.uint8, generator=g)
            norm = None
        else:
            src = torch.randn(b, s_hw[0], s_hw[1], 3, generator=g)
            tgt = torch.randn(b, t_hw[0], t_hw[1], 3, generator=g)
            norm = "dust3r" if dt.endswith("dust3r") else "dinov2"
        if layout == "bchw":
            src, tgt = src.permute(0, 3, 1, 2).contiguous(), tgt.permute(0, 3, 1, 2).contiguous()
        if not batched:
            src, tgt = src[0], tgt[0]
        m = FakeModel(res, seed=7 + i, with_cov=name.startswith("cov_"))
        out = m.predict_correspondences_batched(src, tgt, data_norm_type=norm)
        assert out.covisibility.logits is None
        extra = {}
        if m.with_cov:
            extra = dict(cov_in=np_(m.cov_in), cov_out=np_(out.flow.flow_covariance))
        save(
            f"glue_prepost_{name}.npz",
            **extra,
            src=np_(src),
            tgt=np_(tgt),
            resolutions=np.array(res if isinstance(res, list) else [res]),
            norm=np.array(norm if norm else ""),
            fake_seed=np.array(7 + i),
            seen1=np_(m.seen[0]),
            seen2=np_(m.seen[1]),
            flow=np_(out.flow.flow_output),
            mask=np_(out.covisibility.mask),
        )

    # ---------------------------------------------------------------- unmap
    g = torch.Generator().manual_seed(5)
    for name, (h, w), rep0, src0, src1, shp0 in [
        ("full", (42, 56), [0, 42, 0, 56], [0, 75, 0, 100], [0, 60, 0, 90], (75, 100)),
        ("crop", (40, 64), [4, 36, 8, 60], [10, 70, 5, 95], [0, 50, 20, 80], (80, 100)),
    ]:
        fl = torch.randn(2, 2, h, w, generator=g) * 5
        ch = torch.randn(2, 3, h, w, generator=g)
        t = lambda v: torch.tensor(v)  # noqa: E731
        fo, fv = ref_fr.unmap_predicted_flow(fl, t(rep0), t(rep0), t(src0), t(src1), shp0, shp0)
        co, cv = ref_fr.unmap_predicted_channels(ch, t(rep0), t(src0), shp0)
        save(
            f"unmap_{name}.npz",
            flow_in=np_(fl), chan_in=np_(ch), rep0=np.array(rep0), src0=np.array(src0), src1=np.array(src1),
            shape0=np.array(shp0), flow_out=np_(fo), flow_valid=np_(fv), chan_out=np_(co), chan_valid=np_(cv),
        )

    # --------------------------------------------------------------- refine
    cls = ref_ufm.UniFlowMatchClassificationRefinement
    for name, (b, c, h, w), p, temp in [("p5", (2, 16, 28, 42), 5, 4.0), ("p3", (1, 8, 14, 14), 3, 2.0)]:
        g = torch.Generator().manual_seed(11)
        ns = SimpleNamespace(refinement_range=p, temperature=temp, classification_bias=torch.randn(p * p, generator=g) * 0.3)
        ns.obtain_neighborhood_features = types.MethodType(cls.obtain_neighborhood_features, ns)
        ns.compute_refinement_attention = types.MethodType(cls.compute_refinement_attention, ns)
        flow = torch.randn(b, 2, h, w, generator=g) * 4  # some targets land outside -> zero padding exercised
        feats = torch.randn(2 * b, c, h, w, generator=g)
        res, logp = cls.classification_refinement(ns, flow, feats)
        neigh, offs = cls.obtain_neighborhood_features(ns, flow, feats[b:], p)
        save(
            f"refine_{name}.npz",
            flow=np_(flow), feats=np_(feats), bias=np_(ns.classification_bias), temperature=np.array(temp), patch=np.array(p),
            residual=np_(res), log_softmax=np_(logp), offsets=np_(offs), neigh_sample=np_(neigh[:, ::7, ::7]),
        )

    # --------------------------------------------------------------- wiring
    for name, refine in [("confidence", False), ("refine", True)]:
        cfg = R.ufm_tiny_config(refine=refine)
        ref_cls = ref_ufm.UniFlowMatchClassificationRefinement if refine else ref_ufm.UniFlowMatchConfidence
        kw = dict(cfg)
        if refine:
            kw["classification_head_type"] = "patch_mlp"
        model = ref_cls(**kw).eval()
        R.init_weights_(model, seed=3)
        g = torch.Generator().manual_seed(21)
        src = torch.randint(0, 256, (2, 56, 56, 3), dtype=torch.uint8, generator=g)
        tgt = torch.randint(0, 256, (2, 56, 56, 3), dtype=torch.uint8, generator=g)
        with torch.no_grad():
            out = model.predict_correspondences_batched(src, tgt)
        wsum = float(sum(p.double().abs().sum() for p in model.parameters()))
        arrays = dict(
            src=np_(src), tgt=np_(tgt), seed=np.array(3), weight_abs_sum=np.array(wsum),
            flow=np_(out.flow.flow_output), mask=np_(out.covisibility.mask),
            keys=np.array(sorted(model.state_dict().keys())),
        )
        # non-identity resolution through the same model
        src2 = torch.randint(0, 256, (1, 90, 70, 3), dtype=torch.uint8, generator=g)
        tgt2 = torch.randint(0, 256, (1, 64, 80, 3), dtype=torch.uint8, generator=g)
        with torch.no_grad():
            out2 = model.predict_correspondences_batched(src2, tgt2)
        arrays.update(src2=np_(src2), tgt2=np_(tgt2), flow2=np_(out2.flow.flow_output), mask2=np_(out2.covisibility.mask))
        save(f"wiring_{name}.npz", **arrays)

    # -------------------------------------------------- symmetrized encoding
    # ufm.py:336-352: with symmetrized=True only img1[::2] / img2[::2] are encoded and the features interleaved.  Inputs
    # here are deliberately NOT symmetric, so the golden pins which image's features end up in which pair and view.
    for name, refine in [("confidence", False), ("refine", True)]:
        cfg = R.ufm_tiny_config(refine=refine)
        ref_cls = ref_ufm.UniFlowMatchClassificationRefinement if refine else ref_ufm.UniFlowMatchConfidence
        kw = dict(cfg)
        if refine:
            kw["classification_head_type"] = "patch_mlp"
        model = ref_cls(**kw).eval()
        R.init_weights_(model, seed=3)
        g = torch.Generator().manual_seed(29)
        a, b = torch.randn(4, 3, 56, 56, generator=g), torch.randn(4, 3, 56, 56, generator=g)
        with torch.no_grad():
            out = model(dict(img=a, symmetrized=True, data_norm_type="dinov2"), dict(img=b, symmetrized=True, data_norm_type="dinov2"))
        arrays = dict(img1=np_(a), img2=np_(b), seed=np.array(3), flow=np_(out.flow.flow_output), mask=np_(out.covisibility.mask))
        if refine:
            arrays["feature_map_1"] = np_(out.classification_refinement.feature_map_1)
        save(f"wiring_symmetrized_{name}.npz", **arrays)

    # ----------------------------------------------------------------- UNet
    # models/unet_encoder.py loads standalone: the reference's OWN class, seeded weights (R.init_weights_), odd sizes that
    # exercise the nearest fix-up at :66-67 (54 -> 27 -> 13 -> 6; 13 vs 2*6, 27 vs 2*13)
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_unet_encoder", os.path.join(REF, "uniflowmatch", "models", "unet_encoder.py"))
    ref_unet = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_unet)
    for name, feats, shape in [("small_odd", [8, 16, 32, 64], (1, 3, 38, 54)), ("small_even", [8, 16, 32, 64], (2, 3, 64, 48)),
                               ("full_odd", [64, 128, 256, 512], (1, 3, 42, 70))]:
        net = ref_unet.UNet(in_channels=3, out_channels=16, features=feats).eval()
        R.init_weights_(net, seed=5)
        gx = torch.Generator().manual_seed(31)
        x = torch.randn(shape, generator=gx)
        with torch.no_grad():
            y = net(x)
        save(f"unet_{name}.npz", x=np_(x), y=np_(y), seed=np.array(5), features=np.array(feats),
             keys=np.array(sorted(net.state_dict().keys())), weight_abs_sum=np.array(float(sum(p.double().abs().sum() for p in net.parameters()))))

    # the reference's real UFM-Refine forward with use_unet_feature=True (ufm.py:816-825, :915-917, :967-983)
    for method in ("conv", "modulate"):
        cfg = R.ufm_tiny_config(refine=True, use_unet_feature=True, feature_combine_method=method)
        kw = dict(cfg)
        kw["classification_head_type"] = "patch_mlp"
        model = ref_ufm.UniFlowMatchClassificationRefinement(**kw).eval()
        R.init_weights_(model, seed=3)
        g = torch.Generator().manual_seed(23)
        src = torch.randint(0, 256, (2, 56, 56, 3), dtype=torch.uint8, generator=g)
        tgt = torch.randint(0, 256, (2, 56, 56, 3), dtype=torch.uint8, generator=g)
        with torch.no_grad():
            out = model.predict_correspondences_batched(src, tgt)
            s_n, t_n = R.to_bchw_normalised(src, tgt, "dinov2", None)
            low = model(dict(img=s_n, symmetrized=False, data_norm_type="dinov2"), dict(img=t_n, symmetrized=False, data_norm_type="dinov2"))
        wsum = float(sum(p.double().abs().sum() for p in model.parameters()))
        save(f"wiring_refine_unet_{method}.npz", src=np_(src), tgt=np_(tgt), seed=np.array(3), weight_abs_sum=np.array(wsum),
             flow=np_(out.flow.flow_output), mask=np_(out.covisibility.mask), feature_map_0=np_(low.classification_refinement.feature_map_0),
             residual=np_(low.classification_refinement.residual), keys=np.array(sorted(model.state_dict().keys())))

    # -------------------------------------------------------------- selfdemo
    sel = ref_fr.AutomaticShapeSelection(
        ref_fr.ResizeToFixedManipulation((200, 512)), ref_fr.ResizeToFixedManipulation((512, 200)), strategy="closest_aspect"
    )
    i0 = torch.zeros(1, 145, 256, 3)
    i1 = torch.zeros(1, 135, 256, 3)
    r0, r1, s0, s1, p0, p1 = sel(i0, i1)
    save("selfdemo.npz", shape0=np.array(r0.shape), shape1=np.array(r1.shape), src0=np_(s0), src1=np_(s1), rep0=np_(p0), rep1=np_(p1))


if __name__ == "__main__":
    main()
