"""`ufm infer` runner, CPU side: the restated flow colouring against the Middlebury wheel's anchor
colours (flow_vis is absent: parity unpinned), the argument surface of the reference CLI (cli.py:12-47), and its error
behaviour for unreadable images (cli.py:100-102: message + exit code 1)."""
import numpy as np
import pytest

from ufm_amd import cli, viz


def test_flow_to_color_wheel_anchors():
    # zero flow -> white; unit flows along the axes -> the wheel's anchor colours at full saturation
    f = np.zeros((2, 3, 2), np.float32)
    assert (viz.flow_to_color(f) == 255).all()
    f = np.zeros((1, 4, 2), np.float32)
    f[0, 0] = (1, 0)    # +x : red
    f[0, 1] = (0, 1)    # +y : the wheel a quarter turn on (yellow-green side)
    f[0, 2] = (-1, 0)   # -x : cyan side
    f[0, 3] = (0, -1)   # -y : blue-magenta side
    img = viz.flow_to_color(f).astype(int)
    # hand-computed from the wheel (55 entries: RY 15, YG 6, GC 4, CB 11, BM 13, MR 6), fk = (atan2(-v,-u)/pi + 1)/2 * 54:
    assert tuple(img[0, 0]) == (255, 0, 0)      # fk = 0            -> RY[0]
    assert tuple(img[0, 1]) == (255, 229, 0)    # fk = 13.5         -> (RY[13] + RY[14]) / 2 = (255, (221 + 238) / 2, 0)
    assert tuple(img[0, 2]) == (0, 209, 255)    # fk = 27           -> CB[2] = (0, 255 - floor(255 * 2 / 11), 255)
    assert tuple(img[0, 3]) == (88, 0, 255)     # fk = 40.5         -> (BM[4] + BM[5]) / 2 = ((78 + 98) / 2, 0, 255)
    # half magnitude -> half-way to white, same hue
    g = np.zeros((1, 2, 2), np.float32)
    g[0, 0], g[0, 1] = (1, 0), (0.5, 0)
    img = viz.flow_to_color(g).astype(int)
    assert tuple(img[0, 0]) == (255, 0, 0) and img[0, 1, 0] == 255 and 120 <= img[0, 1, 1] <= 135 and img[0, 1, 1] == img[0, 1, 2]


def test_flow_to_color_is_scale_invariant_and_uint8():
    rng = np.random.default_rng(0)
    f = rng.normal(size=(17, 23, 2)).astype(np.float32)
    a, b = viz.flow_to_color(f), viz.flow_to_color(f * 37.0)
    assert a.dtype == np.uint8 and a.shape == (17, 23, 3)
    assert np.abs(a.astype(int) - b.astype(int)).max() <= 1  # normalised by the largest magnitude


def test_cli_argument_surface_matches_the_reference():
    p = cli.build_parser()
    a = p.parse_args(["infer", "s.png", "t.png"])
    assert (a.command, a.source, a.target, a.output, a.model) == ("infer", "s.png", "t.png", None, "base")
    a = p.parse_args(["infer", "s.png", "t.png", "-o", "out", "--model", "refine"])
    assert (a.output, a.model) == ("out", "refine")
    assert p.parse_args(["test"]).command == "test"
    d = p.parse_args(["demo", "--port", "1234", "--share"])
    assert (d.command, d.port, d.share, d.model) == ("demo", 1234, True, "base")
    with pytest.raises(SystemExit):
        p.parse_args(["infer", "only_one.png"])


def test_cli_unreadable_image_exits_1(capsys, tmp_path):
    with pytest.raises(SystemExit) as e:
        cli.main(["infer", str(tmp_path / "missing_a.png"), str(tmp_path / "missing_b.png")])
    assert e.value.code == 1
    assert "Error: Could not load one or both images" in capsys.readouterr().out


def test_warp_has_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        viz.warp_image_with_flow(np.zeros((4, 4, 3), np.uint8), None, np.zeros((4, 4, 3), np.uint8), np.zeros((4, 4, 2), np.float32))


def test_example_inference_figure(tmp_path):
    """example_inference.py:45-90: the 2 x 3 figure (host side; the warped panel is passed in because the warp is a GPU kernel)."""
    from PIL import Image

    from ufm_amd import example_inference as ex

    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (30, 40, 3), dtype=np.uint8)
    tgt = rng.integers(0, 256, (30, 40, 3), dtype=np.uint8)
    flow = rng.normal(size=(2, 30, 40)).astype(np.float32)
    cov = rng.random((30, 40)).astype(np.float32)
    out = tmp_path / "fig.png"
    fig = ex.visualize_results(src, tgt, flow, cov, str(out), warped_image=np.clip(tgt / 255.0, 0, 1))
    assert len(fig.axes) == 7  # 6 panels + the colour bar
    titles = [a.get_title() for a in fig.axes[:6]]
    assert titles == ["Source Image", "Target Image", "Warped Source Image", "Flow Visualization (Valid at Covisible Pixels)",
                      "Covisibility Mask (>0.5)", "Covisibility Confidence"]
    im = Image.open(out)
    assert im.size[0] > 800 and im.size[1] > 500
    with pytest.raises(ValueError, match="Could not load image"):
        ex.load_image(tmp_path / "missing.png")


def test_save_png_rounds_like_cv2(tmp_path):
    """cv2.imwrite saturate-casts float images (round to nearest, clamp); ADVICE r1: truncation biased every level down."""
    from PIL import Image

    a = np.array([[[0.4, 0.5, 0.6], [254.5, 255.7, -3.0]]], dtype=np.float32)
    viz.save_png(tmp_path / "x.png", a)
    got = np.array(Image.open(tmp_path / "x.png"))
    assert got.tolist() == [[[0, 0, 1], [254, 255, 0]]]  # rint: halves to even
