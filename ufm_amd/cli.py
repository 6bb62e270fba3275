#!/usr/bin/env python3
"""``ufm`` command line for the MI355X implementation.

Same surface as the reference's ``uniflowmatch/cli.py`` (sub-commands ``demo`` / ``infer`` / ``test``; ``infer SOURCE
TARGET [-o DIR] [--model base|refine]``) and the same three artefacts for ``infer`` (cli.py:124-150):
``flow_visualization.png``, ``covisibility_mask.png``, ``warped_source.png``; an unreadable image or a failing run
prints the reference's message and exits with status 1.

What differs, all forced by this environment: PIL reads and writes the images (OpenCV is not installed); ``--weights``
names a local ``save_pretrained`` directory or a ``.ckpt`` (no network for the Hub ids the reference hard-codes --
without it the Hub id is tried and fails the way the reference does offline); ``--random-init`` runs the architecture
on seeded random weights (a plumbing check); ``--numerics`` picks the engine's arithmetic.  ``demo`` (Gradio) is out
of scope.  ``test`` checks this stack: imports, the HIP library's ABI, the GPU.
"""

from __future__ import annotations

import argparse
import sys
from pathlib import Path

ARTEFACTS = ("flow_visualization.png", "covisibility_mask.png", "warped_source.png")
HUB_IDS = {"base": "infinity1096/UFM-Base", "refine": "infinity1096/UFM-Refine"}  # cli.py:109-111
MODEL_FLAG = dict(choices=sorted(HUB_IDS), default="base", help="Model variant to use (default: base)")


def build_parser() -> argparse.ArgumentParser:
    root = argparse.ArgumentParser(prog="ufm", description="UFM: Unified Dense Correspondence with Flow")
    commands = root.add_subparsers(dest="command", help="Available commands")

    gradio = commands.add_parser("demo", help="Launch interactive Gradio demo (not built in ufm_amd)")
    gradio.add_argument("--port", type=int, default=7860)
    gradio.add_argument("--share", action="store_true")
    gradio.add_argument("--model", **MODEL_FLAG)

    pair = commands.add_parser("infer", help="Run inference on image pairs")
    for positional in ("source", "target"):
        pair.add_argument(positional, help=f"{positional.capitalize()} image path")
    pair.add_argument("--output", "-o", help="Output directory (default: current directory)")
    pair.add_argument("--model", **MODEL_FLAG)
    pair.add_argument("--weights", help="local save_pretrained directory or .ckpt file (default: the reference's Hub id)")
    pair.add_argument("--random-init", action="store_true", help="seeded random weights instead of a checkpoint")
    pair.add_argument("--numerics", choices=("fast", "precise", "parity"), default="fast")

    commands.add_parser("test", help="Test installation")
    return root


def load_model(args):
    """Construct the requested model on the GPU (monkey-patched by the tests to use a tiny configuration)."""
    import ufm_amd
    from ufm_amd.modules import init_weights_

    refine = args.model == "refine"
    cls = ufm_amd.UniFlowMatchClassificationRefinement if refine else ufm_amd.UniFlowMatchConfidence
    if args.random_init:
        model = cls(**(ufm_amd.ufm_refine_config() if refine else ufm_amd.ufm_base_config()))
        init_weights_(model, seed=0)
    elif args.weights and str(args.weights).endswith(".ckpt"):
        model = cls.from_pretrained_ckpt(args.weights)
    else:
        model = cls.from_pretrained(args.weights or HUB_IDS[args.model])
    return model.eval().to("cuda").set_numerics(args.numerics)


def _fail(message: str) -> None:
    print(message)
    sys.exit(1)


def _infer(args) -> Path:
    import numpy as np
    import torch

    from ufm_amd import viz

    try:
        images = [viz.load_rgb(path) for path in (args.source, args.target)]
    except OSError:  # FileNotFoundError, PIL.UnidentifiedImageError
        _fail("Error: Could not load one or both images")
    source, target = images
    model = load_model(args)
    print("Running inference...")
    with torch.no_grad():
        prediction = model.predict_correspondences_batched(
            source_image=torch.from_numpy(source).to("cuda"), target_image=torch.from_numpy(target).to("cuda")
        )
    flow_hw2 = prediction.flow.flow_output[0].permute(1, 2, 0).cpu().numpy()
    covis = prediction.covisibility.mask[0].cpu().numpy()

    out_dir = Path(args.output) if args.output else Path.cwd()
    out_dir.mkdir(exist_ok=True)
    # cli.py:139-146: the target warped into the source frame, white where the pair does not overlap
    warped = viz.warp_image_with_flow(source, None, target, flow_hw2)
    blended = covis[..., None] * warped + (1.0 - covis[..., None]) * 255.0
    images_out = (viz.flow_to_color(flow_hw2), (covis * 255).astype(np.uint8), blended)
    for name, image in zip(ARTEFACTS, images_out):
        viz.save_png(out_dir / name, image)
    return out_dir


def run_inference(args) -> None:
    """cli.py:85-156."""
    try:
        out_dir = _infer(args)
    except ImportError as exc:
        _fail(f"Error importing dependencies: {exc}\nPlease ensure all dependencies are installed")
    except SystemExit:
        raise
    except Exception as exc:  # the reference reports the error and exits 1 (cli.py:154-156)
        _fail(f"Error during inference: {exc}")
    print(f"Results saved to: {out_dir}")
    for name in ARTEFACTS:
        print(f"- {name}")


def test_installation() -> None:
    """The reference's ``ufm test`` (cli.py:159-212) for this stack."""
    print("Testing UFM installation...")
    try:
        import numpy
        import PIL
        import torch

        for label, version in (("PyTorch", torch.__version__), ("NumPy", numpy.__version__), ("Pillow", PIL.__version__)):
            print(f"✓ {label} {version}")
        from ufm_amd import hip
        from ufm_amd.ufm import UniFlowMatchConfidence  # noqa: F401

        print("✓ UFM model imports")
        lib = hip.lib()
        print(f"✓ libufm_hip.so ABI {lib.ufm_abi_version()} built for {lib.ufm_built_arch().decode()}")
        if torch.cuda.is_available():
            print(f"✓ ROCm {torch.version.hip} available\n  GPU: {torch.cuda.get_device_name(0)}")
        else:
            print("⚠ no GPU visible: ufm_amd has no CPU path, inference will raise")
        print("\n✅ Installation test completed successfully!")
    except ImportError as exc:
        _fail(f"❌ Import error: {exc}\nPlease check your installation")
    except Exception as exc:
        _fail(f"❌ Unexpected error: {exc}")


def main(argv=None) -> None:
    parser = build_parser()
    args = parser.parse_args(argv)
    if args.command == "infer":
        run_inference(args)
    elif args.command == "test":
        test_installation()
    elif args.command == "demo":
        _fail("ufm demo (Gradio) is not part of ufm_amd (SURVEY section 2 row 8: out of scope)")
    else:
        parser.print_help()


if __name__ == "__main__":
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    main()
