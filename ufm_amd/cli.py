#!/usr/bin/env python3
"""``ufm`` command line (reference ``uniflowmatch/cli.py``): ``ufm infer SOURCE TARGET [-o DIR] [--model base|refine]``
writes the reference's three artefacts (cli.py:124-150): ``flow_visualization.png``, ``covisibility_mask.png``,
``warped_source.png``.  Differences, all forced by this environment and stated here: images are read/written with PIL
(OpenCV is not installed); ``--weights`` points at a local ``save_pretrained`` directory or a ``.ckpt`` (there is no
network for the Hub ids the reference hard-codes; without it the Hub id is tried and fails the same way the reference
does offline); ``--random-init`` runs the architecture with seeded random weights (plumbing check).  ``ufm demo``
(Gradio) is out of scope.  ``ufm test`` is the reference's installation check restated for
this stack."""

import argparse
import sys
from pathlib import Path


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="UFM: Unified Dense Correspondence with Flow", prog="ufm")
    sub = parser.add_subparsers(dest="command", help="Available commands")
    demo = sub.add_parser("demo", help="Launch interactive Gradio demo (not built in ufm_amd)")
    demo.add_argument("--port", type=int, default=7860)
    demo.add_argument("--share", action="store_true")
    demo.add_argument("--model", choices=["base", "refine"], default="base")
    infer = sub.add_parser("infer", help="Run inference on image pairs")
    infer.add_argument("source", help="Source image path")
    infer.add_argument("target", help="Target image path")
    infer.add_argument("--output", "-o", help="Output directory (default: current directory)")
    infer.add_argument("--model", choices=["base", "refine"], default="base", help="Model variant to use (default: base)")
    infer.add_argument("--weights", help="local save_pretrained directory or .ckpt file (default: the reference's Hub id)")
    infer.add_argument("--random-init", action="store_true", help="seeded random weights instead of a checkpoint")
    infer.add_argument("--numerics", choices=["fast", "parity"], default="fast")
    sub.add_parser("test", help="Test installation")
    return parser


def load_model(args):
    import ufm_amd
    from ufm_amd.modules import init_weights_

    cls = ufm_amd.UniFlowMatchClassificationRefinement if args.model == "refine" else ufm_amd.UniFlowMatchConfidence
    if args.random_init:
        cfg = ufm_amd.ufm_refine_config() if args.model == "refine" else ufm_amd.ufm_base_config()
        model = cls(**cfg)
        init_weights_(model, seed=0)
    elif args.weights and str(args.weights).endswith(".ckpt"):
        model = cls.from_pretrained_ckpt(args.weights)
    else:
        hub_id = "infinity1096/UFM-Refine" if args.model == "refine" else "infinity1096/UFM-Base"  # cli.py:109-111
        model = cls.from_pretrained(args.weights or hub_id)
    return model.eval().to("cuda").set_numerics(args.numerics)


def run_inference(args) -> None:
    """cli.py:85-156."""
    try:
        import numpy as np
        import torch

        from ufm_amd import viz

        try:
            source_rgb, target_rgb = viz.load_rgb(args.source), viz.load_rgb(args.target)
        except (FileNotFoundError, OSError):
            print("Error: Could not load one or both images")
            sys.exit(1)
        model = load_model(args)
        print("Running inference...")
        with torch.no_grad():
            result = model.predict_correspondences_batched(
                source_image=torch.from_numpy(source_rgb).to("cuda"), target_image=torch.from_numpy(target_rgb).to("cuda")
            )
            flow = result.flow.flow_output[0].cpu().numpy()
            covisibility = result.covisibility.mask[0].cpu().numpy()
        output_dir = Path(args.output) if args.output else Path.cwd()
        output_dir.mkdir(exist_ok=True)
        viz.save_png(output_dir / "flow_visualization.png", viz.flow_to_color(flow.transpose(1, 2, 0)))
        viz.save_png(output_dir / "covisibility_mask.png", (covisibility * 255).astype(np.uint8))
        warped = viz.warp_image_with_flow(source_rgb, None, target_rgb, flow.transpose(1, 2, 0))
        warped = covisibility[..., None] * warped + (1 - covisibility[..., None]) * 255 * np.ones_like(warped)
        viz.save_png(output_dir / "warped_source.png", warped)
        print(f"Results saved to: {output_dir}")
        print("- flow_visualization.png")
        print("- covisibility_mask.png")
        print("- warped_source.png")
    except ImportError as e:
        print(f"Error importing dependencies: {e}")
        print("Please ensure all dependencies are installed")
        sys.exit(1)
    except SystemExit:
        raise
    except Exception as e:  # the reference reports and exits 1 (cli.py:154-156)
        print(f"Error during inference: {e}")
        sys.exit(1)


def test_installation() -> None:
    """cli.py:159-212 for this stack: imports, the HIP library and its ABI, the GPU."""
    print("Testing UFM installation...")
    try:
        import numpy
        import torch

        print(f"✓ PyTorch {torch.__version__}")
        print(f"✓ NumPy {numpy.__version__}")
        import PIL

        print(f"✓ Pillow {PIL.__version__}")
        from ufm_amd import hip
        from ufm_amd.ufm import UniFlowMatchConfidence  # noqa: F401

        print("✓ UFM model imports")
        lib = hip.lib()
        print(f"✓ libufm_hip.so ABI {lib.ufm_abi_version()} built for {lib.ufm_built_arch().decode()}")
        if torch.cuda.is_available():
            print(f"✓ ROCm {torch.version.hip} available")
            print(f"  GPU: {torch.cuda.get_device_name(0)}")
        else:
            print("⚠ no GPU visible: ufm_amd has no CPU path, inference will raise")
        print("\n✅ Installation test completed successfully!")
    except ImportError as e:
        print(f"❌ Import error: {e}")
        print("Please check your installation")
        sys.exit(1)
    except Exception as e:
        print(f"❌ Unexpected error: {e}")
        sys.exit(1)


def main(argv=None) -> None:
    parser = build_parser()
    args = parser.parse_args(argv)
    if args.command == "demo":
        print("ufm demo (Gradio) is not part of ufm_amd (SURVEY section 2 row 8: out of scope)")
        sys.exit(1)
    elif args.command == "infer":
        run_inference(args)
    elif args.command == "test":
        test_installation()
    else:
        parser.print_help()


if __name__ == "__main__":
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    main()
