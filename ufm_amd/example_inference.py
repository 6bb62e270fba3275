#!/usr/bin/env python3
"""Example runner with the 2 x 3 result figure -- the MI355X counterpart of the reference's ``example_inference.py``
(load a model, ``predict_correspondences_batched`` on one image pair, save a figure with source / target / warped source
on the top row and flow colouring / thresholded covisibility / covisibility confidence below; example_inference.py:23-134).

    python -m ufm_amd.example_inference --source a.png --target b.png --weights DIR_OR_CKPT [--model refine] [-o out.png]
    python -m ufm_amd.example_inference --source a.png --target b.png --random-init      # plumbing check, no checkpoint

Differences forced by this environment (same as ``ufm_amd.cli``): PIL instead of OpenCV for image I/O, the repo's own
restatement of ``flow_vis.flow_to_color`` (``ufm_amd.viz``; flow_vis is not installed: parity unpinned), and ``--weights``
/ ``--random-init`` because the Hub ids the reference hard-codes (example_inference.py:110-112) need a network.  The
image warp runs on the GPU (``ufm_warp_bilinear``), the figure is drawn by matplotlib on the host.
"""

from __future__ import annotations

import argparse

import numpy as np
import torch

from . import viz
from .cli import HUB_IDS, load_model


def load_image(image_path) -> np.ndarray:
    """RGB uint8 HWC (example_inference.py:23-28)."""
    try:
        return viz.load_rgb(image_path)
    except OSError as exc:
        raise ValueError(f"Could not load image: {image_path}") from exc


def predict_correspondences(model, source_image: np.ndarray, target_image: np.ndarray):
    """(flow (2,H,W), covisibility (H,W)) as numpy arrays (example_inference.py:31-42)."""
    with torch.no_grad():
        result = model.predict_correspondences_batched(
            source_image=torch.from_numpy(source_image).to("cuda"), target_image=torch.from_numpy(target_image).to("cuda")
        )
    return result.flow.flow_output[0].cpu().numpy(), result.covisibility.mask[0].cpu().numpy()


def compose_warp(source_image, target_image, flow_output, covisibility) -> np.ndarray:
    """Target warped into the source frame, white where the pair does not overlap, in [0, 1] (example_inference.py:59-64)."""
    warped = viz.warp_image_with_flow(source_image, None, target_image, flow_output.transpose(1, 2, 0))
    warped = covisibility[..., None] * warped + (1 - covisibility[..., None]) * 255 * np.ones_like(warped)
    return np.clip(warped / 255.0, 0, 1)


def visualize_results(source_image, target_image, flow_output, covisibility, output_path="ufm_output.png", warped_image=None):
    """The reference's 2 x 3 figure (example_inference.py:45-90).  ``warped_image`` may be passed in (tests without a GPU)."""
    import matplotlib

    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt

    fig, axs = plt.subplots(2, 3, figsize=(15, 10))
    if warped_image is None:
        warped_image = compose_warp(source_image, target_image, flow_output, covisibility)
    panels = (
        (axs[0, 0], source_image, "Source Image", {}),
        (axs[0, 1], target_image, "Target Image", {}),
        (axs[0, 2], warped_image, "Warped Source Image", {}),
        (axs[1, 0], viz.flow_to_color(flow_output.transpose(1, 2, 0)), "Flow Visualization (Valid at Covisible Pixels)", {}),
        (axs[1, 1], covisibility > 0.5, "Covisibility Mask (>0.5)", dict(cmap="gray", vmin=0, vmax=1)),
    )
    for ax, img, title, kw in panels:
        ax.imshow(img, **kw)
        ax.set_title(title)
        ax.axis("off")
    heatmap = axs[1, 2].imshow(covisibility, cmap="viridis", vmin=0, vmax=1)
    axs[1, 2].set_title("Covisibility Confidence")
    axs[1, 2].axis("off")
    plt.colorbar(heatmap, ax=axs[1, 2], shrink=0.6)
    plt.tight_layout()
    plt.savefig(output_path, dpi=150, bbox_inches="tight")
    print(f"Visualization saved to: {output_path}")
    return fig


def main(argv=None) -> None:
    parser = argparse.ArgumentParser(description="UFM inference example (MI355X)")
    parser.add_argument("--source", "-s", default="examples/image_pairs/fire_academy_0.png", help="Path to source image")
    parser.add_argument("--target", "-t", default="examples/image_pairs/fire_academy_1.png", help="Path to target image")
    parser.add_argument("--model", choices=sorted(HUB_IDS), default="base", help="Model variant to use")
    parser.add_argument("--output", "-o", default="ufm_output.png", help="Output visualization path")
    parser.add_argument("--show", action="store_true", help="Display the visualization")
    parser.add_argument("--weights", help="local save_pretrained directory or .ckpt file (default: the reference's Hub id)")
    parser.add_argument("--random-init", action="store_true", help="seeded random weights instead of a checkpoint")
    parser.add_argument("--numerics", choices=("fast", "parity"), default="fast")
    args = parser.parse_args(argv)

    print(f"Loading UFM {args.model} model...")
    model = load_model(args)
    print("Model loaded successfully!")
    print(f"Loading images: {args.source}, {args.target}")
    source_image, target_image = load_image(args.source), load_image(args.target)
    print(f"Image shapes: {source_image.shape}, {target_image.shape}")
    print("Running inference...")
    flow_output, covisibility = predict_correspondences(model, source_image, target_image)
    visualize_results(source_image, target_image, flow_output, covisibility, args.output)
    if args.show:
        import matplotlib.pyplot as plt

        plt.show()
    print("Inference completed!")


if __name__ == "__main__":
    main()
