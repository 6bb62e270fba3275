"""ufm_amd -- MI355X-native (gfx950) implementation of the UFM dense-correspondence inference hot path.

Drop-in surface (same names as ``uniflowmatch``): ``UniFlowMatch``, ``UniFlowMatchConfidence``
(UFM-Base), ``UniFlowMatchClassificationRefinement`` (UFM-Refine), the output dataclasses, and
``predict_correspondences_batched`` / ``from_pretrained``.  Compute = hand-written HIP kernels in
``ufm_amd/csrc`` behind the C ABI of ``include/ufm_hip.h``; there is no PyTorch/CPU fallback.
"""

from .base import (
    UFMClassificationRefinementOutput,
    UFMFlowFieldOutput,
    UFMMaskFieldOutput,
    UFMOutputInterface,
    UniFlowMatchModelsBase,
)
from .configs import make_config, ufm_base_config, ufm_refine_config, ufm_tiny_config
from .graph import GraphedPredictor
from .ufm import UniFlowMatch, UniFlowMatchClassificationRefinement, UniFlowMatchConfidence

__all__ = [
    "UFMClassificationRefinementOutput",
    "UFMFlowFieldOutput",
    "UFMMaskFieldOutput",
    "UFMOutputInterface",
    "UniFlowMatchModelsBase",
    "UniFlowMatch",
    "UniFlowMatchClassificationRefinement",
    "UniFlowMatchConfidence",
    "GraphedPredictor",
    "make_config",
    "ufm_base_config",
    "ufm_refine_config",
    "ufm_tiny_config",
]
