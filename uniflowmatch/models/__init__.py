from ufm_amd import (  # noqa: F401
    UFMClassificationRefinementOutput,
    UFMFlowFieldOutput,
    UFMMaskFieldOutput,
    UFMOutputInterface,
    UniFlowMatch,
    UniFlowMatchClassificationRefinement,
    UniFlowMatchConfidence,
    UniFlowMatchModelsBase,
)
