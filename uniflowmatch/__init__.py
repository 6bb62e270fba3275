"""Import-path alias so reference callers (``from uniflowmatch import UniFlowMatchConfidence``,
``from uniflowmatch.models.ufm import ...``) resolve to the MI355X-native implementation in
``ufm_amd``.  See INTEGRATION.md."""

from ufm_amd import UniFlowMatch, UniFlowMatchClassificationRefinement, UniFlowMatchConfidence

__all__ = ["UniFlowMatch", "UniFlowMatchClassificationRefinement", "UniFlowMatchConfidence"]
