from ufm_amd.ufm import UniFlowMatch, UniFlowMatchClassificationRefinement, UniFlowMatchConfidence, modify_state_dict  # noqa: F401
