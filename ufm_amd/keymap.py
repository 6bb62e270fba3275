"""Checkpoint key handling, in ONE place.

What the reference does with checkpoint keys (uniflowmatch/models/ufm.py):
  * Lightning training checkpoints (``"state_dict"`` present): keep the keys under the ``model.`` prefix, strip it, DROP
    ``feature_matching_proj*`` and ``encoder.model.mask_token`` (``modify_state_dict``, ufm.py:85-117, :203-211), load strict;
  * plain checkpoints (``"model"``): ``load_state_dict(strict=False)`` and ASSERT that no key is missing (ufm.py:212-217);
  * Hub weights (``model.safetensors`` via ``PyTorchModelHubMixin``): the mixin's default is ``strict=False`` with NO check,
    which would leave any differently-named parameter at its random initial value without a word.

``ufm_amd`` loads all three through ``load_checked``: after the renames below, NO parameter of the model may be missing
from the file, and the file may only carry extra keys that are on ``ALLOWED_UNEXPECTED`` (buffers the reference's
third-party modules register and this implementation does not need).  Anything else raises.

Parity note: the sub-module key names under ``encoder.*`` / ``info_sharing.*`` / ``head1.*`` / ``uncertainty_head.*`` /
``classification_head.*`` come from the third-party ``uniception`` package, which is absent from the reference mount
(empty submodule) -- they are restated from the published module layouts and **unpinned**; ``RENAMES`` is where a
difference found against a real checkpoint gets recorded, and the strict check is what makes such a difference loud.
"""

from __future__ import annotations

import re
from typing import Dict, Iterable, List, Mapping, Optional, Tuple

import torch

LIGHTNING_PREFIX = "model."  # ufm.py:205-207

# substring -> replacement (None = drop); first matching rule wins (modify_state_dict semantics, ufm.py:85-117)
DROP_RULES: Dict[str, Optional[str]] = {
    "feature_matching_proj": None,     # ufm.py:209
    "encoder.model.mask_token": None,  # ufm.py:209 (DINOv2 mask token: unused at inference)
}

# renames from a released checkpoint's names to this implementation's (none known: see the parity note above)
RENAMES: Dict[str, Optional[str]] = {}

# extra keys a checkpoint may carry without being an error (regular expressions, full match)
# -- exactly what the reference itself drops before its strict load (ufm.py:203-211) plus BatchNorm's step counter
ALLOWED_UNEXPECTED: Tuple[str, ...] = (
    r"encoder\.model\.mask_token",
    r".*\.num_batches_tracked",
    r"feature_matching_proj\..*",
)

# keys that identify an encoder / info-sharing VARIANT this build does not implement: loading such a checkpoint into the
# plain DINOv2 encoder would run without error and produce wrong outputs, so they raise by name
UNSUPPORTED_VARIANT_KEYS: Tuple[Tuple[str, str], ...] = (
    (r"encoder\.model\.register_tokens", "a DINOv2-with-registers encoder (register tokens are not implemented)"),
)


def modify_state_dict(original_state_dict: Mapping[str, torch.Tensor], mappings: Mapping[str, Optional[str]]) -> Dict[str, torch.Tensor]:
    """Rename / drop checkpoint keys by substring (ufm.py:85-117): first matching rule wins, None drops."""
    out: Dict[str, torch.Tensor] = {}
    for key, value in original_state_dict.items():
        new_key, drop = key, False
        for old, new in mappings.items():
            if old in key:
                if new is None:
                    drop = True
                else:
                    new_key = key.replace(old, new)
                break
        if not drop:
            out[new_key] = value
    return out


def normalise(state_dict: Mapping[str, torch.Tensor], lightning: bool = False) -> Dict[str, torch.Tensor]:
    """Checkpoint keys -> this implementation's state-dict namespace."""
    sd: Mapping[str, torch.Tensor] = state_dict
    if lightning:
        sd = {k[len(LIGHTNING_PREFIX):]: v for k, v in sd.items() if k.startswith(LIGHTNING_PREFIX)}
    sd = modify_state_dict(sd, DROP_RULES)
    if RENAMES:
        sd = modify_state_dict(sd, RENAMES)
    return dict(sd)


def check_load_result(missing: Iterable[str], unexpected: Iterable[str], what: str = "checkpoint") -> None:
    """No parameter may be missing (ufm.py:216-217); unexpected keys must be on the allow list."""
    missing = sorted(missing)
    unexpected = list(unexpected)
    for pat, what_variant in UNSUPPORTED_VARIANT_KEYS:
        hit = [k for k in unexpected if re.fullmatch(pat, k)]
        if hit:
            raise RuntimeError(f"{what} holds {hit[0]}: it was saved from {what_variant}; loading it into this model would run and give wrong outputs")
    bad_unexpected = sorted(k for k in unexpected if not any(re.fullmatch(p, k) for p in ALLOWED_UNEXPECTED))
    if missing or bad_unexpected:
        def head(keys: List[str]) -> str:
            return ", ".join(keys[:8]) + (f", ... (+{len(keys) - 8})" if len(keys) > 8 else "")
        raise RuntimeError(
            f"{what} does not match the model: "
            + (f"{len(missing)} parameter(s) missing from the file (they would stay at their initial values): {head(missing)}. " if missing else "")
            + (f"{len(bad_unexpected)} key(s) in the file that the model does not have: {head(bad_unexpected)}. " if bad_unexpected else "")
            + "Add the rename to ufm_amd/keymap.py:RENAMES (or the key to ALLOWED_UNEXPECTED) if the file is right."
        )


def load_checked(model: torch.nn.Module, state_dict: Mapping[str, torch.Tensor], lightning: bool = False, what: str = "checkpoint"):
    sd = normalise(state_dict, lightning)
    result = model.load_state_dict(sd, strict=False)
    # Aliased parameters (the DPT head registers act_postprocess[N] and act_N_postprocess for the same tensors) are stored
    # once by safetensors' save_model: a "missing" key whose storage was filled through its alias is not missing.
    own = model.state_dict()
    filled = {own[k].data_ptr() for k in sd if k in own}
    missing = [k for k in result.missing_keys if own[k].data_ptr() not in filled]
    check_load_result(missing, result.unexpected_keys, what)
    return result
