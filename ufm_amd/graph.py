"""HIP-graph replay of ``predict_correspondences_batched`` for one (batch, layout, dtype, size) signature.

At batch 1 the ~700 ctypes launches of a forward cost as much host time as the kernels take on the GPU
("Launch path"); every C-ABI entry point is asynchronous on the current stream, allocates nothing and
never synchronises, so the whole call is capturable.  ``GraphedPredictor`` warms the engine up eagerly (tables,
workspace, packed weights), captures one call into a ``torch.cuda.CUDAGraph`` (a hipGraph on ROCm) with static
input / output buffers, and afterwards ``__call__`` = two device copies + one graph launch.  Results are bitwise
those of the eager call (same kernels, same order, same workspace; tests/test_model_gpu.py).

The captured graph holds raw pointers into the workspace and the packed weights of the Engine it was captured on.  That
Engine is PRIVATE to the GraphedPredictor: eager calls on the same model (another batch size or resolution reallocates the
shared engine's workspace, ``set_numerics`` drops it) never touch the graph's buffers.  The packed weights themselves are
immutable and shared between all engines of a model (``engine._PACK_CACHE``: one copy per model and numerics; the private
engine keeps them alive for as long as the graph exists).  What a replay cannot follow is a
change of the model's parameters: ``__call__`` compares the parameters' versions and storage with those at capture time and
raises instead of replaying stale weights.

The returned tensors are the graph's static outputs: they are overwritten by the next call (clone to keep).
"""

from __future__ import annotations

from typing import Optional

import torch

from .base import UFMOutputInterface


class GraphedPredictor:
    def __init__(self, model, source_image: torch.Tensor, target_image: torch.Tensor, data_norm_type: Optional[str] = None, warmup: int = 2):
        if not (source_image.is_cuda and target_image.is_cuda):
            raise RuntimeError("GraphedPredictor needs device tensors (ufm_amd has no CPU path)")
        self.model, self.norm = model, data_norm_type
        self._src = source_image.clone()
        self._tgt = target_image.clone()
        from .engine import Engine

        shared = model._engine
        eng = self._engine = Engine(model, model.numerics)  # private workspace: nothing else ever runs on it (packed weights: shared, immutable)
        if shared is not None:
            eng.concurrent_heads, eng.fused_tail = shared.concurrent_heads, shared.fused_tail
            eng.group_heads, eng.last_layer_view1, eng.conv_splitk = shared.group_heads, shared.last_layer_view1, shared.conv_splitk
            eng.level_streams, eng.level_streams_max_images = shared.level_streams, shared.level_streams_max_images
        eng.micro_batches = 1  # one stream, one host thread: the capture records a single linear launch sequence
        model._engine = eng
        try:
            side = torch.cuda.Stream(device=source_image.device)
            side.wait_stream(torch.cuda.current_stream(source_image.device))
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):  # builds tables / workspace / packed weights outside the capture
                    model.predict_correspondences_batched(self._src, self._tgt, data_norm_type)
            torch.cuda.current_stream(source_image.device).wait_stream(side)
            torch.cuda.synchronize(source_image.device)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._out = model.predict_correspondences_batched(self._src, self._tgt, data_norm_type)
        finally:
            model._engine = shared  # eager calls go back to the shared engine (created on demand if there was none)
        # The guard below runs on EVERY replay, in front of the graph launch: walking the module tree (model.parameters()) cost
        # 0.3-0.9 ms of host time per call, 4-9 % of the one-pair latency; reading version + address of a cached list costs 0.08 ms.
        # A Parameter OBJECT swapped into a sub-module after the capture shows in the private engine's per-model epoch (torch's global
        # registration hooks, counted only for modules of THIS model's tree: engine._TRACKED); in-place edits, load_state_dict and .to() in
        # version / address.  (Edits made through ``p.data`` bump nothing: they are invisible to any guard.)
        self._epoch_ref, self._epoch = eng._epoch, eng._epoch.n
        self._plist = list(self.model.parameters())
        self._weights_key = self._params_key()

    def _params_key(self):
        if self._epoch_ref.n != self._epoch:  # something was registered on a module of this model's tree: did a Parameter object change?
            from .engine import _track_tree

            fresh = list(self.model.parameters())
            if len(fresh) != len(self._plist) or any(a is not b for a, b in zip(fresh, self._plist)):
                return None  # never equal to the captured key
            _track_tree(self.model, self._epoch_ref)  # (a swapped-in sub-module's own children count from here on)
            self._epoch = self._epoch_ref.n
        return tuple((p._version, p.data_ptr()) for p in self._plist)

    def __call__(self, source_image: torch.Tensor, target_image: torch.Tensor) -> UFMOutputInterface:
        if source_image.shape != self._src.shape or source_image.dtype != self._src.dtype or target_image.shape != self._tgt.shape:
            raise ValueError("GraphedPredictor was captured for a different input signature")
        if self._params_key() != self._weights_key:
            raise RuntimeError("the model's parameters changed (or moved) after the graph was captured: capture a new GraphedPredictor")
        self._src.copy_(source_image, non_blocking=True)
        self._tgt.copy_(target_image, non_blocking=True)
        self.graph.replay()
        return self._out
