"""Batch-split data parallelism over the pair dimension (one process per GPU).

Every pair is independent (the reference only ever concatenates / chunks along dim 0,
``uniflowmatch/models/ufm.py:308,313``), so N GPUs = N contiguous shards of the pair batch with
replicated weights and no data-path collective.  The only exchange is the trivial result gather:
ONE ``all_gather`` per batch of a packed ``[flow(2) | covisibility(1)]`` fp32 buffer
(RCCL over xGMI on GPUs: backend "nccl"; "gloo" for the CPU tests of this logic).
"""

from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of ``n_pairs`` for ``rank``; sizes differ by at most one pair."""
    base, extra = divmod(n_pairs, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_result(flow: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """(b,2,H,W) + (b,H,W) -> (b,3,H,W): one buffer, one collective."""
    return torch.cat([flow, mask.unsqueeze(1)], dim=1).contiguous()


def gather_results(packed_local: torch.Tensor, n_pairs: int, group=None) -> torch.Tensor:
    """All ranks end up with the packed results of all ``n_pairs`` pairs in global pair order."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    max_b = -(-n_pairs // world)
    lo, hi = shard_bounds(n_pairs, rank, world)
    assert packed_local.shape[0] == hi - lo
    pad = torch.zeros((max_b,) + tuple(packed_local.shape[1:]), dtype=packed_local.dtype, device=packed_local.device)
    pad[: hi - lo] = packed_local
    out = torch.empty((world * max_b,) + tuple(packed_local.shape[1:]), dtype=packed_local.dtype, device=packed_local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = []
    for r in range(world):
        l, h = shard_bounds(n_pairs, r, world)
        parts.append(out[r * max_b : r * max_b + (h - l)])
    return torch.cat(parts, dim=0)


def predict_sharded(
    predict: Callable[[torch.Tensor, torch.Tensor], Tuple[torch.Tensor, torch.Tensor]],
    source: torch.Tensor,
    target: torch.Tensor,
    group=None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Run ``predict`` on this rank's shard of the global batch and gather everyone's results.
    ``predict(src, tgt) -> (flow (b,2,H,W), covisibility (b,H,W))``."""
    n = source.shape[0]
    lo, hi = shard_bounds(n, dist.get_rank(group), dist.get_world_size(group))
    flow, mask = predict(source[lo:hi], target[lo:hi])
    full = gather_results(pack_result(flow, mask), n, group)
    return full[:, :2], full[:, 2]
