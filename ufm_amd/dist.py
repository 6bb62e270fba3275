"""Batch-split data parallelism over the pair dimension (one process per GPU).

Every pair is independent (the reference only ever concatenates / chunks along dim 0,
``uniflowmatch/models/ufm.py:308,313``), so N GPUs = N contiguous shards of the pair batch with
replicated weights and no data-path collective.  The only exchange is the trivial result gather:
ONE ``all_gather`` per batch of a packed ``[flow(2) | covisibility(1)]`` fp32 buffer
(RCCL over xGMI on GPUs: backend "nccl"; "gloo" for the CPU tests of this logic).

``predict_sharded`` is the synchronous form; ``ShardedPredictor`` is the same path with the gather issued
asynchronously into a ring of buffers, so that step i's gather (xGMI) overlaps step i+1's compute --
``bench.py --gpus N`` runs exactly this class.
"""

from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

Predict = Callable[[torch.Tensor, torch.Tensor], Tuple[torch.Tensor, torch.Tensor]]


def shard_bounds(n_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of ``n_pairs`` for ``rank``; sizes differ by at most one pair
    (ranks past ``n_pairs`` get an empty shard and still take part in the gather)."""
    base, extra = divmod(n_pairs, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_result(flow: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """(b,2,H,W) + (b,H,W) -> (b,3,H,W): one buffer, one collective."""
    return torch.cat([flow, mask.unsqueeze(1)], dim=1).contiguous()


def _unpack(gathered: torch.Tensor, n_pairs: int, world: int, max_b: int, stride: int = 0) -> torch.Tensor:
    """``stride``: rows per rank in ``gathered`` (max_b, or max_b + 1 with ShardedPredictor's status row)."""
    stride = stride or max_b
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_pairs, r, world)
        parts.append(gathered[r * stride : r * stride + (hi - lo)])
    return torch.cat(parts, dim=0)


class ShardFailed(RuntimeError):
    """``predict`` raised on at least one rank of the gather (``.ranks``: which; the local exception is chained on its rank)."""

    def __init__(self, ranks, ticket):
        super().__init__(f"predict failed on rank(s) {ranks} for ticket {ticket}; the gathered results are not valid")
        self.ranks, self.ticket = ranks, ticket


POISON = 1.0e30  # written into the flag row of a failing rank's pad (results are flow in pixels / masks in [0, 1])


class ShardedPredictor:
    """Runs ``predict`` on this rank's shard of every global batch and gathers all ranks' results.

    ``submit(source, target)`` launches the shard's compute and the (asynchronous) all_gather of its packed
    result into ring slot ``i % depth``; ``result(i)`` waits for that gather and returns ``(flow, mask)`` of ALL
    pairs in global order.  A slot is reused ``depth`` submits later (its gather is waited for first).
    A rank whose shard is empty (fewer pairs than ranks) skips ``predict`` and contributes a zero pad, so no rank
    ever misses the collective.

    Failure is collective on EVERY path.  A rank whose ``predict`` RAISES still joins the collective: its pad carries
    a poison flag (one extra fp32 row per rank).  The gathered flag rows are identical on all ranks, and every path
    that retires a slot -- ``result``, ``wait`` (``check=True`` is the default), ``drain`` and the slot reuse inside
    ``submit`` -- reads them, so all ranks raise ``ShardFailed`` for the same ticket at the same point of their
    (identical) call sequences, BEFORE any of them enters a further collective: nobody is left blocked in an
    all_gather until it times out.  On a GPU the flag rows are copied to pinned host memory by a side stream right
    behind the gather, so reading them ``depth`` submits later costs an (already complete) event wait, not a stall of
    the compute stream."""

    def __init__(self, predict: Predict, group=None, depth: int = 2):
        self.predict, self.group, self.depth = predict, group, depth
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._slots: List[Optional[dict]] = [None] * depth
        self._count = 0
        self._side = None  # CUDA side stream of the flag copies
        self.gather_ms: List[float] = []  # per retired ticket (GPU only): own shard packed -> all ranks' results here (rank skew + transfer)

    def _buffers(self, slot: int, max_b: int, tail: Tuple[int, ...], like: torch.Tensor) -> dict:
        s = self._slots[slot]
        shape = (max_b + 1,) + tail  # row max_b = this rank's status row (all POISON when its predict raised)
        if s is None or s["pad"].shape != shape or s["pad"].device != like.device:
            s = dict(
                pad=torch.zeros(shape, dtype=torch.float32, device=like.device),
                out=torch.empty((self.world * (max_b + 1),) + tail, dtype=torch.float32, device=like.device),
                work=None, n=0, max_b=max_b, error=None, ticket=-1, flags_host=None, flags_event=None,
            )
            if like.is_cuda:
                s["flags_host"] = torch.zeros(self.world, dtype=torch.float32).pin_memory()
                s["flags_event"] = torch.cuda.Event(enable_timing=True)
                s["compute_done"] = torch.cuda.Event(enable_timing=True)
            self._slots[slot] = s
        return s

    def submit(self, source: torch.Tensor, target: torch.Tensor, out_hw: Optional[Tuple[int, int]] = None) -> int:
        """``source``/``target`` hold the GLOBAL batch (leading dim = all pairs).  ``out_hw``: the result's (H, W),
        needed only by a rank whose shard is empty (defaults to the source image size)."""
        n = int(source.shape[0])
        ticket = self._count
        slot = ticket % self.depth
        if self._slots[slot] is not None and (self._slots[slot]["work"] is not None or not self._slots[slot].get("checked", True)):
            # the buffers of `depth` submits ago are free again.  A failure of that submit on ANY rank raises here, on
            # every rank, before anybody joins the new collective
            self._finish(self._slots[slot], True)
        lo, hi = shard_bounds(n, self.rank, self.world)
        max_b = -(-n // self.world)
        def default_hw():
            if out_hw is not None:
                return tuple(out_hw)
            return tuple(source.shape[1:3]) if source.shape[-1] == 3 else tuple(source.shape[-2:])

        error = None
        if hi > lo:
            try:
                flow, mask = self.predict(source[lo:hi], target[lo:hi])
            except Exception as exc:  # still join the collective, flagged; raised from result()/wait()/drain() on EVERY rank
                error = exc
        if hi > lo and error is None:
            tail = (3,) + tuple(flow.shape[2:])
            s = self._buffers(slot, max_b, tail, flow)
            s["pad"][: hi - lo, :2].copy_(flow)
            s["pad"][: hi - lo, 2].copy_(mask)
            filled = hi - lo
        else:
            s = self._buffers(slot, max_b, (3,) + default_hw(), source)
            filled = 0
        s["pad"][filled:].zero_()
        if error is not None:
            s["pad"][max_b].fill_(POISON)
        s["n"], s["error"], s["ticket"], s["checked"] = n, error, ticket, False
        if s["flags_event"] is not None:
            s["compute_done"].record()  # this rank's shard is packed; what follows until flags_event is the gather (skew + xGMI)
        s["work"] = dist.all_gather_into_tensor(s["out"], s["pad"], group=self.group, async_op=True)
        if s["flags_event"] is not None:  # flag rows -> pinned host memory, on a side stream, right behind the gather
            if self._side is None:
                self._side = torch.cuda.Stream(device=s["out"].device)
            with torch.cuda.stream(self._side):
                s["work"].wait()  # stream-level: only the side stream waits for the gather here
                flags = s["out"].view(self.world, max_b + 1, -1)[:, max_b, 0]
                s["flags_host"].copy_(flags, non_blocking=True)
                s["flags_event"].record(self._side)
        self._count += 1
        return ticket

    def _finish(self, s: dict, check: bool) -> None:
        """Complete the slot's gather; with ``check`` read every rank's status row and raise ``ShardFailed`` -- on all
        ranks alike, the rows being the gathered ones -- if any of them failed (the local exception is chained on its
        own rank).  ``check=False`` only waits (stream-level on nccl) and leaves the slot unchecked: the next path that
        retires it (slot reuse in ``submit``, ``result``, ``drain``) still raises."""
        if s["work"] is not None:
            s["work"].wait()
            s["work"] = None
        if check and not s.get("checked", False):
            if s["flags_event"] is not None:
                s["flags_event"].synchronize()
                flags = s["flags_host"].tolist()
                self.gather_ms.append(s["compute_done"].elapsed_time(s["flags_event"]))
            else:
                stride = s["max_b"] + 1
                flags = s["out"].view(self.world, stride, -1)[:, s["max_b"], 0].tolist()
            bad = [r for r, v in enumerate(flags) if v >= 0.5 * POISON]  # (fp32(1e30) != the Python double 1e30)
            s["checked"] = True
            err, s["error"] = s["error"], None
            if bad:
                raise ShardFailed(bad, s["ticket"]) from err

    def wait(self, ticket: int, check: bool = True) -> None:
        """Wait for the gather of ``ticket`` without assembling the result.  ``check=True`` (default) reads the gathered
        status rows and raises ``ShardFailed`` on EVERY rank if any rank's ``predict`` failed; ``check=False`` is a pure
        stream-level wait (the failure then surfaces, still on every rank, when the slot is next retired)."""
        assert self._count - self.depth <= ticket < self._count, "ticket is no longer (or not yet) in the ring"
        self._finish(self._slots[ticket % self.depth], check)

    def result(self, ticket: int) -> Tuple[torch.Tensor, torch.Tensor]:
        assert self._count - self.depth <= ticket < self._count, "ticket is no longer (or not yet) in the ring"
        s = self._slots[ticket % self.depth]
        self._finish(s, True)
        full = _unpack(s["out"], s["n"], self.world, s["max_b"], s["max_b"] + 1)
        return full[:, :2], full[:, 2]

    def drain(self) -> None:
        """Complete every gather in flight; raises ``ShardFailed`` if any rank failed on any of them."""
        failed = None
        for s in self._slots:
            if s is not None and (s["work"] is not None or not s.get("checked", True)):
                try:
                    self._finish(s, True)
                except ShardFailed as exc:
                    failed = failed or exc
        if failed is not None:
            raise failed


def gather_results(packed_local: torch.Tensor, n_pairs: int, group=None) -> torch.Tensor:
    """All ranks end up with the packed results of all ``n_pairs`` pairs in global pair order (synchronous)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    max_b = -(-n_pairs // world)
    lo, hi = shard_bounds(n_pairs, rank, world)
    assert packed_local.shape[0] == hi - lo
    pad = torch.zeros((max_b,) + tuple(packed_local.shape[1:]), dtype=packed_local.dtype, device=packed_local.device)
    pad[: hi - lo] = packed_local
    out = torch.empty((world * max_b,) + tuple(packed_local.shape[1:]), dtype=packed_local.dtype, device=packed_local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return _unpack(out, n_pairs, world, max_b)


def predict_sharded(
    predict: Predict, source: torch.Tensor, target: torch.Tensor, group=None, out_hw: Optional[Tuple[int, int]] = None
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Run ``predict`` on this rank's shard of the global batch and gather everyone's results.
    ``predict(src, tgt) -> (flow (b,2,H,W), covisibility (b,H,W))``."""
    sp = ShardedPredictor(predict, group, depth=1)
    return sp.result(sp.submit(source, target, out_hw))
