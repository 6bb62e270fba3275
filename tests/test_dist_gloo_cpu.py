"""world_size-2 (and 3, uneven shards) test of the N>1 path on CPU with the gloo backend: the
batch-split + single packed all_gather reproduces the unsharded result exactly."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _fake_predict(src, tgt):
    """Stand-in per-pair compute (the HIP model needs a GPU): any function that treats pairs independently."""
    s, t = src.float(), tgt.float()  # small integers: every op below is exact in fp32, so shard == whole bitwise
    flow = torch.stack([s.sum(dim=-1) - t.sum(dim=-1), s[..., 0] * 0.5 + t[..., 1]], dim=1)
    mask = (s - t).sum(dim=-1) / 1024.0
    return flow, mask


def _worker(rank, world, port, n_pairs, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ufm_amd.dist import predict_sharded, shard_bounds

    g = torch.Generator().manual_seed(0)
    src = torch.randint(0, 256, (n_pairs, 6, 5, 3), dtype=torch.uint8, generator=g)
    tgt = torch.randint(0, 256, (n_pairs, 6, 5, 3), dtype=torch.uint8, generator=g)
    flow, mask = predict_sharded(_fake_predict, src, tgt)
    ref_flow, ref_mask = _fake_predict(src, tgt)
    ok = torch.equal(flow, ref_flow) and torch.equal(mask, ref_mask)
    # the asynchronous double-buffered form bench.py runs: three steps in flight over a ring of two slots
    from ufm_amd.dist import ShardedPredictor

    sp = ShardedPredictor(_fake_predict, depth=2)
    tickets, batches = [], []
    for step in range(3):
        s2, t2 = src.roll(step, 0), tgt.roll(step, 0)
        batches.append((s2, t2))
        tickets.append(sp.submit(s2, t2))
        if step >= 1:  # consume the previous step while this one's gather is in flight
            f, m = sp.result(tickets[step - 1])
            rf, rm = _fake_predict(*batches[step - 1])
            ok = ok and torch.equal(f, rf) and torch.equal(m, rm)
    f, m = sp.result(tickets[-1])
    rf, rm = _fake_predict(*batches[-1])
    ok = ok and torch.equal(f, rf) and torch.equal(m, rm)
    sp.drain()
    lo, hi = shard_bounds(n_pairs, rank, world)
    ret[rank] = (ok, lo, hi)
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n_pairs", [(2, 8), (3, 7), (2, 1), (3, 2)])  # (2,1), (3,2): ranks with an EMPTY shard still join the gather
def test_sharded_predict_matches_unsharded(world, n_pairs):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_pairs, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    covered = []
    for r in range(world):
        ok, lo, hi = ret[r]
        assert ok, f"rank {r} result differs from the unsharded run"
        covered += list(range(lo, hi))
    assert covered == list(range(n_pairs))  # shards tile the batch exactly once, in order


def test_shard_bounds_properties():
    from ufm_amd.dist import shard_bounds

    for n in (1, 2, 7, 8, 64):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [h - l for l, h in b]
            assert max(sizes) - min(sizes) <= 1
