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


def _failing_worker(rank, world, port, n_pairs, fail_rank, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ufm_amd.dist import ShardedPredictor, ShardFailed

    g = torch.Generator().manual_seed(0)
    src = torch.randint(0, 256, (n_pairs, 6, 5, 3), dtype=torch.uint8, generator=g)
    tgt = torch.randint(0, 256, (n_pairs, 6, 5, 3), dtype=torch.uint8, generator=g)
    calls = [0]

    def predict(s, t):
        calls[0] += 1
        if rank == fail_rank and calls[0] == 2:  # the second step fails on one rank only
            raise ValueError("boom")
        return _fake_predict(s, t)

    sp = ShardedPredictor(predict, depth=2)
    t0 = sp.submit(src, tgt)
    t1 = sp.submit(src, tgt)  # fail_rank's predict raises in here; submit must still join the collective
    f, m = sp.result(t0)      # the healthy step is intact on every rank
    rf, rm = _fake_predict(src, tgt)
    ok0 = torch.equal(f, rf) and torch.equal(m, rm)
    try:
        sp.result(t1)
        got = None
    except ShardFailed as exc:
        got = (list(exc.ranks), type(exc.__cause__).__name__ if exc.__cause__ is not None else None)
    t2 = sp.submit(src, tgt)  # the predictor stays usable after a failed step
    f2, m2 = sp.result(t2)
    ok2 = torch.equal(f2, rf) and torch.equal(m2, rm)
    sp.drain()
    ret[rank] = (ok0, got, ok2)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,fail_rank", [(2, 1), (3, 0)])
def test_a_failing_rank_poisons_the_gather_instead_of_blocking_the_others(world, fail_rank):
    """ShardedPredictor.submit: an exception inside predict on ONE rank must not leave the others blocked in the
    all_gather -- the failing rank joins the collective with its status row poisoned and every rank raises ShardFailed
    from result() of that step (and only that step)."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, 6, fail_rank, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(world):
        ok0, got, ok2 = ret[r]
        assert ok0 and ok2, r
        assert got is not None, f"rank {r} did not see the failure"
        ranks, cause = got
        assert ranks == [fail_rank]
        assert cause == ("ValueError" if r == fail_rank else None)  # the local exception is chained on its own rank


def _failing_paths_worker(rank, world, port, n_pairs, fail_rank, path, ret):
    """The failing step is never consumed through result(): the failure has to surface -- on EVERY rank, at the same call --
    from wait() (bench.py's path) or from the slot reuse inside submit(), before anybody enters a further collective."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ufm_amd.dist import ShardedPredictor, ShardFailed

    g = torch.Generator().manual_seed(0)
    src = torch.randint(0, 256, (n_pairs, 6, 5, 3), dtype=torch.uint8, generator=g)
    tgt = torch.randint(0, 256, (n_pairs, 6, 5, 3), dtype=torch.uint8, generator=g)
    calls = [0]

    def predict(s, t):
        calls[0] += 1
        if rank == fail_rank and calls[0] == 2:
            raise ValueError("boom")
        return _fake_predict(s, t)

    sp = ShardedPredictor(predict, depth=2)
    raised_at, ranks = None, None
    try:
        t0 = sp.submit(src, tgt)
        t1 = sp.submit(src, tgt)  # fails on fail_rank, joins the gather poisoned
        if path == "wait":
            sp.wait(t0)           # healthy step
            raised_at = "wait"
            sp.wait(t1)           # default check=True: every rank raises here
            raised_at = None
        else:
            sp.wait(t0, check=False)
            sp.wait(t1, check=False)  # pure waits: nobody has looked at the flags yet
            t2 = sp.submit(src, tgt)  # reuses slot of t0 (healthy, unchecked so far)
            raised_at = "submit"
            sp.submit(src, tgt)       # reuses the slot of t1: every rank raises here, before the new all_gather
            raised_at = None
    except ShardFailed as exc:
        ranks = list(exc.ranks)
    # after the collective failure every rank is at the same point: a further step works
    t = sp.submit(src, tgt)
    f, m = sp.result(t)
    rf, rm = _fake_predict(src, tgt)
    sp.drain()
    ret[rank] = (raised_at, ranks, bool(torch.equal(f, rf) and torch.equal(m, rm)))
    dist.destroy_process_group()


@pytest.mark.parametrize("path", ["wait", "submit"])
def test_failure_is_collective_on_the_wait_and_slot_reuse_paths(path):
    world, fail_rank = 3, 1
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_failing_paths_worker, args=(r, world, port, 6, fail_rank, path, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0  # in particular: no rank hung in an all_gather the failing rank never joined
    for r in range(world):
        raised_at, ranks, ok_after = ret[r]
        assert raised_at == path and ranks == [fail_rank], (r, raised_at, ranks)
        assert ok_after, r


def _tagging_predict(src, tgt):
    """Stub predict for the config-3 partition test: every output pixel carries the pair's id (channel 0 of pixel (0,0) of the
    source holds it) and a step tag (from the target), so global order and ring reuse are visible in the gathered result."""
    pid = src[:, 0, 0, 0].float()
    step = tgt[:, 0, 0, 0].float()
    b, h, w = src.shape[0], src.shape[1], src.shape[2]
    flow = torch.stack([pid.view(b, 1, 1).expand(b, h, w), step.view(b, 1, 1).expand(b, h, w)], dim=1).contiguous()
    mask = (pid * 0.5 + step).view(b, 1, 1).expand(b, h, w).contiguous()
    return flow, mask


def _config3_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ufm_amd.dist import ShardedPredictor, ShardFailed, shard_bounds

    n_pairs, steps = 64, 7
    src = torch.zeros(n_pairs, 4, 4, 3, dtype=torch.uint8)
    src[:, 0, 0, 0] = torch.arange(n_pairs, dtype=torch.uint8)
    seen = []

    def predict(s, t):
        seen.append((int(s[0, 0, 0, 0]), int(s.shape[0]), int(t[0, 0, 0, 0])))
        if int(t[0, 0, 0, 0]) == 5 and rank == 6:  # step 5 fails on rank 6 only
            raise RuntimeError("rank 6 lost its GPU")
        return _tagging_predict(s, t)

    sp = ShardedPredictor(predict, depth=2)
    ok, failed_steps, last = True, [], None
    for step in range(steps):
        tgt = torch.full((n_pairs, 4, 4, 3), step, dtype=torch.uint8)
        tk = sp.submit(src, tgt)
        if last is not None:  # bench.py's loop: consume the previous step while this one's gather is in flight
            try:
                f, m = sp.result(last[0])
                ok = ok and bool((f[:, 0, 0, 0] == torch.arange(n_pairs)).all()) and bool((f[:, 1] == last[1]).all()) \
                    and bool(torch.equal(m[:, 0, 0], torch.arange(n_pairs) * 0.5 + last[1]))
            except ShardFailed as exc:
                failed_steps.append((last[1], list(exc.ranks)))
        last = (tk, step)
    f, m = sp.result(last[0])
    ok = ok and bool((f[:, 0, 0, 0] == torch.arange(n_pairs)).all()) and bool((f[:, 1] == last[1]).all())
    sp.drain()
    lo, hi = shard_bounds(n_pairs, rank, world)
    ret[rank] = (ok, failed_steps, seen, lo, hi)
    dist.destroy_process_group()


def test_config3_partition_64_pairs_over_8_ranks_ring_of_two_seven_steps():
    """BASELINE config 3's exact split (B = 64 over 8 ranks, 8 pairs each; /root/reference/uniflowmatch/models/ufm.py:308-315 is why
    the split is legal) on 8 gloo ranks: global order of the gathered pairs, reuse of the two-deep ring over 7 steps, every
    rank computing exactly its contiguous 8 pairs, and one rank failing at step 5 -- seen by all ranks for that step only."""
    world = 8
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_config3_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    for r in range(world):
        ok, failed_steps, seen, lo, hi = ret[r]
        assert (lo, hi) == (8 * r, 8 * r + 8)
        assert ok, r
        assert failed_steps == [(5, [6])], (r, failed_steps)
        assert seen == [(8 * r, 8, s) for s in range(7)], (r, seen)  # its own contiguous shard, every step, in order
