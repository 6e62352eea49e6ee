"""The N>1 path on CPU: two processes, gloo backend, the path's single collective (an all-gather of results)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dpilqr_amd.sharding import gather_results, shard_bounds


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, n, m = 5, 8, 4
    lo, hi = shard_bounds(n_items, world, rank)
    idx = torch.arange(lo, hi, dtype=torch.float64)
    # fake "solved" shard whose content encodes the global item index
    r = dict(X=idx[:, None, None] + torch.zeros(hi - lo, T + 1, n, dtype=torch.float64),
             U=-idx[:, None, None] + torch.zeros(hi - lo, T, m, dtype=torch.float64), J=idx * 10,
             status=torch.ones(hi - lo, dtype=torch.int32), n_bwd=idx.to(torch.int32), n_fwd=(2 * idx).to(torch.int32))
    pad = max(b - a for a, b in (shard_bounds(n_items, world, k) for k in range(world)))
    g = gather_results(r, pad_to=pad)
    rows = []
    for k in range(world):
        a, b = shard_bounds(n_items, world, k)
        rows.append(torch.arange(k * pad, k * pad + (b - a)))
    rows = torch.cat(rows)
    ok = (torch.equal(g["J"][rows], torch.arange(n_items, dtype=torch.float64) * 10)
          and torch.equal(g["X"][rows][:, 0, 0], torch.arange(n_items, dtype=torch.float64))
          and torch.equal(g["n_fwd"][rows], (2 * torch.arange(n_items)).to(torch.int32)))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [10, 7])
def test_all_gather_of_sharded_results(n_items):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def _ragged_worker(rank, world, port, q):
    """gather_results with ragged counts and pad_to=None: the counts are exchanged, every block padded to the largest."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, n, m = 3, 4, 2
    n_local = [5, 2, 0][rank]
    base = [0, 5, 7][rank]
    idx = torch.arange(base, base + n_local, dtype=torch.float64)
    r = dict(X=idx[:, None, None] + torch.zeros(n_local, T + 1, n, dtype=torch.float64),
             U=torch.zeros(n_local, T, m, dtype=torch.float64), J=idx * 3,
             status=torch.ones(n_local, dtype=torch.int32), n_bwd=idx.to(torch.int32), n_fwd=idx.to(torch.int32))
    g = gather_results(r)                               # pad_to=None
    c = gather_results(r, compact=True)
    ok = (g["counts"] == [5, 2, 0] and g["rows_per_rank"] == 5 and g["J"].shape[0] == 15
          and torch.equal(c["J"], torch.arange(7, dtype=torch.float64) * 3) and torch.equal(c["X"][:, 0, 0], torch.arange(7, dtype=torch.float64)))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_shards_without_pad_to_exchange_their_counts():
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True), (2, True)]


def _rows_worker(rank, world, port, S, k, q):
    """The scenario path's collective on CPU: every rank owns the rows of the (scenario, agent) pairs whose sub-problem fell
    into its share of the size buckets (here: a fixed pseudo-random assignment), tags them with their index, pads with -1
    rows to the largest rank's count -- known to every rank without an exchange -- and all-gathers; scattering the gathered
    rows by their tag must rebuild the full array on every rank."""
    from dpilqr_amd.sharding import gather_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = 1 + 6
    owner = np.random.default_rng(5).integers(0, world, size=S * k)          # the same on every rank
    counts = [int((owner == r).sum()) for r in range(world)]
    pad = max(counts + [1])
    mine = np.nonzero(owner == rank)[0]
    rows = torch.full((pad, L), -1.0, dtype=torch.float64)
    rows[:len(mine), 0] = torch.as_tensor(mine, dtype=torch.float64)
    rows[:len(mine), 1:] = torch.as_tensor(mine, dtype=torch.float64)[:, None] * 10 + torch.arange(6, dtype=torch.float64)
    out = gather_rows(rows, pad, None).numpy()
    full = np.full((S * k, 6), np.nan)
    seen = np.zeros(S * k, dtype=int)
    for row in out:
        if row[0] >= 0:
            e = int(row[0]); full[e] = row[1:]; seen[e] += 1
    ok = (seen == 1).all() and np.array_equal(full, np.arange(S * k)[:, None] * 10.0 + np.arange(6)[None, :])
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("S,k,world", [(1, 3, 2), (4, 10, 3)])
def test_tagged_rows_gathered_and_scattered(S, k, world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, S, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]


def _bench_path_worker(rank, world, port, q):
    """bench.py's N > 1 path: equal shards, gather_results with pad_to = the local count (no count exchange)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, T, n, m = 6, 2, 4, 2
    idx = torch.arange(rank * B, (rank + 1) * B, dtype=torch.float64)
    r = dict(X=idx[:, None, None] + torch.zeros(B, T + 1, n, dtype=torch.float64), U=torch.zeros(B, T, m, dtype=torch.float64),
             J=idx, status=torch.ones(B, dtype=torch.int32), n_bwd=idx.to(torch.int32), n_fwd=idx.to(torch.int32))
    g = gather_results(r, pad_to=B)
    q.put((rank, bool(torch.equal(g["J"], torch.arange(world * B, dtype=torch.float64)) and g["counts"] is None)))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_gather_path_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_path_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def _chunked_worker(rank, world, port, B, chunk, q):
    """ResultBuffers: the chunked, overlapped form of the one collective.  Each rank 'solves' by filling its send side item
    by item and reporting a growing finished prefix at its own pace (different per rank), exactly as the solve's progress
    callback does; two jobs in a row reuse the buffers."""
    from dpilqr_amd.sharding import ResultBuffers
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, n, m = 4, 6, 3
    rb = ResultBuffers(B, T, n, m, chunk=chunk)
    rb.warm()
    ok = True
    for job in range(2):
        rb.begin()
        base = 1000.0 * job + 100.0 * rank
        steps = [B // 3, B // 3, B - 1, B] if rank % 2 == 0 else [1, B // 2, B]      # ranks report at different paces
        done = 0
        for upto in steps:
            for i in range(done, upto):
                rb.out["X"][i] = base + i; rb.out["U"][i] = -(base + i)
                rb.out["status"][i] = 1 + (i % 3); rb.out["n_bwd"][i] = i; rb.out["n_fwd"][i] = 2 * i
            done = max(done, upto)
            if upto < B:
                rb.progress(done, B)
        rb.out["J"][:] = torch.arange(B, dtype=torch.float64) + base      # J is written when the solve ends
        rb.finish()
        g = rb.results()
        for r in range(world):
            b = 1000.0 * job + 100.0 * r
            idx = torch.arange(B, dtype=torch.float64)
            ok &= torch.equal(g["X"][r, :, 0, 0], idx + b) and torch.equal(g["X"][r, :, T, n - 1], idx + b)
            ok &= torch.equal(g["U"][r, :, T - 1, m - 1], -(idx + b)) and torch.equal(g["J"][r], idx + b)
            ok &= torch.equal(g["n_fwd"][r], (2 * torch.arange(B)).to(torch.int32))
            ok &= torch.equal(g["status"][r], (1 + torch.arange(B) % 3).to(torch.int32))
        ok &= tuple(g["X"].shape) == (world, B, T + 1, n)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B,chunk", [(2, 12, 4), (3, 10, 4), (2, 7, None), (3, 5, 8)])
def test_chunked_overlapped_gather(world, B, chunk):
    """world 2 and 3; B a multiple of the chunk, ragged last chunk, one chunk (chunk=None), chunk larger than the batch."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_chunked_worker, args=(r, world, port, B, chunk, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]


def _world8_worker(rank, world, port, n_chunks, chunk, q):
    """bench.py's N = 8 shape on CPU: cfg2's real CHUNK COUNT (100 steps x 1024 items in chunks of 2048 = 50 collectives of X and
    50 of U per job; items shrunk to a few doubles so that eight ranks' receive buffers fit a test), every rank reporting its
    finished prefix at its own seeded-random pace -- ranks are whole chunks apart most of the time, as eight GPUs working
    through different seeds are -- with the timeline switched on as bench.py has it."""
    from dpilqr_amd.sharding import ResultBuffers
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, n, m = 1, 2, 1
    B = n_chunks * chunk - 3                       # a ragged last chunk
    rb = ResultBuffers(B, T, n, m, chunk=chunk, timeline=True)
    rb.warm()
    rng = np.random.default_rng(100 + rank)
    ok = True
    for job in range(2):
        rb.begin()
        base = 1e6 * job + 1e4 * rank
        done = 0
        while done < B:
            upto = min(B, done + int(rng.integers(1, 5 * chunk if rank % 3 else chunk // 2 + 2)))      # some ranks crawl, some leap
            idx = torch.arange(done, upto, dtype=torch.float64)
            rb.out["X"][done:upto] = (base + idx)[:, None, None]; rb.out["U"][done:upto] = -(base + idx)[:, None, None]
            rb.out["status"][done:upto] = 1; rb.out["n_bwd"][done:upto] = idx.to(torch.int32); rb.out["n_fwd"][done:upto] = 1
            done = upto
            if done < B and rng.random() < 0.7:     # (the solver does not report after every iteration either)
                rb.progress(done, B)
        rb.out["J"][:] = torch.arange(B, dtype=torch.float64) + base
        t_end = __import__("time").perf_counter()
        rb.finish()
        tl = rb.timeline_relative_to(t_end)
        ok &= [c for c, _, _ in tl] == list(range(n_chunks))                  # every chunk exactly once, in index order, on every rank
        ok &= all(ms is None for _, _, ms in tl) and tl[-1][1] >= 0.0         # (CPU: no stream events; the last chunk can only follow the end)
        g = rb.results()
        idx = torch.arange(B, dtype=torch.float64)
        for r in range(world):
            b = 1e6 * job + 1e4 * r
            ok &= torch.equal(g["X"][r, :, 0, 0], idx + b) and torch.equal(g["X"][r, :, T, n - 1], idx + b)
            ok &= torch.equal(g["U"][r, :, 0, 0], -(idx + b)) and torch.equal(g["J"][r], idx + b)
            ok &= torch.equal(g["n_bwd"][r], torch.arange(B).to(torch.int32))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_chunked_gather_world_8_with_cfg2s_chunk_count_and_ragged_progress():
    """Eight ranks (the node the SCALE run uses), 50 chunks per job, two jobs on the same buffers: the collectives match up by
    order however far apart the ranks' progress reports are; every rank ends with every rank's results."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, 50, 8, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == [(r, True) for r in range(world)]
