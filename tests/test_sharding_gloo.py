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


def _fake_front_end(problem, X, U, radius, xf=None, **kwargs):
    """Stands in for dispatch.solve_scenarios_distributed on CPU: results that encode their inputs."""
    S, T = U.shape[0], U.shape[1]
    Xd = np.repeat(X[:, :1, :], T + 1, axis=1) + np.arange(T + 1)[None, :, None]
    return Xd, U * 2.0, X[:, 0, 0] * 10.0 + (0.0 if xf is None else xf[:, 0]), dict(n=S)


def _scenario_worker(rank, world, port, S, q):
    from dpilqr_amd.sharding import solve_scenarios_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(0)                                # the same scenarios on every rank
    X = rng.normal(size=(S, 1, 6)); U = rng.normal(size=(S, 4, 3)); xf = rng.normal(size=(S, 6))
    Xd, Ud, J, info = solve_scenarios_sharded(None, X, U, 0.5, xf=xf, solver=_fake_front_end)
    Xe, Ue, Je, _ = _fake_front_end(None, X, U, 0.5, xf=xf)
    ok = np.array_equal(Xd, Xe) and np.array_equal(Ud, Ue) and np.array_equal(J, Je) and info["shard"] == shard_bounds(S, world, rank)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("S", [9, 1])
def test_scenarios_sharded_over_ranks_and_gathered(S):
    """Monte-Carlo scenarios sharded over two ranks (ragged: 5 + 4, and 1 + 0), one all-gather, scenario order kept."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_scenario_worker, args=(r, world, port, S, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]
