"""Multi-GPU: shard the batch of sub-problems over ranks, gather the converged trajectories.

The reference's only parallel construct is multiprocessing.Pool.imap_unordered over sub-problems
(distributed.py:80-97).  Sub-problems are independent once the graph is fixed, so here each rank (one
process per GPU) solves its own slice with no data-path communication, and the path has exactly ONE
collective: an all-gather of the converged (X, U, J, status, n_bwd, n_fwd) -- RCCL over xGMI with the
`nccl` backend, gloo in the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_items, world, rank, cost=None):
    """Contiguous [lo, hi) slice of `n_items` work items for `rank`.

    cost (optional, per item, e.g. T*n_x^3 for ragged cluster sizes) balances the summed cost instead
    of the item count; items stay contiguous so that a size bucket remains one launch per rank."""
    if cost is None:
        base, extra = divmod(n_items, world)
        lo = rank * base + min(rank, extra)
        return lo, lo + base + (1 if rank < extra else 0)
    c = np.cumsum(np.asarray(cost, dtype=np.float64))
    total = c[-1] if n_items else 0.0
    cuts = [0] + [int(np.searchsorted(c, total * (r + 1) / world, side="left")) + 1 for r in range(world - 1)] + [n_items]
    cuts = np.clip(np.array(cuts), 0, n_items)
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return int(cuts[rank]), int(cuts[rank + 1])


def pack_results(r):
    """(X,U,J,status,n_bwd,n_fwd) -> one fp64 row per item (the integers are exact in fp64)."""
    B = r["J"].shape[0]
    cols = [r["X"].reshape(B, -1), r["U"].reshape(B, -1), r["J"].reshape(B, 1),
            r["status"].to(torch.float64).reshape(B, 1), r["n_bwd"].to(torch.float64).reshape(B, 1),
            r["n_fwd"].to(torch.float64).reshape(B, 1)]
    return torch.cat(cols, dim=1).contiguous()


def unpack_results(rows, x_shape, u_shape):
    B = rows.shape[0]
    nx = int(np.prod(x_shape)); nu = int(np.prod(u_shape))
    o = 0
    X = rows[:, o:o + nx].reshape((B,) + tuple(x_shape)); o += nx
    U = rows[:, o:o + nu].reshape((B,) + tuple(u_shape)); o += nu
    J = rows[:, o]; o += 1
    status, n_bwd, n_fwd = (rows[:, o + i].to(torch.int32) for i in range(3))
    return dict(X=X, U=U, J=J, status=status, n_bwd=n_bwd, n_fwd=n_fwd)


def gather_results(r, group=None, pad_to=None):
    """The path's single collective: every rank ends up with the results of all ranks, rank-major.

    Ranks may hold different item counts (ragged shards): rows are padded to `pad_to` (default: the
    max count, found with one tiny all-gather of sizes only when counts can differ)."""
    world = dist.get_world_size(group)
    rows = pack_results(r)
    n_local = rows.shape[0]
    if pad_to is None:
        pad_to = n_local
    if pad_to > n_local:
        rows = torch.cat([rows, rows.new_zeros((pad_to - n_local, rows.shape[1]))], dim=0)
    out = rows.new_empty((world * pad_to, rows.shape[1]))
    dist.all_gather_into_tensor(out, rows, group=group)
    res = unpack_results(out, r["X"].shape[1:], r["U"].shape[1:])
    res["rows_per_rank"] = pad_to
    return res


def solve_scenarios_sharded(problem, X, U, radius, xf=None, group=None, solver=None, device=None, **kwargs):
    """cfg4's shape of run: S Monte-Carlo scenarios of one k-agent problem sharded over the ranks (one process per GPU),
    each rank running the many-scenario front end (dispatch.solve_scenarios_distributed) on its contiguous slice, then
    the path's one collective: an all-gather of the stitched (X_dec, U_dec, J_full) rows.  Every rank returns the
    full arrays in scenario order.

    solver(problem, X, U, radius, xf=..., **kwargs) -> (X_dec, U_dec, J, info): injectable for CPU tests of the
    sharding / gather plumbing; device: where the gathered rows live (default: cuda if the backend is nccl)."""
    if solver is None:
        from .dispatch import solve_scenarios_distributed as solver
    X = np.asarray(X, dtype=np.float64); U = np.asarray(U, dtype=np.float64)
    S = X.shape[0]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(S, world, rank)
    xf_l = None if xf is None else np.asarray(xf, dtype=np.float64)[lo:hi]
    if hi > lo:
        Xd, Ud, J, info = solver(problem, X[lo:hi], U[lo:hi], radius, xf=xf_l, **kwargs)
    else:   # more ranks than scenarios: this rank only takes part in the collective
        T, n_u = U.shape[1], U.shape[2]
        Xd, Ud, J, info = np.zeros((0, T + 1, X.shape[2])), np.zeros((0, T, n_u)), np.zeros((0,)), {}
    pad = max(b - a for a, b in (shard_bounds(S, world, r) for r in range(world)))
    if device is None:
        device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    nx, nu = int(np.prod(Xd.shape[1:])), int(np.prod(Ud.shape[1:]))
    rows = torch.cat([torch.as_tensor(Xd).reshape(hi - lo, nx), torch.as_tensor(Ud).reshape(hi - lo, nu),
                      torch.as_tensor(J).reshape(hi - lo, 1)], dim=1)
    rows = torch.cat([rows, rows.new_zeros((pad - (hi - lo), rows.shape[1]))], dim=0).to(device).contiguous()
    out = rows.new_empty((world * pad, rows.shape[1]))
    dist.all_gather_into_tensor(out, rows, group=group)
    keep = torch.cat([torch.arange(r * pad, r * pad + (b - a)) for r, (a, b) in
                      enumerate(shard_bounds(S, world, r) for r in range(world))]).to(out.device)
    out = out[keep].cpu().numpy()
    return (out[:, :nx].reshape((S,) + Xd.shape[1:]), out[:, nx:nx + nu].reshape((S,) + Ud.shape[1:]), out[:, nx + nu],
            dict(local=info, shard=(lo, hi)))
