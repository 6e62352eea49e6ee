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
    cuts = np.minimum.accumulate(np.array(cuts[::-1]))[::-1] if False else np.array(cuts)
    cuts = np.clip(cuts, 0, n_items)
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
