"""Multi-GPU: shard the batch of sub-problems over ranks, gather the converged trajectories.

The reference's only parallel construct is multiprocessing.Pool.imap_unordered over sub-problems
(distributed.py:80-97).  Sub-problems are independent once the graph is fixed, so here each rank (one
process per GPU) solves its own slice with no data-path communication, and the path has exactly ONE
collective: an all-gather of the converged (X, U, J, status, n_bwd, n_fwd) -- RCCL over xGMI with the
`nccl` backend, gloo in the CPU tests.
"""
from time import perf_counter as _now

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
    nx = int(np.prod(r["X"].shape[1:])); nu = int(np.prod(r["U"].shape[1:]))      # explicit: an empty shard has no -1 to infer
    cols = [r["X"].reshape(B, nx), r["U"].reshape(B, nu), r["J"].reshape(B, 1),
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


def gather_counts(n_local, group=None, device=None):
    """Every rank's item count (one tiny all-gather of int64)."""
    world = dist.get_world_size(group)
    mine = torch.tensor([int(n_local)], dtype=torch.int64, device=device)
    out = torch.empty((world,), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine, group=group)
    return [int(v) for v in out.cpu()]


def gather_rows(rows, pad_to, group=None):
    """The path's single data collective: rows (n_local, L) of every rank, each block padded to pad_to rows with zeros
    -> (world * pad_to, L), rank-major.  Works for RCCL (device tensors, backend nccl) and gloo (CPU tensors) alike."""
    world = dist.get_world_size(group)
    n_local = rows.shape[0]
    if pad_to < n_local:
        raise ValueError(f"pad_to={pad_to} is smaller than this rank's {n_local} rows")
    if pad_to > n_local:
        rows = torch.cat([rows, rows.new_zeros((pad_to - n_local, rows.shape[1]))], dim=0)
    out = rows.new_empty((world * pad_to, rows.shape[1]))
    dist.all_gather_into_tensor(out, rows.contiguous(), group=group)
    return out


def gather_results(r, group=None, pad_to=None, compact=False):
    """The path's single collective: every rank ends up with the results of all ranks, rank-major.

    Ranks may hold different item counts (ragged shards).  pad_to=None: the counts are exchanged first (one all-gather of
    one integer per rank) and every block is padded to the largest; pass pad_to when the caller knows the bound and wants
    to save that exchange.  The result carries `counts` (items per rank, None if they were not exchanged) and
    `rows_per_rank`; compact=True drops the padding rows (needs the counts)."""
    rows = pack_results(r)
    counts = None
    if pad_to is None:
        counts = gather_counts(rows.shape[0], group, rows.device)
        pad_to = max(counts) if counts else 0
    out = gather_rows(rows, pad_to, group)
    if compact:
        if counts is None:
            counts = gather_counts(rows.shape[0], group, rows.device)
        keep = torch.cat([torch.arange(k * pad_to, k * pad_to + c) for k, c in enumerate(counts)]).to(out.device)
        out = out[keep]
    res = unpack_results(out, r["X"].shape[1:], r["U"].shape[1:])
    res["rows_per_rank"] = pad_to
    res["counts"] = counts
    return res


class ResultBuffers:
    """Pre-allocated home of one rank's results AND of the gathered results of all ranks, for the batch-sharded form of
    the path's one collective (bench.py; any Monte-Carlo driver that repeats jobs of one shape).

      * the solve writes X, U, J, status, n_bwd, n_fwd straight into the send side (ProblemBatch.solve(out=rb.out)): no
        packing pass, no allocation inside a timed region (round 2 packed rows with torch.cat and allocated the
        world x 250 MB receive buffer per call);
      * the gather is issued in CHUNKS of `chunk` items on a side stream WHILE the solve is still running: the solve reports
        the finished prefix of its batch (dpilqr_solver_set_progress -> rb.progress) and every chunk that lies below it
        goes out at once -- X and U of the chunk, two all-gathers into chunk-major receive buffers; what is left, and the
        20 bytes of (J, status, n_bwd, n_fwd) per item, follow in finish().  It stays ONE logical all-gather of the converged
        trajectories: same payload, same ranks, no other collective;
      * warm() runs the whole sequence once on the real buffers outside any clock: pages touched, RCCL channels and
        communicators for exactly these message sizes set up.

    Every rank must create it with the same (B, chunk) and call progress/finish for every solve: the chunks are gathered in
    index order on every rank, whenever each rank is ready -- the collectives match up by order.
    Works with backend nccl (= RCCL, device tensors, side stream) and gloo (CPU tensors, no streams: the CPU tests)."""

    def __init__(self, B, T, n_x, n_u, chunk=None, group=None, device=None, dtype=torch.float64, timeline=False):
        self.group = group
        # timeline=True: per chunk, when its gather was ISSUED (host clock) and an event behind it on the side stream (when it
        # was DONE): bench.py --gpus N prints both relative to the end of the solve, so that the exposed tail of the one
        # collective is visible in the first multi-GPU line anyone runs
        self.timeline = [] if timeline else None
        self.collective = dist.is_initialized()            # no process group: one rank, the "gather" is a copy
        self.world = dist.get_world_size(group) if self.collective else 1
        self.B, self.T, self.n_x, self.n_u = int(B), int(T), int(n_x), int(n_u)
        self.chunk = int(chunk) if chunk else max(self.B, 1)
        self.n_chunks = max(1, -(-self.B // self.chunk))
        Bp = self.n_chunks * self.chunk                      # the last chunk is padded (its tail rows are never read)
        dev = torch.device("cpu") if device is None else torch.device(device)
        self.device = dev
        mk = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        self._X, self._U = mk((Bp, T + 1, n_x), dtype), mk((Bp, T, n_u), dtype)
        self._J = mk((Bp,), torch.float64)
        self._st, self._nb, self._nf = (mk((Bp,), torch.int32) for _ in range(3))
        self.out = dict(X=self._X[:self.B], U=self._U[:self.B], J=self._J[:self.B], status=self._st[:self.B],
                        n_bwd=self._nb[:self.B], n_fwd=self._nf[:self.B])
        W, C_, c = self.world, self.n_chunks, self.chunk
        self._Xall, self._Uall = mk((C_, W, c, T + 1, n_x), dtype), mk((C_, W, c, T, n_u), dtype)
        self._stats, self._stats_all = mk((Bp, 4), torch.float64), mk((W, Bp, 4), torch.float64)
        self._side = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._sent = 0

    # -- the collective, chunk by chunk
    def _gather_chunk(self, c):
        lo, hi = c * self.chunk, (c + 1) * self.chunk
        if not self.collective:
            self._Xall[c, 0].copy_(self._X[lo:hi]); self._Uall[c, 0].copy_(self._U[lo:hi])
            return
        dist.all_gather_into_tensor(self._Xall[c].view(-1), self._X[lo:hi].view(-1), group=self.group)
        dist.all_gather_into_tensor(self._Uall[c].view(-1), self._U[lo:hi].view(-1), group=self.group)

    def _on_side(self):
        import contextlib
        return torch.cuda.stream(self._side) if self._side is not None else contextlib.nullcontext()

    def progress(self, n_finished, n_items=None):
        """Items [0, n_finished) of this rank's batch are final: gather every chunk that is complete (side stream)."""
        full = self.n_chunks if n_finished >= self.B else n_finished // self.chunk
        if full > self._sent:
            with self._on_side():
                while self._sent < full:
                    self._gather_chunk(self._sent)
                    if self.timeline is not None:
                        ev = None
                        if self._side is not None:
                            ev = torch.cuda.Event(enable_timing=True); ev.record(self._side)
                        self.timeline.append((self._sent, _now(), ev))
                    self._sent += 1

    def finish(self):
        """After the solve has returned: the chunks not yet gathered, then (J, status, n_bwd, n_fwd); the caller's stream
        waits for the side stream, so the gathered results are ordered before anything enqueued next."""
        if self._side is not None:       # (the synchronous solve has already waited for its stream; an enqueue-only one has not)
            self._side.wait_stream(torch.cuda.current_stream(self.device))
        self.progress(self.B)
        with self._on_side():
            self._stats[:, 0].copy_(self._J); self._stats[:, 1].copy_(self._st); self._stats[:, 2].copy_(self._nb)
            self._stats[:, 3].copy_(self._nf)
            if not self.collective:
                self._stats_all[0].copy_(self._stats)
            else:
                dist.all_gather_into_tensor(self._stats_all.view(-1), self._stats.view(-1), group=self.group)
        if self._side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._side)
        self._sent = 0

    def begin(self):
        """Before a solve that writes into self.out: the side stream must have finished reading the previous job's results."""
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(self.device))
            torch.cuda.current_stream(self.device).wait_stream(self._side)
        self._sent = 0
        if self.timeline is not None:
            self.timeline = []

    def timeline_relative_to(self, t_host, ev_stream=None):
        """[(chunk, seconds from t_host to the chunk's ISSUE on the host, milliseconds from ev_stream to its COMPLETION on the
        side stream or None)]: negative = before the reference point (overlapped with the solve), positive = after it (exposed).
        Call after the device has been synchronised."""
        out = []
        for c, t, ev in (self.timeline or []):
            out.append((c, t - t_host, (ev_stream.elapsed_time(ev) if (ev is not None and ev_stream is not None) else None)))
        return out

    def warm(self):
        self.begin()
        self.finish()
        if self._side is not None:
            torch.cuda.synchronize(self.device)

    def results(self):
        """All ranks' results, rank-major: X (world, B, T+1, n_x), U (world, B, T, n_u), J, status, n_bwd, n_fwd (world, B).
        X and U are re-ordered from the chunk-major receive buffers on demand (a copy when there is more than one chunk)."""
        W, B = self.world, self.B
        X = self._Xall.permute(1, 0, 2, 3, 4).reshape(W, -1, self.T + 1, self.n_x)[:, :B]
        U = self._Uall.permute(1, 0, 2, 3, 4).reshape(W, -1, self.T, self.n_u)[:, :B]
        s = self._stats_all[:, :B]
        return dict(X=X, U=U, J=s[:, :, 0], status=s[:, :, 1].to(torch.int32), n_bwd=s[:, :, 2].to(torch.int32),
                    n_fwd=s[:, :, 3].to(torch.int32))


def solve_scenarios_sharded(problem, X, U, radius, xf=None, group=None, window=None, device_out=False, **kwargs):
    """cfg4's shape of run on several GPUs (one process per GPU): S Monte-Carlo scenarios of one k-agent problem.

    Partitioning as SURVEY 8(e) prescribes: every rank builds the (cheap) front end for ALL scenarios -- interaction graphs,
    de-duplicated (scenario, neighbourhood) sub-problems, sorted into size buckets -- and solves a contiguous 1 / world share
    of EVERY bucket, so that the cost (~ T n_x^3 per sub-problem, equal within a bucket) is balanced by construction
    however ragged the cluster sizes are.  Then the path's one collective: an all-gather of one row per (scenario, agent),
    [index | the agent's columns of X | of U], padded to the largest rank's row count (every rank can count every rank's
    rows from the replicated front end, so no size exchange is needed).  Every rank scatters all rows into X_dec, U_dec on
    its device and rolls the full problem out for J_full.  Results stay on the device end to end (device_out=True returns
    the tensors); nothing passes through host memory except the k + 1 bucket counts."""
    from .dispatch import full_rollout_cost, solve_scenarios_distributed
    from .lowering import describe
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    fe, solved, info = solve_scenarios_distributed(problem, X, U, radius, xf=xf, window=window, shard=(rank, world), **kwargs)
    # row counts of every rank, from the replicated front end
    n_rows = []
    for r in range(world):
        if r == rank:
            _, n = fe.pack_rows(solved, count_only=True)
        else:
            other = {kc: (None, None) + shard_slice(int(fe.counts[kc]), world, r) for kc in fe.sizes()}
            _, n = fe.pack_rows({kc: (fe.X, fe.U, lo, cnt) for kc, (_, _, lo, cnt) in other.items() if cnt > 0}, count_only=True)
        n_rows.append(n)
    pad = max(n_rows + [1])
    rows, n_mine = fe.pack_rows(solved, pad_to=pad)
    assert n_mine == n_rows[rank]
    allrows = gather_rows(rows, pad, group)
    X_dec, U_dec = fe.scatter_rows(allrows)
    J = full_rollout_cost(describe(problem), fe, U_dec)
    info = dict(info, shard_rows=n_rows, rank=rank, cluster_bits=fe.bits.cpu().numpy().reshape(fe.S, fe.k))
    if device_out:
        return X_dec, U_dec, J, info
    return X_dec.cpu().numpy(), U_dec.cpu().numpy(), J.cpu().numpy(), info


def shard_slice(n_items, world, rank):
    """(first, count) of rank's contiguous share of n_items."""
    lo, hi = shard_bounds(n_items, world, rank)
    return lo, hi - lo
