"""Batched dispatch: many (sub)problems -> a few device launches.

This is what replaces the reference's per-agent loop / multiprocessing pool (distributed.py:55-97): all
sub-problems of all scenarios are collected, identical ones are solved once (the reference solves one
sub-problem per AGENT, so agents with the same neighbourhood repeat the same solve: quirk Q11), the rest are
grouped into shape buckets (k agents, model family, horizon) and every bucket is ONE windowed device solve.
"""
from collections import defaultdict
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib
from .control import ilqrSolver
from .device import empty, ptr, stream_handle, to_dev
from .batch import ProblemBatch
from .lowering import describe, is_lowerable, lower_problems


# One pool of worker threads for the bucket solves of every call: the threads (and with them the per-thread solver objects
# of _lib.solver(): a pinned mailbox and a few events each) persist instead of being re-created per call and per
# receding-horizon round.
_bucket_pool = ThreadPoolExecutor(max_workers=8, thread_name_prefix="dpilqr-bucket")


SOLVE_KWARGS = ("n_lqr_iter", "tol", "t_kill")       # ilqrSolver.solve's keyword arguments that reach the device solve


def solve_kwargs(kwargs, where="solve"):
    """The keyword arguments of ilqrSolver.solve (control.py:150) among `kwargs`; anything else is named in a warning
    instead of being dropped silently (`verbose` is the reference's print switch and means nothing to a batched solve)."""
    unknown = sorted(k for k in kwargs if k not in SOLVE_KWARGS and k != "verbose")
    if unknown:
        import warnings
        warnings.warn(f"{where}: keyword arguments {unknown} are not arguments of the batched solve and are ignored", stacklevel=3)
    return {k: v for k, v in kwargs.items() if k in SOLVE_KWARGS}


def solve_problem_list(problems, x0s, U0s, keys=None, window=None, **kwargs):
    """Solve problems[i] from (x0s[i], U0s[i]); returns [(X, U, J, info)] in input order.

    keys[i] (hashable, optional): problems with equal keys are declared identical and solved once.
    kwargs: n_lqr_iter, tol, t_kill (every item's own solve-time limit, decided on the device: control.py:213-218)."""
    n = len(problems)
    out = [None] * n
    first_of = {}
    todo = []
    for i in range(n):
        key = keys[i] if keys is not None else None
        if key is not None and key in first_of:
            continue
        if key is not None:
            first_of[key] = i
        todo.append(i)
    buckets = defaultdict(list)
    for i in todo:
        p = problems[i]
        T = np.asarray(U0s[i]).shape[0]
        if is_lowerable(p):
            d = describe(p)
            buckets[(d["k"], p.dynamics.n_x // d["k"], T, d["dt"], d["w_ref"], d["w_prox"])].append(i)
        else:  # host plugins: the solver's own loop, sweep on the GPU
            X, U, J = ilqrSolver(p, T).solve(np.asarray(x0s[i]), np.asarray(U0s[i]), verbose=False, **kwargs)
            out[i] = (X, U, J, dict(status=None))
    solve_kw = solve_kwargs(kwargs)
    for shape, idx in buckets.items():
        T = shape[2]
        pb = lower_problems([problems[i] for i in idx], T)
        x0 = np.stack([np.asarray(x0s[i], dtype=np.float64).reshape(-1) for i in idx])
        U0 = np.stack([np.asarray(U0s[i], dtype=np.float64) for i in idx])
        r = pb.solve(x0, U0, window=window, **solve_kw)
        X, U, J = r["X"].cpu().numpy(), r["U"].cpu().numpy(), r["J"].cpu().numpy()
        st, nb, nf = r["status"].cpu().numpy(), r["n_bwd"].cpu().numpy(), r["n_fwd"].cpu().numpy()
        for j, i in enumerate(idx):
            out[i] = (X[j], U[j], float(J[j]), dict(status=int(st[j]), n_bwd=int(nb[j]), n_fwd=int(nf[j])))
    if keys is not None:
        for i in range(n):
            if out[i] is None:
                out[i] = out[first_of[keys[i]]]
    return out


def device_constants(d, kc=None):
    """The per-agent constants of a described problem (lowering.describe) as device tensors, for its first kc agents (all of
    them by default), and their descriptor hints -- made once per description and kept in it: the receding-horizon loop builds a
    batch per round and per cluster size, and six small uploads per batch were a tenth of the Monte-Carlo study's time."""
    kc = d["k"] if kc is None else int(kc)
    cache = d.setdefault("_dev", {})
    kc = (torch.cuda.current_device(), kc)      # (the tensors live on ONE device: a process that moves to another gets its own copies)
    if kc not in cache:
        n_ = kc[1]
        cache[kc] = dict(model=to_dev(d["model"][:n_], torch.int32), n_dims=to_dev(d["n_dims"][:n_], torch.int32),
                         Q=to_dev(d["Q"][:n_]), R=to_dev(d["R"][:n_]), Qf=to_dev(d["Qf"][:n_]),
                         word=ProblemBatch.hint_word(d["model"][:n_], d["n_dims"][:n_], d["Q"][:n_], d["R"][:n_], d["Qf"][:n_]))
    return cache[kc]


def pairwise_graph(X, radius, k, n_s):
    """define_inter_graph_threshold (distributed.py:224-247) for S scenarios on the device.
    X: (S, N, k*n_s) sampled trajectories (N may be 1); returns adjacency (S, k, k) int32 incl. self loops."""
    X = to_dev(np.asarray(X, dtype=np.float64)) if not isinstance(X, torch.Tensor) else X.contiguous()
    S, N = X.shape[0], X.shape[1]
    rad = to_dev(np.broadcast_to(np.asarray(radius, dtype=np.float64), (S,)))
    adj = empty((S, k, k), torch.int32)
    _lib.check(_lib.load().dpilqr_pairwise_graph(S, N, k, n_s, ptr(X), ptr(rad), ptr(adj), stream_handle()))
    return adj


class ScenarioFrontEnd:
    """Device-side front and back end of solve_distributed for S scenarios of one k-agent problem (include/dpilqr_hip.h
    section 7, csrc/frontend.hpp): interaction graphs as bit masks, one representative per distinct neighbourhood (the
    reference solves one sub-problem per agent: quirk Q11), a stable sort of the representatives by cluster size, the
    gathered inputs of every size's sub-problems, and the stitching of the owners' columns.  Only the k + 1 bucket counts
    come back to the host (they size the launches)."""

    def __init__(self, d, X, U, radius, xf, ignore=None):
        self.lib = _lib.load()
        self.d = d
        k = self.k = d["k"]
        if k > _lib.MAX_AGENTS:
            raise ValueError(f"at most {_lib.MAX_AGENTS} agents per problem (one bit per agent)")
        self.X = to_dev(X).contiguous(); self.U = to_dev(U).contiguous()
        self.S, self.N = int(self.X.shape[0]), int(self.X.shape[1])
        self.T = int(self.U.shape[1])
        self.n_s, self.n_c = int(self.X.shape[2]) // k, int(self.U.shape[2]) // k
        S = self.S
        self.xf = to_dev(np.broadcast_to(d["xf"], (S, k * self.n_s)).copy() if xf is None else xf).reshape(S, k * self.n_s).contiguous()
        rad = to_dev(np.broadcast_to(np.asarray(radius, dtype=np.float64), (S,)).copy())
        ign = None
        if ignore is not None and any(ignore):
            ign = to_dev(np.asarray(ignore, dtype=np.int32), torch.int32)
        n = max(S * k, 1)
        self.bits = empty((n,), torch.int64)
        self.rep, self.size, self.order, self.slot = (empty((n,), torch.int32) for _ in range(4))
        self.bstart, self.bcount = empty((k + 1,), torch.int32), empty((k + 1,), torch.int32)
        _lib.check(self.lib.dpilqr_dispatch_graph(S, self.N, k, self.n_s, ptr(self.X), ptr(rad), 1, ptr(ign), ptr(self.bits),
                                                  ptr(self.rep), ptr(self.size), ptr(self.order), ptr(self.slot),
                                                  ptr(self.bstart), ptr(self.bcount), stream_handle()))
        self.counts = self.bcount.cpu().numpy()           # the one host read of the front end
        self.starts = self.bstart.cpu().numpy()
        # per-agent parameters: one copy for every sub-problem when all agents are alike, gathered per item otherwise
        self.uniform_agents = all(bool((np.asarray(d[key]) == np.asarray(d[key])[0]).all()) for key in ("model", "n_dims", "Q", "R", "Qf"))
        if not self.uniform_agents:
            self.dev_params = {key: to_dev(np.asarray(d[key]), torch.int32 if key in ("model", "n_dims") else torch.float64)
                               for key in ("model", "n_dims", "Q", "R", "Qf")}

    def sizes(self):
        return [int(c) for c in range(1, self.k + 1) if self.counts[c] > 0]

    def bucket(self, kc, lo=0, hi=None):
        """ProblemBatch + (x0, U0) device tensors of the sub-problems [lo, hi) of the size-kc bucket."""
        d, k, ns, nc, T = self.d, self.k, self.n_s, self.n_c, self.T
        hi = int(self.counts[kc]) if hi is None else hi
        cnt = hi - lo
        x0 = empty((cnt, kc * ns)); xfb = empty((cnt, kc * ns)); U0 = empty((cnt, T, kc * nc))
        members = None if self.uniform_agents else empty((cnt, kc), torch.int32)
        _lib.check(self.lib.dpilqr_dispatch_gather(k, ns, nc, T, self.N, kc, ptr(self.order), int(self.starts[kc]) + lo, cnt,
                                                   ptr(self.bits), ptr(self.X), ptr(self.U), ptr(self.xf), k * ns, ptr(x0),
                                                   ptr(xfb), ptr(U0), ptr(members), stream_handle()))
        if self.uniform_agents:
            c = device_constants(d, kc)
            pb = ProblemBatch(c["model"], c["n_dims"], xfb, c["Q"], c["R"], c["Qf"], d["radius"], d["dt"], T, w_ref=d["w_ref"],
                              w_prox=d["w_prox"], B=cnt, hints=(kc, ns, nc, c["word"]))
        else:
            g = {}
            for key, width, eb, dt_ in (("model", 1, 4, torch.int32), ("n_dims", 1, 4, torch.int32), ("Q", ns * ns, 8, torch.float64),
                                        ("R", nc * nc, 8, torch.float64), ("Qf", ns * ns, 8, torch.float64)):
                out = empty((cnt, kc, width), dt_)
                _lib.check(self.lib.dpilqr_dispatch_gather_params(cnt, kc, width, eb, ptr(members), ptr(self.dev_params[key]),
                                                                  ptr(out), stream_handle()))
                g[key] = out
            # hints that hold for every subset of the agents: one n_dims for all of them; DoubleInt4D / Unicycle4D agents only
            word = 0
            nd_all = np.asarray(d["n_dims"], dtype=np.int32)
            if bool((nd_all == nd_all[0]).all()):
                word |= (1 + int(nd_all[0])) << 8
            if bool(np.isin(np.asarray(d["model"]), (0, 3)).all()):
                word |= 1 << 17
            pb = ProblemBatch(g["model"], g["n_dims"], xfb, g["Q"], g["R"], g["Qf"], d["radius"], d["dt"], T, w_ref=d["w_ref"],
                              w_prox=d["w_prox"], B=cnt, hints=(kc, ns, nc, word))
        return pb, x0, U0

    def results_struct(self, solved):
        """solved: {kc: (X, U, first, count)} device tensors of the slices solved here -> dpilqr_bucket_results."""
        R = _lib.BucketResults()
        for kc, (Xs, Us, first, count) in solved.items():
            R.X[kc], R.U[kc], R.first[kc], R.count[kc] = Xs.data_ptr(), Us.data_ptr(), int(first), int(count)
        self._keep = solved          # the struct holds raw pointers: keep the tensors alive
        return R

    def stitch(self, solved):
        S, k, ns, nc, T = self.S, self.k, self.n_s, self.n_c, self.T
        X_dec = torch.zeros((S, T + 1, k * ns), dtype=torch.float64, device=self.X.device)
        U_dec = torch.zeros((S, T, k * nc), dtype=torch.float64, device=self.X.device)
        R = self.results_struct(solved)
        import ctypes as C
        _lib.check(self.lib.dpilqr_dispatch_stitch(S, k, ns, nc, T, ptr(self.bits), ptr(self.rep), ptr(self.size), ptr(self.slot),
                                                   C.addressof(R), ptr(X_dec), ptr(U_dec), stream_handle()))
        return X_dec, U_dec

    def pack_rows(self, solved, count_only=False, pad_to=None):
        """One row [s*k+i | X columns | U columns] per (scenario, agent) whose sub-problem is in `solved`; returns
        (rows or None, n_rows).  Rows beyond n_rows (up to pad_to) carry index -1."""
        import ctypes as C
        S, k, ns, nc, T = self.S, self.k, self.n_s, self.n_c, self.T
        R = self.results_struct(solved)
        row_of = empty((max(S * k, 1),), torch.int32); n_rows = empty((1,), torch.int32)
        row_len = 1 + (T + 1) * ns + T * nc
        rows = None
        if not count_only:
            rows = torch.zeros((int(pad_to), row_len), dtype=torch.float64, device=self.X.device)
            rows[:, 0] = -1.0
        _lib.check(self.lib.dpilqr_dispatch_pack_rows(S, k, ns, nc, T, ptr(self.bits), ptr(self.rep), ptr(self.size), ptr(self.slot),
                                                      C.addressof(R), ptr(row_of), ptr(n_rows), ptr(rows), row_len, stream_handle()))
        return rows, int(n_rows.item())

    def scatter_rows(self, rows):
        S, k, ns, nc, T = self.S, self.k, self.n_s, self.n_c, self.T
        X_dec = torch.zeros((S, T + 1, k * ns), dtype=torch.float64, device=rows.device)
        U_dec = torch.zeros((S, T, k * nc), dtype=torch.float64, device=rows.device)
        _lib.check(self.lib.dpilqr_dispatch_scatter_rows(int(rows.shape[0]), k, ns, nc, T, ptr(rows), int(rows.shape[1]), ptr(X_dec),
                                                         ptr(U_dec), stream_handle()))
        return X_dec, U_dec


def solve_scenarios_distributed(problem, X, U, radius, xf=None, window=None, concurrent=True, ignore_ids=None, device_out=False,
                                shard=None, audit=False, desc=None, **kwargs):
    """solve_distributed (distributed.py:25-103) for S scenarios of ONE k-agent problem at once -- the Monte-Carlo
    front end (scripts/analysis.py:126-174 runs it seed by seed).  Everything between the trajectories and the stitched
    result stays on the device: graph, de-duplication, size buckets, gathered sub-problem inputs (ScenarioFrontEnd), one
    windowed solve per cluster size, stitching, and the full-problem rollout for J_full.

    problem : lowerable k-agent ilqrProblem (models, Q/R/Qf per agent, n_dims, the proximity cost's own radius)
    X (S, T+1, n_x) or (S, 1, n_x) [the reference's `X = x0` first call]; U (S, T, n_u); radius: graph threshold/2
    xf (S, n_x): per-scenario goals (default: the problem's own for every scenario); NumPy arrays or device tensors
    shard = (rank, world): solve only this rank's share of every size bucket (sharding.solve_scenarios_sharded)
    audit : keep every bucket's sub-problem inputs and full solve record (x0, xf, U0, X, U, J, status, n_bwd, n_fwd and the
            decision trace) as host arrays in info["audit"][cluster size] -- what a parity check of the individual
            sub-problem solves needs; the solves themselves are the same
    returns X_dec (S, T+1, n_x), U_dec (S, T, n_u), J_full (S,), info (clusters as bit masks, counts) -- NumPy arrays, or
    device tensors with device_out=True; with `shard` the unstitched (front end, solved slices) pair instead.
    """
    from .sharding import shard_bounds
    d = desc if desc is not None else describe(problem)      # desc: the caller's description (its device constants are reused)
    k = d["k"]
    ignore = None
    if ignore_ids:
        ids = list(problem.ids)
        ignore = [1 if id_ in ignore_ids else 0 for id_ in ids]
    solve_kw = solve_kwargs(kwargs, "solve_scenarios_distributed")
    from time import perf_counter as pc
    t0 = pc()
    fe = ScenarioFrontEnd(d, X, U, radius, xf, ignore)
    t_front = pc() - t0                      # uploads, graph, de-duplication, bucket sort, the one host read
    S, T = fe.S, fe.T
    sizes = fe.sizes()
    dev_index = torch.cuda.current_device()

    def solve_bucket(kc):
        """The sub-problems of one cluster size (this rank's share of them): one windowed device solve on this thread's
        own stream.  The current HIP device is per host thread and a new thread starts on device 0: pin the worker to the
        caller's GPU (one process per GPU: rank r's buckets must not land on GPU 0)."""
        n_kc = int(fe.counts[kc])
        lo, hi = (0, n_kc) if shard is None else shard_bounds(n_kc, shard[1], shard[0])
        if hi <= lo:
            return kc, None
        with torch.cuda.device(dev_index), torch.cuda.stream(torch.cuda.Stream(device=dev_index)):
            pb, x0, U0 = fe.bucket(kc, lo, hi)
            r = pb.solve(x0, U0, window=window, trace=bool(audit), **solve_kw)
            torch.cuda.current_stream().synchronize()
            rec = None
            if audit:
                rec = {key: v.cpu().numpy() for key, v in r.items()}
                rec.update(x0=x0.cpu().numpy(), U0=U0.cpu().numpy(), xf=pb._xf.cpu().numpy().reshape(hi - lo, -1), lo=lo)
            return kc, (r["X"], r["U"], lo, hi - lo, int(r["n_bwd"].sum().item()), rec)

    # buckets are independent and, for a handful of scenarios, small: their solves run concurrently, each on its own
    # HIP stream from its own host thread
    torch.cuda.synchronize()
    t0 = pc()
    if concurrent and len(sizes) > 1:
        results = list(_bucket_pool.map(solve_bucket, sizes))
    else:
        results = [solve_bucket(kc) for kc in sizes]
    t_solve = pc() - t0                      # gathers + the windowed device solves of all cluster sizes
    solved = {kc: r[:4] for kc, r in results if r is not None}
    n_bwd_total = sum(r[4] for _, r in results if r is not None)
    info = dict(n_subproblems=int(S * k), n_unique=int(fe.counts.sum()),
                sizes={int(kc): int(fe.counts[kc]) for kc in sizes}, n_bwd=n_bwd_total,
                seconds=dict(front_end=t_front, solves=t_solve))
    if audit:
        info["audit"] = {int(kc): r[5] for kc, r in results if r is not None}
    if shard is not None:
        return fe, solved, info
    # stitch: agent i of scenario s takes ITS columns of the sub-problem solved for its neighbourhood; then J_full:
    # the cost of the stitched controls rolled out on the full problem (distributed.py:100-101)
    t0 = pc()
    X_dec, U_dec = fe.stitch(solved)
    J = full_rollout_cost(d, fe, U_dec)
    torch.cuda.synchronize()
    info["seconds"]["stitch_and_rollout"] = pc() - t0
    info["cluster_bits"] = fe.bits.cpu().numpy().reshape(S, k)
    if device_out:
        return X_dec, U_dec, J, info
    return X_dec.cpu().numpy(), U_dec.cpu().numpy(), J.cpu().numpy(), info


def full_rollout_cost(d, fe, U_dec):
    """J_full of every scenario: _rollout(X[0], U_dec) on the full k-agent problem (distributed.py:100-101)."""
    c = device_constants(d)
    full = ProblemBatch(c["model"], c["n_dims"], fe.xf, c["Q"], c["R"], c["Qf"], d["radius"], d["dt"], fe.T,
                        w_ref=d["w_ref"], w_prox=d["w_prox"], B=fe.S, hints=(d["k"], fe.n_s, fe.n_c, c["word"]))
    _, J = full.rollout(fe.X[:, 0].contiguous(), U_dec)
    return J
