"""Batched dispatch: many (sub)problems -> a few device launches.

This is what replaces the reference's per-agent loop / multiprocessing pool (distributed.py:55-97): all
sub-problems of all scenarios are collected, identical ones are solved once (the reference solves one
sub-problem per AGENT, so agents with the same neighbourhood repeat the same solve: quirk Q11), the rest are
grouped into shape buckets (k agents, model family, horizon) and every bucket is ONE windowed device solve.
"""
from collections import defaultdict

import numpy as np
import torch

from . import _lib
from .control import ilqrSolver
from .device import empty, ptr, stream_handle, to_dev
from .batch import ProblemBatch
from .lowering import describe, is_lowerable, lower_problems


def solve_problem_list(problems, x0s, U0s, keys=None, window=None, **kwargs):
    """Solve problems[i] from (x0s[i], U0s[i]); returns [(X, U, J, info)] in input order.

    keys[i] (hashable, optional): problems with equal keys are declared identical and solved once.
    kwargs: n_lqr_iter, tol (t_kill forces the per-problem host loop, as in control.ilqrSolver)."""
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
        if is_lowerable(p) and not kwargs.get("t_kill"):
            d = describe(p)
            buckets[(d["k"], p.dynamics.n_x // d["k"], T, d["dt"], d["w_ref"], d["w_prox"])].append(i)
        else:  # host plugins: the solver's own loop, sweep on the GPU
            X, U, J = ilqrSolver(p, T).solve(np.asarray(x0s[i]), np.asarray(U0s[i]), verbose=False, **kwargs)
            out[i] = (X, U, J, dict(status=None))
    solve_kw = {k: v for k, v in kwargs.items() if k in ("n_lqr_iter", "tol")}
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


def pairwise_graph(X, radius, k, n_s):
    """define_inter_graph_threshold (distributed.py:224-247) for S scenarios on the device.
    X: (S, N, k*n_s) sampled trajectories (N may be 1); returns adjacency (S, k, k) int32 incl. self loops."""
    X = to_dev(np.asarray(X, dtype=np.float64)) if not isinstance(X, torch.Tensor) else X.contiguous()
    S, N = X.shape[0], X.shape[1]
    rad = to_dev(np.broadcast_to(np.asarray(radius, dtype=np.float64), (S,)))
    adj = empty((S, k, k), torch.int32)
    _lib.check(_lib.load().dpilqr_pairwise_graph(S, N, k, n_s, ptr(X), ptr(rad), ptr(adj), stream_handle()))
    return adj


def solve_scenarios_distributed(problem, X, U, radius, xf=None, window=None, concurrent=True, **kwargs):
    """solve_distributed (distributed.py:25-103) for S scenarios of ONE k-agent problem at once -- the Monte-Carlo
    front end (scripts/analysis.py:126-174 runs it seed by seed).  Everything between the interaction graph and
    the stitched trajectories is array code: no per-sub-problem Python objects, one windowed device solve per
    cluster size.

    problem : lowerable k-agent ilqrProblem (models, Q/R/Qf per agent, n_dims, the proximity cost's own radius)
    X (S, T+1, n_x) or (S, 1, n_x) [the reference's `X = x0` first call]; U (S, T, n_u); radius: graph threshold/2
    xf (S, n_x): per-scenario goals (default: the problem's own for every scenario)
    returns X_dec (S, T+1, n_x), U_dec (S, T, n_u), J_full (S,), info (clusters as bit masks, counts)
    """
    d = describe(problem)
    k, T = d["k"], int(np.asarray(U).shape[1])
    X = np.asarray(X, dtype=np.float64); U = np.asarray(U, dtype=np.float64)
    S = X.shape[0]
    n_s, n_c = problem.dynamics.n_x // k, problem.dynamics.n_u // k
    xf = np.broadcast_to(d["xf"], (S, k * n_s)) if xf is None else np.asarray(xf, dtype=np.float64).reshape(S, k * n_s)
    solve_kw = {key: v for key, v in kwargs.items() if key in ("n_lqr_iter", "tol")}

    # 1. interaction graphs of all scenarios (device), as one bit mask per (scenario, agent): bit j = j is a neighbour
    adj = pairwise_graph(X, radius, k, n_s).cpu().numpy().astype(bool)                   # (S, k, k), self loops set
    bits = (adj.astype(np.int64) << np.arange(k, dtype=np.int64)).sum(axis=2)           # (S, k)
    # 2. one sub-problem per DISTINCT (scenario, neighbourhood); the reference solves one per agent (quirk Q11)
    flat = np.stack([np.repeat(np.arange(S, dtype=np.int64), k), bits.reshape(-1)], axis=1)
    uniq, inverse = np.unique(flat, axis=0, return_inverse=True)
    inverse = np.asarray(inverse).reshape(S, k)
    u_s, u_bits = uniq[:, 0], uniq[:, 1]
    u_mask = ((u_bits[:, None] >> np.arange(k)) & 1).astype(bool)                         # (n_unique, k)
    u_size = u_mask.sum(axis=1)
    X_dec = np.zeros((S, T + 1, k * n_s)); U_dec = np.zeros((S, T, k * n_c))
    X0 = X[:, 0].reshape(S, k, n_s); Uk = U.reshape(S, T, k, n_c); xfk = xf.reshape(S, k, n_s)
    agent_of = np.arange(k)

    def solve_bucket(kc):
        """All distinct sub-problems with kc agents: one windowed device solve, on this thread's own stream."""
        sel = np.nonzero(u_size == kc)[0]                                                 # sub-problems of this size
        members = np.nonzero(u_mask[sel])[1].reshape(len(sel), kc)                        # sorted agent ids, (Bk, kc)
        rows = u_s[sel][:, None]
        # the current HIP device is per host thread and a new thread starts on device 0: pin this worker to the
        # caller's GPU (one process per GPU: rank r's buckets must not land on GPU 0)
        with torch.cuda.device(dev_index), torch.cuda.stream(torch.cuda.Stream(device=dev_index)):
            pb = ProblemBatch(d["model"][members], d["n_dims"][members], xfk[rows, members].reshape(len(sel), kc * n_s),
                              d["Q"][members], d["R"][members], d["Qf"][members], d["radius"], d["dt"], T,
                              w_ref=d["w_ref"], w_prox=d["w_prox"], B=len(sel))
            x0 = X0[rows, members].reshape(len(sel), kc * n_s)
            U0 = Uk[rows, :, members]                                                     # (Bk, kc, T, n_c)
            U0 = np.ascontiguousarray(np.transpose(U0, (0, 2, 1, 3))).reshape(len(sel), T, kc * n_c)
            r = pb.solve(x0, U0, window=window, **solve_kw)
            Xs = r["X"].cpu().numpy().reshape(len(sel), T + 1, kc, n_s)
            Us = r["U"].cpu().numpy().reshape(len(sel), T, kc, n_c)
            return sel, Xs, Us, int(r["n_bwd"].sum().item())

    # buckets are independent and, for a handful of scenarios, small: their solves run concurrently, each on its own
    # HIP stream from its own host thread (the library keeps its per-solve state per thread)
    sizes = [int(v) for v in np.unique(u_size)]
    dev_index = torch.cuda.current_device()
    torch.cuda.synchronize()
    if concurrent and len(sizes) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(8, len(sizes))) as pool:
            results = list(pool.map(solve_bucket, sizes))
    else:
        results = [solve_bucket(kc) for kc in sizes]
    n_bwd_total = 0
    for sel, Xs, Us, nb in results:
        n_bwd_total += nb
        # 3. stitch: agent i of scenario s takes ITS columns of the sub-problem solved for its neighbourhood
        local = np.full(len(uniq), -1, dtype=np.int64); local[sel] = np.arange(len(sel))
        owner_s, owner_i = np.nonzero(local[inverse] >= 0)
        j = local[inverse[owner_s, owner_i]]
        pos = (u_mask[sel][j] & (agent_of[None, :] < owner_i[:, None])).sum(axis=1)       # rank of i in its cluster
        X_dec.reshape(S, T + 1, k, n_s)[owner_s, :, owner_i] = Xs[j, :, pos]
        U_dec.reshape(S, T, k, n_c)[owner_s, :, owner_i] = Us[j, :, pos]
    # 4. J_full: cost of the stitched controls rolled out on the full problem (distributed.py:100-101)
    full = ProblemBatch(d["model"], d["n_dims"], xf, d["Q"], d["R"], d["Qf"], d["radius"], d["dt"], T,
                        w_ref=d["w_ref"], w_prox=d["w_prox"], B=S)
    _, J = full.rollout(X[:, 0], U_dec)
    info = dict(cluster_bits=bits, n_subproblems=int(S * k), n_unique=int(len(uniq)),
                sizes={int(kc): int((u_size == kc).sum()) for kc in np.unique(u_size)}, n_bwd=n_bwd_total)
    return X_dec, U_dec, J.cpu().numpy(), info
