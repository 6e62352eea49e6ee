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
