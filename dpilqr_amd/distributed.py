"""DP-iLQR driver: same entry points as the reference's dpilqr/distributed.py
(solve_distributed :25-103, solve_rhc :106-221, define_inter_graph_threshold :224-247, solve_centralized :250-258).

The per-agent sub-problem loop (and its optional multiprocessing pool) is replaced by one batched device
dispatch (dispatch.solve_problem_list); graph construction and stitching are unchanged in meaning.
"""
import itertools
import logging
from time import perf_counter as pc

import numpy as np

from .control import ilqrSolver
from .dispatch import solve_kwargs, solve_problem_list, solve_scenarios_distributed  # noqa: F401
from .util import compute_pairwise_distance, split_graph


def define_inter_graph_threshold(X, radius, x_dims, ids):
    """Interaction graph {id: sorted ids within 2*radius (planar) at any sampled step, incl. itself}."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    n_rows = X.shape[0]
    step = max(n_rows // 10, 1)                      # ~10 samples over the trajectory
    near = compute_pairwise_distance(X, x_dims)[0:n_rows + 1:step] < 2 * radius
    graph = {id_: [id_] for id_ in ids}
    # like the reference (distributed.py:240-246, quirk Q10) the neighbours come out of a NumPy array of id pairs, i.e.
    # as NumPy integers next to the agent's own Python int: they compare and hash like ints, but they print differently
    # inside the `subgraphs` field of solve_rhc's CSV rows, which is reproduced byte for byte
    for col, (a, b) in enumerate(np.array(list(itertools.combinations(ids, 2))).reshape(-1, 2)):
        if near[:, col].any():
            graph[int(a)].append(b)
            graph[int(b)].append(a)
    return {id_: sorted(members) for id_, members in graph.items()}


def solve_distributed(problem, X, U, radius, ignore_ids=None, pool=None, verbose=True, **kwargs):
    """One sub-problem per agent (its closed neighbourhood), all solved in one batched dispatch.

    Returns X_dec (N+1, n_x), U_dec (N, n_u), J_full, solve_info {id: (seconds, neighbourhood)}.
    `pool` is accepted for signature compatibility; the device batch takes its place.  `ignore_ids=None`
    means "ignore nobody" (the reference raises TypeError there, quirk Q9)."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64)); U = np.asarray(U, dtype=np.float64)
    x_dims, u_dims = problem.game_cost.x_dims, problem.game_cost.u_dims
    N, n_s, n_c, ids = U.shape[0], x_dims[0], u_dims[0], problem.ids
    ignore_ids = list(ignore_ids) if ignore_ids else []
    if any(id_ not in ids for id_ in ignore_ids):
        raise ValueError(f"Some of {ignore_ids} not in {ids}.")
    graph = define_inter_graph_threshold(X, radius, x_dims, ids)
    if verbose:
        print("=" * 80 + f"\nInteraction Graph: {graph}")
    x0_split = split_graph(X[np.newaxis, 0], x_dims, graph)
    U_split = split_graph(U, u_dims, graph)
    subproblems = problem.split(graph)
    todo = [i for i, id_ in enumerate(ids) if id_ not in ignore_ids]
    t0 = pc()
    res = solve_problem_list([subproblems[i] for i in todo], [x0_split[i] for i in todo], [U_split[i] for i in todo],
                             keys=[tuple(graph[ids[i]]) for i in todo], **kwargs)
    dt_each = (pc() - t0) / max(len(todo), 1)
    X_dec = np.zeros((N + 1, len(ids) * n_s)); U_dec = np.zeros((N, len(ids) * n_c))
    solve_info = {}
    for i, (Xi, Ui, _, _) in zip(todo, res):
        Xa, Ua = subproblems[i].extract(Xi, Ui, ids[i])
        X_dec[:, i * n_s:(i + 1) * n_s] = Xa
        U_dec[:, i * n_c:(i + 1) * n_c] = Ua
        solve_info[ids[i]] = (dt_each, graph[ids[i]])
    for id_ in ignore_ids:
        solve_info[id_] = (0.0, [id_])
    _, J_full = ilqrSolver(problem, N)._rollout(X[0], U_dec)
    return X_dec, U_dec, J_full, solve_info


def solve_centralized(solver, xi, U, ids, verbose, **kwargs):
    t0 = pc()
    X, U, J = solver.solve(xi, U, verbose=verbose, **kwargs)
    dt = pc() - t0
    return X, U, J, {id_: (dt, ids) for id_ in ids}


def rhc_log_row(model_name, n_agents, i_trial, centralized, last, t, J, N, dt, converged, ids, times, subgraphs, left):
    """One CSV row of solve_rhc's log, in the format of distributed.py:190-194 / :215-219 (header: analysis.py:120-123):
    dynamics,n_agents,trial,centralized,last,t,J,horizon,dt,converged,ids,times,subgraphs,dist_left."""
    return (f'"{model_name}",{n_agents},{i_trial},{centralized},{last},{t},{J},{N},{dt},{converged},"{ids}",'
            f'"{times}","{subgraphs}","{left}"')


def solve_rhc(problem, x0, N, *args, centralized=True, n_d=2, step_size=1, J_converge=None, dist_converge=None,
              t_diverge=None, i_trial=None, verbose=False, **kwargs):
    """Receding-horizon loop around solve_centralized / solve_distributed (distributed.py:106-221); logs the
    same CSV rows through `logging.info`.  The warm start is drawn from NumPy's global RNG as in the reference."""
    if (J_converge is None) == (dist_converge is None):
        raise ValueError("Must either specify a convergence cost or distance")
    xf = problem.game_cost.xf
    n_states, n_agents = problem.dynamics.x_dims[0], problem.n_agents
    n_x, n_u = problem.dynamics.n_x, problem.dynamics.n_u

    def distance_to_goal(x):
        return np.linalg.norm((x - xf).reshape(n_agents, n_states)[:, :n_d], axis=1)

    if J_converge:
        keep_going = lambda x, J: J >= J_converge
    else:
        keep_going = lambda x, J: bool(np.any(distance_to_goal(x) > dist_converge))
    model_name = type(problem.dynamics.submodels[0]).__name__
    xi = np.asarray(x0, dtype=np.float64).reshape(1, -1)
    X = xi.copy()
    U = np.random.rand(N, n_u) * 0.01
    solver = ilqrSolver(problem, N)
    t, J, converged, dt, ids = 0, np.inf, True, problem.dynamics.dt, list(problem.ids)
    X_full = np.zeros((0, n_x)); U_full = np.zeros((0, n_u))
    times, subgraphs, left = [], [], distance_to_goal(xi.ravel()).tolist()
    while keep_going(xi.ravel(), J):
        if centralized:
            X, U, J, info = solve_centralized(solver, xi, U, ids, False, **kwargs)
        else:
            X, U, J, info = solve_distributed(problem, X, U, *args, verbose=False, **kwargs)
        xi = X[step_size]
        X_full = np.r_[X_full, X[:step_size]]; U_full = np.r_[U_full, U[:step_size]]
        X = np.r_[X[step_size:], np.tile(X[-1], (step_size, 1))]     # stay at the last visited state
        U = np.r_[U[step_size:], np.zeros((step_size, n_u))]
        times = [v[0] for v in info.values()]; subgraphs = [v[1] for v in info.values()]
        left = distance_to_goal(xi).tolist()
        logging.info(rhc_log_row(model_name, n_agents, i_trial, centralized, False, t, J, N, dt, converged, ids, times,
                                 subgraphs, left))
        if t_diverge and t >= t_diverge:
            converged = False
            break
        t += step_size * dt
    if not X_full.size and not U_full.size:
        X_full = np.asarray(x0, dtype=np.float64).copy(); U_full = np.zeros((1, n_u))
    _, J_full = ilqrSolver(problem, U_full.shape[0])._rollout(np.asarray(x0, dtype=np.float64), U_full)
    logging.info(rhc_log_row(model_name, n_agents, i_trial, centralized, True, U_full.shape[0] * dt, J_full, N, dt, converged,
                             ids, times, subgraphs, left))
    return X_full, U_full, J_full


def solve_rhc_scenarios(problem, x0, N, radius, xf=None, U0=None, centralized=False, n_d=2, step_size=1,
                        J_converge=None, dist_converge=None, t_diverge=None, window=None, i_trial=None, rows=None, **kwargs):
    """solve_rhc (distributed.py:106-221) for S Monte-Carlo scenarios of one k-agent problem in lock step: every
    receding-horizon round is ONE batched solve over the scenarios still running (solve_scenarios_distributed, or one
    ProblemBatch when centralized), with the reference's warm-start shift and stopping rules applied per scenario.
    The state of the loop -- current states, shifted warm starts (:184-185), executed prefixes -- stays on the device
    between rounds; per round only the S stopping flags come to the host.

    x0 (S, n_x); xf (S, n_x) goals (default: the problem's own); U0 (S, N, n_u) warm starts (default: drawn per
    scenario, in order, as solve_rhc draws them: np.random.rand(N, n_u) * 0.01).
    rows: a list to fill with S lists of CSV rows -- for every scenario exactly the rows solve_rhc logs for it
    (distributed.py:190-194,215-219: one per receding-horizon round and the closing one), in its format (rhc_log_row);
    i_trial: their trial numbers (a sequence of S, or one value).  The `times` field is wall-clock there and here: each
    agent is given its share of the round's batched solve.  kwargs: n_lqr_iter, tol, t_kill reach every solve.
    Returns a list of S tuples (X_full, U_full, J_full, converged), what solve_rhc returns plus its `converged` flag."""
    import torch
    from .batch import ProblemBatch
    from .device import to_dev
    from .dispatch import device_constants, solve_scenarios_distributed
    from .lowering import describe
    if (J_converge is None) == (dist_converge is None):
        raise ValueError("Must either specify a convergence cost or distance")
    d = describe(problem)
    k, dt = d["k"], d["dt"]
    n_x, n_u = problem.dynamics.n_x, problem.dynamics.n_u
    n_s = n_x // k
    x0 = np.asarray(x0, dtype=np.float64).reshape(-1, n_x)
    S = x0.shape[0]
    xf_h = np.broadcast_to(d["xf"], (S, n_x)).copy() if xf is None else np.asarray(xf, dtype=np.float64).reshape(S, n_x)
    U_h = np.stack([np.random.rand(N, n_u) * 0.01 for _ in range(S)]) if U0 is None else np.array(U0, dtype=np.float64)
    solve_kw = solve_kwargs(kwargs, "solve_rhc_scenarios")
    xf_d, U, xi = to_dev(xf_h), to_dev(U_h), to_dev(x0)
    X = None                                       # first round: the graph is built from x0 alone (distributed.py:152)
    J = torch.full((S,), float("inf"), dtype=torch.float64, device=xi.device)
    t = [0] * S                                    # the reference's t: the int 0 until the first `t += step_size * dt`
    converged = np.ones(S, dtype=bool)
    X_parts = [[] for _ in range(S)]; U_parts = [[] for _ in range(S)]
    want_rows = rows is not None
    ids = list(problem.ids)
    model_name = type(problem.dynamics.submodels[0]).__name__
    trial_of = list(i_trial) if isinstance(i_trial, (list, tuple, np.ndarray)) else [i_trial] * S
    row_log = [[] for _ in range(S)]
    last_fields = [([], [], None) for _ in range(S)]       # (times, subgraphs, distance left) of a scenario's last round

    def distance_left(idx):
        return torch.linalg.vector_norm((xi[idx] - xf_d[idx]).reshape(len(idx), k, n_s)[:, :, :n_d], dim=2)

    if want_rows:                                  # what solve_rhc holds before any round (a loop that never runs)
        left0 = distance_left(torch.arange(S, device=xi.device)).cpu().numpy()
        last_fields = [([], [], left0[s].tolist()) for s in range(S)]

    def keep_going(idx):      # one small device -> host read per round: the stopping flags
        if J_converge:
            return (J[idx] >= J_converge).cpu().numpy()
        return (distance_left(idx) > dist_converge).any(dim=1).cpu().numpy()

    active = np.nonzero(keep_going(torch.arange(S, device=xi.device)))[0]
    while active.size:
        ia = torch.as_tensor(active, device=xi.device)
        t_round = pc()
        bits = None
        if centralized:
            c = device_constants(d)
            pb = ProblemBatch(c["model"], c["n_dims"], xf_d[ia], c["Q"], c["R"], c["Qf"], d["radius"], dt, N,
                              w_ref=d["w_ref"], w_prox=d["w_prox"], B=len(active), hints=(k, n_s, n_u // k, c["word"]))
            r = pb.solve(xi[ia], U[ia], window=window, **solve_kw)
            Xa, Ua, Ja = r["X"], r["U"], r["J"]
        else:
            Xin = xi[ia][:, None, :] if X is None else X[ia]
            Xa, Ua, Ja, info = solve_scenarios_distributed(problem, Xin, U[ia], radius, xf=xf_d[ia], window=window, device_out=True,
                                                           desc=d, **solve_kw)
            bits = info["cluster_bits"]
        if X is None:
            X = torch.zeros((S, N + 1, n_x), dtype=torch.float64, device=xi.device)
        # what the round contributes to the executed trajectory: ONE (|active|, step_size, .) copy per round.  (Views into
        # Xa / Ua here kept every round's whole (|active|, N + 1, n_x) result alive on the device until the end.)
        Xkeep, Ukeep = Xa[:, :step_size].clone(), Ua[:, :step_size].clone()
        for j, s in enumerate(active):
            X_parts[s].append(Xkeep[j]); U_parts[s].append(Ukeep[j])
        xi[ia] = Xa[:, step_size]
        # warm start of the next round: shift, stay at the last visited state, zero controls (distributed.py:184-185)
        X[ia] = torch.cat([Xa[:, step_size:], Xa[:, -1:].expand(-1, step_size, -1)], dim=1)
        U[ia] = torch.cat([Ua[:, step_size:], torch.zeros((len(active), step_size, n_u), dtype=torch.float64, device=xi.device)], dim=1)
        J[ia] = Ja
        if want_rows:
            share = (pc() - t_round) / (len(active) * k)
            J_h, left_h = Ja.cpu().numpy(), distance_left(ia).cpu().numpy()
            for j, s in enumerate(active):
                if centralized:
                    subgraphs = [ids] * k                                    # solve_centralized: {id: (dt, ids)}
                else:      # the agent's own id as a Python int, its neighbours as NumPy ints (quirk Q10, as the rows print them)
                    subgraphs = [sorted([ids[i]] + [np.int64(ids[q]) for q in range(k) if q != i and (int(bits[j, i]) >> q) & 1])
                                 for i in range(k)]
                last_fields[s] = ([share] * k, subgraphs, left_h[j].tolist())
                row_log[s].append(rhc_log_row(model_name, k, trial_of[s], centralized, False, t[s], float(J_h[j]), N, dt,
                                              bool(converged[s]), ids, *last_fields[s]))
        diverged = np.zeros(len(active), dtype=bool)
        if t_diverge:
            diverged = np.array([t[s] >= t_diverge for s in active])
            converged[active[diverged]] = False
        for s in active[~diverged]:
            t[s] += step_size * dt
        still = keep_going(ia) & ~diverged
        active = active[still]
    # the executed trajectories, and J_full = the cost of rolling the executed controls out from x0 (distributed.py:208): one
    # batched rollout per executed length instead of one launch per scenario
    XU = []
    for s in range(S):
        if X_parts[s]:
            XU.append((torch.cat(X_parts[s]).cpu().numpy(), torch.cat(U_parts[s]).cpu().numpy()))
        else:
            XU.append((x0[s].copy(), np.zeros((1, n_u))))
    J_full = np.zeros(S)
    by_len = {}
    for s in range(S):
        by_len.setdefault(XU[s][1].shape[0], []).append(s)
    for L_exec, idx in by_len.items():
        pbl = ProblemBatch(d["model"], d["n_dims"], xf_h[idx], d["Q"], d["R"], d["Qf"], d["radius"], dt, L_exec,
                           w_ref=d["w_ref"], w_prox=d["w_prox"], B=len(idx))
        _, Jl = pbl.rollout(x0[idx], np.stack([XU[s][1] for s in idx]))
        J_full[idx] = Jl.cpu().numpy()
    out = []
    for s in range(S):
        Xf, Uf = XU[s]
        out.append((Xf, Uf, float(J_full[s]), bool(converged[s])))
        if want_rows:
            row_log[s].append(rhc_log_row(model_name, k, trial_of[s], centralized, True, Uf.shape[0] * dt, out[-1][2], N, dt,
                                          bool(converged[s]), ids, *last_fields[s]))
    if want_rows:
        rows.extend(row_log)
    return out
