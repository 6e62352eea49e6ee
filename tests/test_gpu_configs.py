"""BASELINE.json's configurations at their stated sizes, on the GPU, against the oracle (cfg1, cfg2 and cfg5 also have
reference-generated golden tests: test_gpu_api.py::test_solver_solve[cfg1], test_gpu_parity.py, test_gpu_big.py).

  cfg3  15-agent UnicycleDynamics4D DP-iLQR, proximity-graph split -> variable-size sub-problem batch, T = 100
  cfg4  Monte-Carlo random-goal seeds x 10-agent QuadcopterDynamics6D, T = 75 (1024 of the 8192 seeds here -- one rank's
        share at 8 GPUs; all 8192 on one GPU: scripts/montecarlo.py, profiles/)
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from tests.golden_util import relerr

pytestmark = pytest.mark.gpu
G = 9.80665


@pytest.fixture(scope="module")
def dp():
    import dpilqr_amd
    from dpilqr_amd import _lib
    _lib.require_gpu()
    return dpilqr_amd


def build(dp, model_cls, k, x0, xf, radius=0.5, dt=0.1):
    n_s = model_cls(dt).n_x
    ids = [100 + i for i in range(k)]
    if n_s == 4:
        Q, R = np.diag([1.0, 1, 0, 0]), np.eye(2)
    else:
        Q, R = 50.0 * np.eye(6), np.eye(3)
    Qf = 1000.0 * np.eye(n_s)
    n_d = 3 if n_s == 6 else 2
    dyn = dp.MultiDynamicalModel([model_cls(dt, id_) for id_ in ids])
    refs = [dp.ReferenceCost(xf[i * n_s:(i + 1) * n_s], Q.copy(), R.copy(), Qf.copy(), id_) for i, id_ in enumerate(ids)]
    return dp.ilqrProblem(dyn, dp.GameCost(refs, dp.ProximityCost([n_s] * k, radius, [n_d] * k))), (Q, R, Qf, n_d)


@pytest.mark.parametrize("seed", [0, 3])
def test_cfg3_fifteen_unicycles_T100_solve_distributed(dp, seed):
    """One DP-iLQR call from x0 (the reference's first call) and a second one seeded with the first result: 15
    sub-problems of 8..15 agents each (n_x 32..60) in one bucketed dispatch, against the oracle's dispatch layer."""
    from oracle import oracle as orc
    from dpilqr_amd.util import random_setup
    k, T = 15, 100
    np.random.seed(seed)
    a, b = random_setup(k, 4, is_rotation=False, rel_dist=k, var=k / 2, n_d=2, random=True, energy=10.0)
    x0, xf = a.ravel(), b.ravel()
    prob, (Q, R, Qf, nd) = build(dp, dp.UnicycleDynamics4D, k, x0, xf)
    U0 = np.zeros((T, 2 * k))
    Xd, Ud, Jf, info = dp.solve_distributed(prob, x0[None], U0, 0.5, verbose=False)
    p = orc.Problem([3] * k, [2] * k, xf, Q, R, Qf, 0.5, 0.1, T)
    Xo, Uo, Jo, graph = orc.solve_distributed(p, x0[None], U0, 0.5)
    sizes = sorted(len(v) for v in graph.values())
    assert sizes[-1] >= 8                                         # the nearly-centralised regime of cfg3
    assert [sorted(int(j) - 100 for j in info[100 + i][1]) for i in range(k)] == [graph[i] for i in range(k)]
    Xp, Up, Jp, _ = orc.solve_distributed(p, (x0 * (1 + 1e-13))[None], U0, 0.5)
    ns, nc = 4, 2
    n_plain = 0
    for i in range(k):      # agent by agent: its own sub-problem's answer
        cols, ucols = slice(i * ns, (i + 1) * ns), slice(i * nc, (i + 1) * nc)
        sens = relerr(Xp[:, cols], Xo[:, cols])
        err = relerr(Xd[:, cols], Xo[:, cols])
        assert err <= 100 * max(1e-10, sens), (i, err, sens)
        if sens < 1e-7:       # the usual amplification of a 1e-13 perturbation (up to ~1e6): the fixed 1e-5 holds too
            n_plain += 1
            assert err < 1e-5 and relerr(Ud[:, ucols], Uo[:, ucols]) < 1e-5, (i, err)
    assert n_plain >= 3
    # second call, the receding-horizon pattern: graph from the whole previous trajectory
    Xd2, Ud2, Jf2, _ = dp.solve_distributed(prob, Xo, Uo, 0.5, verbose=False)
    Xo2, Uo2, Jo2, _ = orc.solve_distributed(p, Xo, Uo, 0.5)
    Xp2, _, _, _ = orc.solve_distributed(p, Xo * (1 + 1e-13), Uo, 0.5)
    for i in range(k):
        cols = slice(i * ns, (i + 1) * ns)
        assert relerr(Xd2[:, cols], Xo2[:, cols]) <= 100 * max(1e-10, relerr(Xp2[:, cols], Xo2[:, cols])), i
    assert np.isfinite(Xd2).all() and np.isfinite(Jf2)


def test_cfg4_monte_carlo_1024_seeds_ten_quadcopters_T75(dp):
    """1024 random-goal scenarios of 10 QuadcopterDynamics6D (an eighth of cfg4's 8192: one rank's share at 8 GPUs) through
    the many-scenario front end -- one windowed device solve per cluster size -- with EVERY distinct sub-problem solve
    (about 8 000 of 1..10 agents) held to the ensemble envelope of oracle/parity.py through every iteration, no agent
    exempt; then the interaction graphs bit for bit, the stitching column by column against the audited sub-problem
    results, and J_full against the oracle's rollout of the stitched controls."""
    from oracle import oracle as orc, parity
    from dpilqr_amd.dispatch import solve_scenarios_distributed
    from dpilqr_amd.util import random_setup
    k, T, S = 10, 75, 1024
    ns, nc = 6, 3
    x0 = np.zeros((S, ns * k)); xf = np.zeros((S, ns * k))
    for s in range(S):
        np.random.seed(s)
        a, b = random_setup(k, ns, is_rotation=False, rel_dist=k, var=k / 2, n_d=3, random=True, energy=10.0)
        x0[s], xf[s] = a.ravel(), b.ravel()
    prob, (Q, R, Qf, nd) = build(dp, dp.QuadcopterDynamics6D, k, x0[0], xf[0])
    U0 = np.zeros((S, T, nc * k)); U0[:, :, 0::3] = G
    Xd, Ud, J, info = solve_scenarios_distributed(prob, x0[:, None, :], U0, 0.5, xf=xf, audit=True)
    assert info["n_subproblems"] == S * k and set(info["sizes"]) <= set(range(1, 11)) and len(info["sizes"]) >= 6
    audit = info["audit"]
    assert sum(len(a["J"]) for a in audit.values()) == info["n_unique"]

    # (1) every distinct sub-problem solve against the oracle's envelope
    n_items = n_flipped = n_tight = 0
    worst = 0.0
    for kc, a in sorted(audit.items()):
        proto = orc.Problem([4] * kc, [3] * kc, a["xf"][0], Q, R, Qf, 0.5, 0.1, T)
        rep = parity.envelope(a, proto, a["x0"], a["xf"], a["U0"])
        sm = rep["summary"]
        assert sm["all_ok"], (kc, sm, [f"item {i}: {w}" for i, w in enumerate(rep["why"]) if w][:6])
        tight = rep["spreadX"] < 1e-6
        assert (rep["errX"][tight] < 1e-5).all() and (rep["errU"][tight] < 1e-5).all(), kc
        n_items += sm["items"]; n_flipped += int(rep["flipped"].sum()); n_tight += int(tight.sum())
        worst = max(worst, sm["max_err_over_bound"])
    print(f"cfg4 x {S} seeds: {n_items} sub-problem solves, {n_flipped} with a decision that is not the oracle's own "
          f"(all explained), {n_tight} with a reference ensemble tighter than 1e-6, worst err/bound {worst:.3f}")
    assert n_tight > 0.8 * n_items and n_flipped < 0.05 * n_items

    # (2) graphs, stitching, J_full
    lookup = {kc: {a["x0"][j].tobytes() + a["xf"][j].tobytes(): j for j in range(len(a["J"]))} for kc, a in audit.items()}
    for s in range(S):
        p = orc.Problem([4] * k, [3] * k, xf[s], Q, R, Qf, 0.5, 0.1, T)
        graph = orc.define_inter_graph_threshold(x0[s][None], 0.5, k, ns)
        assert list(info["cluster_bits"][s]) == [sum(1 << j for j in graph[i]) for i in range(k)], s
        for i in range(k):
            mem = graph[i]
            key = np.concatenate([x0[s, a_ * ns:(a_ + 1) * ns] for a_ in mem]).tobytes() + \
                np.concatenate([xf[s, a_ * ns:(a_ + 1) * ns] for a_ in mem]).tobytes()
            a = audit[len(mem)]; j = lookup[len(mem)][key]; pos = mem.index(i)
            assert np.array_equal(Xd[s][:, i * ns:(i + 1) * ns], a["X"][j][:, pos * ns:(pos + 1) * ns]), (s, i)
            assert np.array_equal(Ud[s][:, i * nc:(i + 1) * nc], a["U"][j][:, pos * nc:(pos + 1) * nc]), (s, i)
        _, Jo = p.rollout(x0[s], Ud[s])
        assert abs(J[s] - Jo) <= 1e-9 * abs(Jo) or not np.isfinite(Jo), (s, J[s], Jo)
    assert np.isfinite(Xd).all() and np.isfinite(J).all()
