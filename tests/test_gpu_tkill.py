"""t_kill, the reference's real-time bail-out (control.py:213-218; scripts/analysis.py:145-147 runs its default study
with t_kill = dt), inside the batched device solve: every item has its own clock from its admission, the decision is
taken on the device at the point where the reference takes it (after an accepted, unconverged step), and a killed item's
result is exactly the solve with n_lqr_iter = its n_bwd -- which is how the (clock-free) oracle checks it."""
import warnings

import numpy as np
import pytest

from tests.golden_util import relerr
from tests.test_host_logic import problem_from

pytestmark = pytest.mark.gpu
KILLED, CONVERGED, LS_FAILED, MAX_ITER = 5, 1, 2, 3


@pytest.fixture(scope="module")
def dp():
    import dpilqr_amd
    from dpilqr_amd import _lib
    _lib.require_gpu()
    assert _lib.STATUS_KILLED == KILLED
    return dpilqr_amd


def cfg2_batch(dp, B, k=5, T=50):
    from dpilqr_amd.util import random_setup
    x0 = np.zeros((B, 4 * k)); xf = np.zeros((B, 4 * k))
    for s in range(B):
        np.random.seed(s)
        a, b = random_setup(k, 4, is_rotation=False, rel_dist=k, var=k / 2, n_d=2, random=True, energy=10.0)
        x0[s], xf[s] = a.ravel(), b.ravel()
    Q, R, Qf = np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4)
    pb = dp.ProblemBatch([0] * k, [2] * k, xf, Q, R, Qf, 0.5, 0.1, T)
    return pb, x0, xf, np.zeros((B, T, 2 * k)), (Q, R, Qf)


def host(r):
    return {k: v.cpu().numpy() for k, v in r.items()}


def test_tiny_deadline_every_item_runs_exactly_one_iteration(dp):
    """t_kill below one clock tick: the reference would still finish the iteration it is in (the check sits at the END of
    an iteration), so every item runs one backward pass + line search and ends KILLED -- unless that very iteration
    converged or failed its line search, which the reference checks first.  The iterate is the accepted one: finite,
    cheaper than the warm start's rollout, and what a one-iteration solve returns (GPU bit for bit, oracle to 1e-9)."""
    from oracle import oracle as orc
    B = 96
    pb, x0, xf, U0, (Q, R, Qf) = cfg2_batch(dp, B)
    r = host(pb.solve(x0, U0, t_kill=1e-12, trace=True))
    assert (r["n_bwd"] == 1).all()
    assert set(np.unique(r["status"])) <= {KILLED, CONVERGED, LS_FAILED}
    assert (r["status"] == KILLED).sum() >= B * 3 // 4
    _, J0 = pb.rollout(x0, U0)
    J0 = J0.cpu().numpy()
    killed = r["status"] == KILLED
    assert np.isfinite(r["X"]).all() and np.isfinite(r["U"]).all() and np.isfinite(r["J"]).all()
    assert (r["J"][killed] < J0[killed]).all()                       # an accepted step: the cost went down
    one = host(pb.solve(x0, U0, n_lqr_iter=1, trace=True))
    assert (one["status"][killed] == MAX_ITER).all()
    for key in ("X", "U", "J", "n_fwd"):
        assert np.array_equal(r[key], one[key]), key
    assert np.array_equal(r["trace"][:, 0], one["trace"][:, 0])
    proto = orc.Problem([0] * 5, [2] * 5, xf[0], Q, R, Qf, 0.5, 0.1, 50)
    o = orc.solve_batch(proto, x0, xf, U0, n_lqr_iter=1)
    assert np.array_equal(o["n_fwd"], r["n_fwd"])
    assert max(relerr(r["X"][i], o["X"][i]) for i in range(B)) < 1e-9 and relerr(r["J"], o["J"]) < 1e-9


def test_generous_deadline_changes_nothing(dp):
    B = 64
    pb, x0, xf, U0, _ = cfg2_batch(dp, B)
    a = host(pb.solve(x0, U0, trace=True)); b = host(pb.solve(x0, U0, t_kill=30.0, trace=True))
    for key in ("X", "U", "J", "status", "n_bwd", "n_fwd"):
        assert np.array_equal(a[key], b[key]), key
    assert KILLED not in b["status"]
    c = host(pb.solve(x0, U0, t_kill=0.0)); d = host(pb.solve(x0, U0, t_kill=None))      # falsy: no limit (control.py:213)
    assert np.array_equal(c["X"], a["X"]) and np.array_equal(d["X"], a["X"])


def test_mid_solve_deadline_returns_a_prefix_of_the_unlimited_solve(dp):
    """A deadline inside the solve: killed items carry fewer iterations than their unlimited solve, their decision trace
    is a PREFIX of the unlimited one, and (X, U, J) are those of the solve with n_lqr_iter = n_bwd.  Items that finish on
    their own before their time is up are untouched."""
    import time
    import torch
    B = 2048
    pb, x0, xf, U0, _ = cfg2_batch(dp, B)
    full = host(pb.solve(x0, U0, trace=True))                      # warm: allocator, kernels
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pb.solve(x0, U0)
    torch.cuda.synchronize(); t_full = time.perf_counter() - t0
    got = None
    for frac in (0.35, 0.2, 0.5, 0.1):                            # the box's speed is not ours to know: look for a cut inside
        r = host(pb.solve(x0, U0, t_kill=t_full * frac, trace=True))
        n_k = int((r["status"] == KILLED).sum())
        if 0 < n_k < B:
            got = r
            break
    assert got is not None, "no deadline fraction cut the batch in two"
    r = got
    killed = r["status"] == KILLED
    assert (r["n_bwd"][killed] < full["n_bwd"][killed]).all()
    same = ~killed
    for key in ("X", "U", "J", "status", "n_bwd", "n_fwd"):
        assert np.array_equal(r[key][same], full[key][same]), key
    for i in np.nonzero(killed)[0]:
        nb = r["n_bwd"][i]
        assert np.array_equal(r["trace"][i, :nb], full["trace"][i, :nb])
        assert r["trace"][i, nb - 1, 1] >= 0 and r["J"][i] == r["trace"][i, nb - 1, 2]      # the last step was accepted
    # the killed items again, each with n_lqr_iter = the count it was stopped at: the same iterate bit for bit
    for nb in np.unique(r["n_bwd"][killed])[:4]:
        idx = np.nonzero(killed & (r["n_bwd"] == nb))[0]
        cap = host(pb.solve(x0, U0, n_lqr_iter=int(nb)))
        assert np.array_equal(cap["X"][idx], r["X"][idx]) and np.array_equal(cap["U"][idx], r["U"][idx])


def test_window_smaller_than_the_batch_each_item_has_its_own_clock(dp):
    """Items admitted late start their clock at admission (the reference's t0 is per solve): with a window of 64 out of
    256 and a tiny deadline every item still gets its one iteration."""
    B = 256
    pb, x0, xf, U0, _ = cfg2_batch(dp, B)
    r = host(pb.solve(x0, U0, t_kill=1e-12, window=64))
    assert (r["n_bwd"] == 1).all() and (r["status"] == KILLED).sum() >= B * 3 // 4


def test_enqueue_form_honours_the_deadline(dp):
    import torch
    B = 128
    pb, x0, xf, U0, _ = cfg2_batch(dp, B)
    r, st = pb.solve_enqueue(x0, U0, n_global_iter=4, t_kill=1e-12)
    torch.cuda.synchronize()
    assert (r["n_bwd"].cpu().numpy() == 1).all() and (r["status"].cpu().numpy() != 0).all()
    ref = host(pb.solve(x0, U0, t_kill=1e-12))
    assert np.array_equal(r["X"].cpu().numpy(), ref["X"])


def test_reference_surface_routes_t_kill_to_the_device(dp, golden):
    """ilqrSolver.solve(t_kill=...), solve_subproblem / solve_problem_list and solve_distributed's **kwargs: recognised
    plugins stay on the batched device path (round 4 fell back to a per-pass host loop whenever t_kill was given)."""
    z = golden("g4_solves_misc"); tag = "cfg1"
    prob = problem_from(z, tag + "_")
    x0 = z[tag + "_x0"].reshape(-1)
    N = 50
    sol = dp.ilqrSolver(prob, N)
    called = []
    orig = sol._solve_host_loop
    sol._solve_host_loop = lambda *a, **k: called.append(1) or orig(*a, **k)
    X, U, J = sol.solve(x0, None, t_kill=1e-12, verbose=False)
    assert not called and sol.status == KILLED and sol.n_bwd == 1
    X1, U1, J1 = dp.ilqrSolver(prob, N).solve(x0, None, n_lqr_iter=1, verbose=False)
    assert np.array_equal(X, X1) and np.array_equal(U, U1) and J == J1
    # μ, Δ left as the reference would leave them: one accepted step
    assert sol.μ == 0.5 and sol.Δ == 0.5
    from dpilqr_amd.dispatch import solve_problem_list
    res = solve_problem_list([prob, prob], [x0, x0], [np.zeros((N, 6))] * 2, t_kill=1e-12)
    assert all(info["status"] == KILLED and info["n_bwd"] == 1 for *_, info in res)
    assert np.array_equal(res[0][0], X)
    Xd, Ud, Jd, info = dp.solve_distributed(prob, x0.reshape(1, -1), np.zeros((N, 6)), 0.5, verbose=False, t_kill=1e-12)
    Xd1, Ud1, Jd1, _ = dp.solve_distributed(prob, x0.reshape(1, -1), np.zeros((N, 6)), 0.5, verbose=False, n_lqr_iter=1)
    assert np.array_equal(Xd, Xd1) and np.array_equal(Ud, Ud1) and Jd == Jd1


def test_rhc_scenarios_pass_t_kill_and_warn_on_unknown_kwargs(dp, golden):
    z = golden("g4_solves_misc"); tag = "cfg1"
    prob = problem_from(z, tag + "_")
    S, N = 3, 12
    rng = np.random.default_rng(5)
    x0 = np.tile(z[tag + "_x0"].reshape(1, -1), (S, 1)) + np.pad(rng.normal(scale=0.1, size=(S, 3, 2)), ((0, 0), (0, 0), (0, 2))).reshape(S, -1)
    U0 = np.zeros((S, N, 6))
    kw = dict(centralized=False, step_size=3, dist_converge=0.1, t_diverge=2.0, U0=U0)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                              # t_kill, n_lqr_iter, tol, verbose: known, no warning
        a = dp.solve_rhc_scenarios(prob, x0, N, 0.5, t_kill=30.0, verbose=False, **kw)
        b = dp.solve_rhc_scenarios(prob, x0, N, 0.5, **kw)
        c = dp.solve_rhc_scenarios(prob, x0, N, 0.5, t_kill=1e-12, **kw)
        d = dp.solve_rhc_scenarios(prob, x0, N, 0.5, n_lqr_iter=1, **kw)
    for s in range(S):
        assert np.array_equal(a[s][0], b[s][0]) and a[s][2] == b[s][2]       # a generous limit: the unlimited loop
        assert np.array_equal(c[s][0], d[s][0]) and c[s][2] == d[s][2]       # a tiny one: one iteration per solve
    assert any(not np.array_equal(b[s][0], c[s][0]) for s in range(S))       # ...and it did reach the solves
    with pytest.warns(UserWarning, match="n_iter_lqr"):
        dp.solve_rhc_scenarios(prob, x0[:1], N, 0.5, n_iter_lqr=3, **dict(kw, U0=U0[:1]))


@pytest.mark.parametrize("case", ["fp32", "big", "wg"])
def test_every_solve_path_reads_the_clock(dp, case):
    """The decision lives in two places -- linesearch_decide (the wavefront / team line searches) and k_forward's own copy (the
    generic kernel: the fp32 arm, clusters beyond n_x = 60) -- and every sweep family feeds them: a limit below one tick is one
    iteration per item on each."""
    import torch
    from dpilqr_amd.util import random_setup
    model, k, T, dtype = {"fp32": (0, 3, 20, torch.float32), "big": (3, 16, 12, torch.float64), "wg": (4, 6, 20, torch.float64)}[case]
    ns, nc = (6, 3) if model == 4 else (4, 2)
    nd = 3 if ns == 6 else 2
    B = 24
    x0 = np.zeros((B, k * ns)); xf = np.zeros((B, k * ns))
    for s in range(B):
        np.random.seed(700 + s)
        a, b = random_setup(k, ns, is_rotation=False, rel_dist=k, var=k / 2, n_d=nd, random=True, energy=10.0)
        x0[s], xf[s] = a.ravel(), b.ravel()
    Q, R = (50.0 * np.eye(6), np.eye(3)) if ns == 6 else (np.diag([1.0, 1, 0, 0]), np.eye(2))
    U0 = np.zeros((B, T, k * nc))
    if model == 4:
        U0[:, :, 0::3] = 9.80665
    pb = dp.ProblemBatch([model] * k, [nd] * k, xf, Q, R, 1000.0 * np.eye(ns), 0.5, 0.1, T)
    r = host(pb.solve(x0, U0, t_kill=1e-12, dtype=dtype))
    one = host(pb.solve(x0, U0, n_lqr_iter=1, dtype=dtype))
    assert (r["n_bwd"] == 1).all() and set(np.unique(r["status"])) <= {KILLED, CONVERGED, LS_FAILED}
    assert (r["status"] == KILLED).sum() >= B // 2
    assert np.array_equal(r["X"], one["X"]) and np.array_equal(r["J"], one["J"])
    free = host(pb.solve(x0, U0, dtype=dtype))
    assert (free["n_bwd"] >= r["n_bwd"]).all() and (free["n_bwd"] > 1).any()
