"""The batched Monte-Carlo harness (dpilqr_amd/analysis.py; reference: scripts/analysis.py:35-174).  Pinned to the reference
DIRECTLY by G10 (the real multi_agent_run on a seeded stream: test_harness_rows_equal_the_references_own); beside that its CSV
equals, row for row, what per-trial solve_rhc runs log on the same seeded inputs -- and solve_rhc itself is pinned to the
reference's own rows by G7 (tests/test_gpu_api.py::test_solve_rhc_vs_reference)."""
import logging

import numpy as np
import pytest

from tests.test_host_logic import _parse_row

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dp():
    import dpilqr_amd
    from dpilqr_amd import _lib
    _lib.require_gpu()
    return dpilqr_amd


class _Rows(logging.Handler):
    def __init__(self):
        super().__init__()
        self.rows = []

    def emit(self, record):
        self.rows.append(record.getMessage())


def _same_rows(mine, ref):
    assert len(mine) == len(ref), (len(mine), len(ref))
    for m, r in zip(mine, ref):
        a, b = _parse_row(m), _parse_row(r)
        for key in ("model_name", "n_agents", "i_trial", "centralized", "last", "N", "dt", "converged", "ids"):
            assert a[key] == b[key], (key, m, r)
        assert repr(a["subgraphs"]) == repr(b["subgraphs"]) and repr(a["t"]) == repr(b["t"])
        assert abs(a["J"] - b["J"]) <= 1e-6 * abs(b["J"]) and np.allclose(a["left"], b["left"], rtol=1e-6, atol=1e-9)
        assert len(a["times"]) == len(b["times"])                 # wall-clock seconds: the one field that cannot match


def test_harness_rows_equal_the_references_own(dp, golden):
    """G10 (tests/golden/make_golden.py g10): the reference's multi_agent_run itself, run on the harness's seed of each trial.
    The batched harness on the same cells -- all trials of a cell in ONE call, as the study runs them -- logs the reference's
    rows: every field but the wall-clock `times` (J and the distances to 1e-6), in the reference's order, and returns its
    trajectories (1e-5 of the largest entry; the receding-horizon loop chains up to eleven solves)."""
    from dpilqr_amd import analysis
    from tests.test_host_logic import _g10_cells
    z = golden("g10_harness")
    dt, N, radius, energy, seed0 = float(z["dt"]), int(z["N"]), float(z["radius"]), float(z["energy"]), int(z["seed0"])
    kw = dict(dist_converge=float(z["dist_converge"]), t_diverge=float(z["t_diverge"]))
    cells = {}
    for tag, model, n_agents, i_trial in _g10_cells(z):
        cells.setdefault((model, n_agents), []).append((tag, i_trial))
    assert len(cells) == 3
    for (model, n_agents), trials in cells.items():
        n_states, n_d = model(-1).n_x, (3 if model.__name__ == "QuadcopterDynamics6D" else 2)
        got = []
        res = analysis.multi_agent_run(model, [n_states] * n_agents, dt, N, radius, n_d=n_d, trials=[i for _, i in trials], seed0=seed0,
                                       emit=got.append, energy=energy, t_kill=None, verbose=False, **kw)
        ref_rows = [str(r) for tag, _ in trials for r in z[tag + "_rows"]]
        _same_rows(got, ref_rows)
        for tag, i in trials:
            (Xc, Uc, Jc, _), (Xd, Ud, Jd, _) = res[i]
            for mine, ref in ((Xc, z[tag + "_Xc"]), (Uc, z[tag + "_Uc"]), (Xd, z[tag + "_Xd"]), (Ud, z[tag + "_Ud"])):
                assert mine.shape == ref.shape and np.max(np.abs(mine - ref)) <= 1e-5 * np.max(np.abs(ref)), tag
            assert abs(Jc - float(z[tag + "_Jc"])) <= 1e-6 * abs(float(z[tag + "_Jc"])), tag
            assert abs(Jd - float(z[tag + "_Jd"])) <= 1e-6 * abs(float(z[tag + "_Jd"])), tag


@pytest.mark.parametrize("model_name,n_agents", [("DoubleIntDynamics4D", 3), ("UnicycleDynamics4D", 4), ("QuadcopterDynamics6D", 3)])
def test_harness_rows_equal_per_trial_solve_rhc(dp, model_name, n_agents):
    """analysis 1 (t_kill = None): every row of the batched harness equals the row per-trial solve_rhc logs, both branches,
    in the reference's order (trial by trial: centralized rows, then distributed rows)."""
    from dpilqr_amd import analysis
    model = getattr(dp, model_name)
    n_states, n_d = model(-1).n_x, (3 if model_name == "QuadcopterDynamics6D" else 2)
    dt, N, radius, trials, seed0 = 0.1, 20, 0.5, [0, 1, 2], 4
    kw = dict(dist_converge=0.1, t_diverge=3.0)
    got = []
    res = analysis.multi_agent_run(model, [n_states] * n_agents, dt, N, radius, n_d=n_d, trials=trials, seed0=seed0, emit=got.append,
                                   energy=10.0, t_kill=None, verbose=False, **kw)
    # per trial, the way scripts/analysis.py:35-107 does it -- on the stream the harness documents
    log = logging.getLogger(); old = log.level; log.setLevel(logging.INFO)
    h = _Rows(); log.addHandler(h)
    try:
        for i in trials:
            np.random.seed(analysis.seed_of(model, n_agents, i, seed0))
            x0, xf = dp.random_setup(n_agents, n_states, is_rotation=False, rel_dist=n_agents, var=n_agents / 2, n_d=n_d, random=True,
                                     energy=10.0)
            prob = analysis.build_problem(model, n_agents, dt, radius, xf, n_d)
            Xc, Uc, Jc = dp.solve_rhc(prob, x0, N, radius, centralized=True, n_d=n_d, step_size=analysis.STEP_SIZE, i_trial=i, **kw)
            Xd, Ud, Jd = dp.solve_rhc(prob, x0, N, radius, centralized=False, n_d=n_d, step_size=analysis.STEP_SIZE, i_trial=i, **kw)
            (bXc, bUc, bJc, _), (bXd, bUd, bJd, _) = res[i]
            assert bXc.shape == Xc.shape and np.allclose(bXc, Xc, rtol=1e-6, atol=1e-8) and abs(bJc - Jc) <= 1e-6 * abs(Jc)
            assert bXd.shape == Xd.shape and np.allclose(bXd, Xd, rtol=1e-6, atol=1e-8) and abs(bJd - Jd) <= 1e-6 * abs(Jd)
    finally:
        log.removeHandler(h); log.setLevel(old)
    _same_rows(got, h.rows)
    assert sum(_parse_row(r)["last"] for r in got) == 2 * len(trials)


def test_monte_carlo_analysis_writes_the_references_csv(dp, tmp_path):
    """monte_carlo_analysis(True): the reference's default study (t_kill = dt, t_diverge = N dt) -- header, row format,
    both branches of every trial of every (model, team size) cell."""
    from dpilqr_amd import analysis
    out = tmp_path / "mc.csv"
    res = analysis.monte_carlo_analysis(True, n_trials=3, n_agents_iter=(3, 4), models=analysis.MODELS[:2], N=15, log_file=out)
    logging.shutdown()
    lines = out.read_text().splitlines()
    assert lines[0] == "dynamics,n_agents,trial,centralized,last,t,J,horizon,dt,converged,ids,times,subgraphs,dist_left"
    rows = [_parse_row(l) for l in lines[1:]]
    assert {(r["model_name"], r["n_agents"]) for r in rows} == {(m.__name__, k) for m in analysis.MODELS[:2] for k in (3, 4)}
    last = [r for r in rows if r["last"]]
    assert len(last) == 2 * 3 * 4 and len(res) == 4
    assert all(np.isfinite(r["J"]) for r in rows)
    # the reference's order inside a cell: trial 0 centralized, trial 0 distributed, trial 1 centralized, ...
    cell = [r for r in rows if r["model_name"] == "UnicycleDynamics4D" and r["n_agents"] == 4]
    order = [(r["i_trial"], r["centralized"]) for r in cell if r["last"]]
    assert order == [(0, True), (0, False), (1, True), (1, False), (2, True), (2, False)]


def test_a_tiny_t_kill_is_one_iteration_per_solve(dp):
    """limit_solve_time's mechanism end to end: t_kill reaches every solve of every round of both branches (a limit below
    one clock tick = one iLQR iteration per solve = the same study with n_lqr_iter = 1)."""
    from dpilqr_amd import analysis
    model = dp.DoubleIntDynamics4D
    a, b, c = [], [], []
    common = dict(n_d=2, trials=[0, 1], seed0=1, energy=10.0, dist_converge=0.1, t_diverge=1.5)
    analysis.multi_agent_run(model, [4] * 4, 0.1, 15, 0.5, emit=a.append, t_kill=1e-12, **common)
    analysis.multi_agent_run(model, [4] * 4, 0.1, 15, 0.5, emit=b.append, n_lqr_iter=1, **common)
    analysis.multi_agent_run(model, [4] * 4, 0.1, 15, 0.5, emit=c.append, **common)
    _same_rows(a, b)
    assert [_parse_row(r)["J"] for r in a] != [_parse_row(r)["J"] for r in c]
