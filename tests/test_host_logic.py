"""Host logic of the product (no kernels involved): graph building, splitting, lowering, sharding maths.
Checked against the golden vectors of the real reference.  CPU only."""
import numpy as np
import pytest

import dpilqr_amd as dp
from dpilqr_amd import lowering
from dpilqr_amd.sharding import pack_results, shard_bounds, unpack_results

MODEL_CLASSES = {0: dp.DoubleIntDynamics4D, 1: dp.DoubleIntDynamics6D, 2: dp.CarDynamics3D, 3: dp.UnicycleDynamics4D,
                 4: dp.QuadcopterDynamics6D, 5: dp.HumanDynamics6D, 6: dp.HumanDynamicsLin6D, 7: dp.QuadcopterDynamics12D,
                 8: dp.HumanDynamics6DPadded12}


def problem_from(z, prefix=""):
    g = lambda k: z[prefix + k]
    k, ns = int(g("k")), int(g("n_s"))
    ids = [int(i) for i in g("ids")]
    dyn = dp.MultiDynamicalModel([MODEL_CLASSES[int(m)](float(g("dt")), id_) for m, id_ in zip(g("model"), ids)])
    refs = [dp.ReferenceCost(g("xf")[i * ns:(i + 1) * ns], g("Q")[i], g("R")[i], g("Qf")[i], ids[i]) for i in range(k)]
    return dp.ilqrProblem(dyn, dp.GameCost(refs, dp.ProximityCost([ns] * k, float(g("radius")), [int(v) for v in g("n_dims")])))


@pytest.mark.parametrize("tag", ["uni5", "quad10", "uni8"])
def test_graph_and_split(golden, tag):
    z = golden("g5_dispatch")
    prob = problem_from(z, tag + "_")
    ids = prob.ids
    for Xkey, adjkey in (("x0", "adj_x0"), ("X_dec", "adj_traj")):
        X = z[f"{tag}_{Xkey}"]
        graph = dp.define_inter_graph_threshold(np.atleast_2d(X), float(z[tag + "_radius"]), prob.game_cost.x_dims, ids)
        adj = np.zeros((len(ids), len(ids)), dtype=np.int32)
        for i, id_ in enumerate(ids):
            assert graph[id_] == sorted(graph[id_]) and id_ in graph[id_]
            adj[i, [ids.index(j) for j in graph[id_]]] = 1
        np.testing.assert_array_equal(adj, z[f"{tag}_{adjkey}"])
    graph = dp.define_inter_graph_threshold(z[tag + "_x0"].reshape(1, -1), 0.5, prob.game_cost.x_dims, ids)
    for i, piece in enumerate(dp.split_graph(z[tag + "_x0"].reshape(1, -1), prob.game_cost.x_dims, graph)):
        np.testing.assert_array_equal(piece, z[f"{tag}_x0split_{i}"])
    subs = prob.split(graph)
    for sub, id_ in zip(subs, ids):
        assert sub.ids == graph[id_] and sub.game_cost.prox_cost.n_dims == [int(z[tag + "_n_dims"][ids.index(j)]) for j in graph[id_]]


def test_lowering_recognises_only_known_plugins(golden):
    z = golden("g5_dispatch")
    prob = problem_from(z, "uni5_")
    assert lowering.is_lowerable(prob)
    d = lowering.describe(prob)
    assert d["k"] == 5 and list(d["model"]) == [3] * 5 and d["w_prox"] == 200.0 and d["radius"] == 0.5
    np.testing.assert_array_equal(d["xf"], z["uni5_xf"])

    class MyUnicycle(dp.UnicycleDynamics4D):          # a user subclass with host code is NOT silently lowered
        def linearize(self, x, u):
            return np.eye(4), np.zeros((4, 2))

    dyn = dp.MultiDynamicalModel([MyUnicycle(0.1, 100 + i) for i in range(5)])
    assert not lowering.is_lowerable(dp.ilqrProblem(dyn, prob.game_cost))
    single = dp.ilqrProblem(dp.DoubleIntDynamics4D(0.1, 7), dp.ReferenceCost(np.zeros(4), np.eye(4), np.eye(2), id=7))
    assert lowering.is_lowerable(single) and lowering.describe(single)["k"] == 1


def test_ids_follow_the_reference_quirk():
    dp._reset_ids()
    a, b = dp.DoubleIntDynamics4D(0.1), dp.DoubleIntDynamics4D(0.1, 0)   # id=0 is falsy -> auto counter (quirk Q10)
    assert (a.id, b.id) == (0, 1) and dp.DoubleIntDynamics4D(0.1, 100).id == 100
    multi = dp.MultiDynamicalModel([a, b])
    assert multi.id == -1 and multi.n_x == 8 and multi.ids == [0, 1]


def test_shard_bounds_cover_everything_once():
    for n, world in ((1024, 8), (10, 3), (5, 8), (0, 4)):
        cuts = [shard_bounds(n, world, r) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n and all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        sizes = [hi - lo for lo, hi in cuts]
        assert max(sizes) - min(sizes) <= 1
    cost = np.array([1, 1, 1, 1, 100, 1, 1, 1], dtype=float)      # ragged costs: the heavy item gets a rank of its own
    cuts = [shard_bounds(8, 3, r, cost) for r in range(3)]
    assert cuts[0][0] == 0 and cuts[-1][1] == 8 and all(cuts[i][1] == cuts[i + 1][0] for i in range(2))


def test_result_rows_roundtrip():
    import torch
    B, T, n, m = 3, 4, 6, 2
    r = dict(X=torch.randn(B, T + 1, n, dtype=torch.float64), U=torch.randn(B, T, m, dtype=torch.float64),
             J=torch.randn(B, dtype=torch.float64), status=torch.tensor([1, 2, 3], dtype=torch.int32),
             n_bwd=torch.tensor([4, 5, 25], dtype=torch.int32), n_fwd=torch.tensor([7, 50, 250], dtype=torch.int32))
    back = unpack_results(pack_results(r), (T + 1, n), (T, m))
    for k in r:
        assert torch.equal(back[k], r[k]), k


def test_column_dominance_implies_no_row_exchange():
    """The sweeps skip the pivot search when every column of Q_uu passes |a_jj| (1 - 2^-20) > sum_{i != j} |a_ij|
    (riccati_mfma.hpp, S3).  The claim behind it -- LAPACK's partial pivoting then exchanges no row, at any step --
    checked on matrices that pass the kernel's test by the thinnest margin, with entries spread over twelve decades."""
    import scipy.linalg as sl
    rng = np.random.default_rng(0)
    tested = 0
    for _ in range(3000):
        m = int(rng.integers(2, 31))
        off = rng.normal(size=(m, m)) * 10.0 ** rng.uniform(-6, 6, size=(m, m))
        np.fill_diagonal(off, 0.0)
        colsum = np.abs(off).sum(0)
        A = off + np.diag(colsum * (1 + 2.0 ** -19) * rng.choice([-1.0, 1.0], size=m))
        dg = np.abs(np.diag(A))
        if not np.all(dg * (1 - 2.0 ** -20) > np.abs(A).sum(0) - dg):
            continue
        tested += 1
        _, piv = sl.lu_factor(A)
        assert np.array_equal(piv, np.arange(m))
    assert tested > 2500


def _parse_row(row):
    """A CSV row of solve_rhc's log -> the objects it was printed from (the subgraphs hold NumPy ints, quirk Q10)."""
    import ast
    import csv
    f = next(csv.reader([row]))
    num = lambda s: float(s) if ("." in s or "e" in s or "inf" in s or "nan" in s) else int(s)
    return dict(model_name=f[0], n_agents=int(f[1]), i_trial=ast.literal_eval(f[2]), centralized=f[3] == "True", last=f[4] == "True",
                t=num(f[5]), J=float(f[6]), N=int(f[7]), dt=float(f[8]), converged=f[9] == "True", ids=ast.literal_eval(f[10]),
                times=ast.literal_eval(f[11]), subgraphs=eval(f[12], {"np": np}), left=ast.literal_eval(f[13]))


@pytest.mark.parametrize("tag", ["rhc_c", "rhc_c2", "rhc_d"])
def test_rhc_log_row_format_is_the_references_byte_for_byte(golden, tag):
    """The CSV row of distributed.py:190-194,215-219: printing the reference's own values through this package's formatter
    reproduces the reference's rows byte for byte (the wall-clock `times` field included, as data)."""
    from dpilqr_amd.distributed import rhc_log_row
    z = golden("g7_callers")
    for row in z[tag + "_rows"]:
        f = _parse_row(str(row))
        assert rhc_log_row(f["model_name"], f["n_agents"], f["i_trial"], f["centralized"], f["last"], f["t"], f["J"], f["N"],
                           f["dt"], f["converged"], f["ids"], f["times"], f["subgraphs"], f["left"]) == str(row)


def test_graph_neighbours_are_numpy_ints_like_the_references(golden):
    """Quirk Q10: define_inter_graph_threshold returns the agent's own id as a Python int and its neighbours as NumPy
    ints; the difference is visible in the log rows, so it is kept."""
    z = golden("g5_dispatch"); tag = "uni5"
    prob = problem_from(z, tag + "_")
    g = dp.define_inter_graph_threshold(z[tag + "_x0"].reshape(1, -1), 0.5, prob.game_cost.x_dims, prob.ids)
    for id_, members in g.items():
        assert members == sorted(members) and id_ in members
        for v in members:
            assert isinstance(v, int) if v == id_ else isinstance(v, np.integer)


def test_descriptor_hints_of_a_batch():
    """The uniform_model word of include/dpilqr_hip.h as ProblemBatch.hint_word packs it from host arrays: model (bits 0-7), n_dims
    (8-15), shared weights (16), "DoubleIntDynamics4D / UnicycleDynamics4D agents only" (17: what lets the record-free wavefront
    sweep serve mixed planar clusters)."""
    from dpilqr_amd.batch import ProblemBatch
    Q = np.broadcast_to(np.eye(4), (3, 4, 4)); R = np.broadcast_to(np.eye(2), (3, 2, 2))
    w = ProblemBatch.hint_word([0, 0, 0], [2, 2, 2], Q, R, Q)
    assert w & 0xff == 1 and (w >> 8) & 0xff == 3 and (w >> 16) & 1 == 1 and (w >> 17) & 1 == 1
    w = ProblemBatch.hint_word([3, 0, 3], [2, 2, 2], Q, R, Q)            # unicycles and integrators mixed
    assert w & 0xff == 0 and (w >> 8) & 0xff == 3 and (w >> 17) & 1 == 1
    Qd = np.stack([np.eye(4) * (1 + i) for i in range(3)])
    w = ProblemBatch.hint_word([3, 3, 3], [2, 2, 2], Qd, R, Q)            # per-agent weights
    assert w & 0xff == 4 and (w >> 16) & 1 == 0 and (w >> 17) & 1 == 1
    Q6 = np.broadcast_to(np.eye(6), (2, 6, 6)); R3 = np.broadcast_to(np.eye(3), (2, 3, 3))
    w = ProblemBatch.hint_word([4, 1], [3, 2], Q6, R3, Q6)                # six-state agents, mixed n_dims
    assert w & 0xff == 0 and (w >> 8) & 0xff == 0 and (w >> 17) & 1 == 0


def test_solve_kwargs_names_what_it_drops():
    """Keyword arguments of the batched solve: n_lqr_iter, tol, t_kill pass (and `verbose` is the reference's print switch);
    anything else is named in a warning instead of being dropped silently (round-4 review: solve_rhc_scenarios filtered its
    kwargs to two names without a word, t_kill among the casualties)."""
    import warnings
    from dpilqr_amd.dispatch import solve_kwargs
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert solve_kwargs(dict(n_lqr_iter=3, tol=1e-4, t_kill=0.1, verbose=False)) == dict(n_lqr_iter=3, tol=1e-4, t_kill=0.1)
    with pytest.warns(UserWarning, match="n_iter"):
        assert solve_kwargs(dict(n_iter=3, tol=1e-4), "here") == dict(tol=1e-4)


def test_monte_carlo_trial_inputs_follow_the_references_stream():
    """dpilqr_amd/analysis.py: a trial consumes NumPy's global stream in the order scripts/analysis.py:45-54 and
    distributed.py:152 do -- random_setup, then the centralized branch's warm start, then the distributed branch's -- seeded per
    (model, team size, trial), distinct seeds for distinct cells."""
    from dpilqr_amd import analysis
    x0, xf, Uc, Ud = analysis.trial_inputs(4, 4, 8, 10, 10.0, 2, seed=12345)
    np.random.seed(12345)
    a, b = dp.random_setup(4, 4, is_rotation=False, rel_dist=4, var=2.0, n_d=2, random=True, energy=10.0)
    c = np.random.rand(10, 8) * 0.01; d = np.random.rand(10, 8) * 0.01
    assert np.array_equal(x0, a) and np.array_equal(xf, b) and np.array_equal(Uc, c) and np.array_equal(Ud, d)
    seeds = {analysis.seed_of(m, k, i) for m in analysis.MODELS for k in (3, 4, 5, 6, 7) for i in range(64)}
    assert len(seeds) == 3 * 5 * 64
    Q, R, Qf = analysis.weights_of(analysis.MODELS[2], 6)
    assert Q[0, 0] == 50.0 and R.shape == (3, 3) and Qf[5, 5] == 1000.0          # analysis.py:62-69
    assert analysis.HEADER.split(",")[:4] == ["dynamics", "n_agents", "trial", "centralized"]


def _g10_cells(z):
    from dpilqr_amd import analysis
    by_name = {m.__name__: m for m in analysis.MODELS}
    for tag in z["tags"]:
        name, n_agents, i_trial = str(tag).rsplit("_", 2)
        yield str(tag), by_name[name], int(n_agents), int(i_trial)


def test_harness_draws_are_the_references_own(golden):
    """G10: the REAL scripts/analysis.py::multi_agent_run, run on a seeded stream (tests/golden/make_golden.py g10), recorded what
    it drew: (x0, xf), then the centralized solve_rhc's warm start, then the distributed one's, captured from np.random.rand
    itself.  analysis.trial_inputs -- what the batched harness feeds its trials -- reproduces every draw bit for bit on the
    harness's own seed of the trial, and leaves the stream where the reference's whole trial left it (nothing else consumes
    it: analysis.py:35-107, distributed.py:106-221)."""
    from dpilqr_amd import analysis
    z = golden("g10_harness")
    N, energy, seed0 = int(z["N"]), float(z["energy"]), int(z["seed0"])
    n = 0
    for tag, model, n_agents, i_trial in _g10_cells(z):
        seed = analysis.seed_of(model, n_agents, i_trial, seed0)
        assert seed == int(z[tag + "_seed"]), tag
        n_states, n_u = model(-1).n_x, n_agents * model(-1).n_u
        n_d = 3 if model is analysis.QuadcopterDynamics6D else 2
        x0, xf, U_c, U_d = analysis.trial_inputs(n_agents, n_states, n_u, N, energy, n_d, seed)
        assert np.array_equal(x0, z[tag + "_x0"]) and np.array_equal(xf, z[tag + "_xf"]), tag
        assert np.array_equal(U_c, z[tag + "_U_c"]) and np.array_equal(U_d, z[tag + "_U_d"]), tag
        assert np.array_equal(np.random.get_state()[1][:4], z[tag + "_stream_after"]), tag
        # the problem the harness builds carries the reference's weights, ids and goal
        prob = analysis.build_problem(model, n_agents, float(z["dt"]), float(z["radius"]), xf, n_d)
        assert prob.ids == [100 + i for i in range(n_agents)] and np.array_equal(prob.game_cost.xf.ravel(), z[tag + "_xf"].ravel())
        rows = [_parse_row(str(r)) for r in z[tag + "_rows"]]
        assert {r["model_name"] for r in rows} == {model.__name__} and {r["i_trial"] for r in rows} == {i_trial}
        assert [r["centralized"] for r in rows if r["last"]] == [True, False]       # the centralized branch's rows, then the distributed one's
        n += 1
    assert n == 5
