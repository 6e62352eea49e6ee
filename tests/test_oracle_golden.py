"""Pins oracle/ (the CPU checker) against golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.golden_util import CFG2_SEEDS, MISC_SOLVES, MODEL_NAMES, PASS_CASES, cfg2_params, relerr

TOL_PASS = 1e-10   # single pass: summation order differs from BLAS, nothing else
TOL_SOLVE = 1e-6   # full solve: 1e-13 perturbations grow ~1e5..1e6x through the iterations (hard part 1)


def problem_from(z, prefix="", T=None):
    g = lambda k: z[prefix + k]
    return orc.Problem(g("model"), g("n_dims"), g("xf"), g("Q"), g("R"), g("Qf"), float(g("radius")), float(g("dt")),
                       int(g("T")) if T is None else int(T))


@pytest.mark.parametrize("name", MODEL_NAMES)
def test_models_vs_reference(golden, name):
    z = golden("g1_models"); m = int(z[f"{name}_enum"])
    for i in range(z[f"{name}_x"].shape[0]):
        x, u, dt = z[f"{name}_x"][i], z[f"{name}_u"][i], float(z[f"{name}_dt"][i])
        assert relerr(orc.model_f(m, x, u), z[f"{name}_f"][i]) < 1e-14
        assert relerr(orc.model_integrate(m, x, u, dt), z[f"{name}_xn"][i]) < 1e-13
        A, B = orc.model_linearize(m, x, u, dt)
        assert relerr(A, z[f"{name}_A"][i]) < 1e-14 and relerr(B, z[f"{name}_B"][i]) < 1e-14


@pytest.mark.parametrize("name", MODEL_NAMES)
def test_models_vs_compiled_reference(golden, name):
    """oracle/_ref = the reference's own bbdynamics.cpp compiled where it lies."""
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref not built (reference not mounted)")
    z = golden("g1_models"); m = int(z[f"{name}_enum"])
    for i in range(z[f"{name}_x"].shape[0]):
        x, u, dt = z[f"{name}_x"][i], z[f"{name}_u"][i], float(z[f"{name}_dt"][i])
        np.testing.assert_array_equal(orc.model_integrate(m, x, u, dt, ref=True), z[f"{name}_xn"][i])
        A, B = orc.model_linearize(m, x, u, dt, ref=True)
        np.testing.assert_array_equal(A, z[f"{name}_A"][i]); np.testing.assert_array_equal(B, z[f"{name}_B"][i])
        assert relerr(orc.model_integrate(m, x, u, dt), orc.model_integrate(m, x, u, dt, ref=True)) < 1e-13


def test_reference_own_double_int_test(golden):
    """Inputs of the reference's tests/test_dynamics.py:31-37 (x0, u, dt=0.5).  Its hand-written
    truth table is wrong as shipped (it forgets the a*dt^2/2 term: 1.25 vs 1.0, outside its own
    atol=0.1), so the expectation here is the reference MODEL's output on those inputs."""
    x = np.array([0.0, 2, 0, -2]); u = np.array([0.0, 2.0])
    for ref_row in golden("g1_models")["DoubleInt4D_reftest_traj"]:
        assert relerr(x, ref_row) < 1e-13
        x = orc.model_integrate(0, x, u, 0.5)


def test_reference_cost(golden):
    z = golden("g2_costs")
    p = orc.Problem([0], [2], z["ref_xf"], z["ref_Q"], z["ref_R"], z["ref_Qf"], 0.5, 0.1, 1)
    for i in range(4):
        for term, tag in ((False, "S"), (True, "T")):
            assert abs(p.cost(z["ref_x"][i], z["ref_u"][i], term) - z[f"ref_cost_{tag}"][i]) < 1e-12
            q = p.quadraticize(z["ref_x"][i], z["ref_u"][i], term)
            for got, nm in zip(q, ["Lx", "Lu", "Lxx", "Luu", "Lux"]):
                assert np.allclose(got, z[f"ref_{nm}_{tag}"][i], rtol=0, atol=1e-12), nm


def test_reference_own_cost_test(golden):
    """tests/test_cost.py:39-78 of the reference (the three tests that pass as shipped)."""
    z = golden("g2_costs"); x, u = z["reftest_x"], z["reftest_u"]
    # Car3D has 3 states / 2 controls like the reference test's ad-hoc cost
    p = orc.Problem([2], [2], np.zeros(3), np.eye(3), np.eye(2), np.diag([1.0, 1, 0]), 0.5, 0.1, 1)
    assert abs(p.cost(x, u) - (np.sum(x ** 2) + np.sum(u ** 2))) < 1e-12
    assert abs(p.cost(x, u, True) - np.sum(x[:-1] ** 2)) < 1e-12
    assert np.allclose([p.cost(x, u), p.cost(x, u, True)], z["reftest_cost"])
    Lx, Lu, Lxx, Luu, Lux = p.quadraticize(x, u)
    assert np.allclose(Lx, 2 * x) and np.allclose(Lu, 2 * u) and np.allclose(Lxx, 2 * np.eye(3))
    assert np.allclose(Luu, 2 * np.eye(2)) and np.allclose(Lux, 0)


def test_quadraticize_distance(golden):
    z = golden("g2_costs")
    for i in range(6):
        g, H = orc.quadraticize_distance(z["qd_pa"][i], z["qd_pb"][i], float(z["qd_radius"]), int(z["qd_nd"][i]))
        assert np.allclose(g, z["qd_Lx"][i], rtol=1e-13, atol=1e-14)
        assert np.allclose(H, z["qd_Lxx"][i], rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("tag,model", [("p2", 0), ("p3", 1), ("pm", 1)])
def test_proximity_cost(golden, tag, model):
    z = golden("g2_costs"); ns = orc.MODEL_DIMS[model][0]
    p = orc.Problem([model] * 3, z[f"{tag}_ndims"], np.zeros(3 * ns), np.eye(ns), np.eye(orc.MODEL_DIMS[model][1]),
                    np.eye(ns), 0.5, 0.1, 1)
    for i in range(4):
        assert abs(p.prox_cost(z[f"{tag}_x"][i]) - z[f"{tag}_cost"][i]) < 1e-13
        Lx, Lxx = p.prox_quadraticize(z[f"{tag}_x"][i])
        assert np.allclose(Lx, z[f"{tag}_Lx"][i], rtol=1e-12, atol=1e-13)
        assert np.allclose(Lxx, z[f"{tag}_Lxx"][i], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("k", [1, 3, 5])
def test_game_cost(golden, k):
    z = golden("g2_costs")
    p = orc.Problem([0] * k, [2] * k, z[f"gc{k}_xf"], np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4), 0.5, 0.1, 1)
    for i in range(3):
        for term, tag in ((False, "S"), (True, "T")):
            x, u = z[f"gc{k}_x"][i], z[f"gc{k}_u"][i]
            assert abs(p.cost(x, u, term) - z[f"gc{k}_cost_{tag}"][i]) < 1e-10 * max(1, abs(z[f"gc{k}_cost_{tag}"][i]))
            q = p.quadraticize(x, u, term)
            for got, nm in zip(q, ["Lx", "Lu", "Lxx", "Luu", "Lux"]):
                ref = z[f"gc{k}_{nm}_{tag}"][i]
                assert np.allclose(got, ref, rtol=1e-12, atol=1e-10), (nm, tag)


@pytest.mark.parametrize("case", PASS_CASES)
def test_passes(golden, case):
    z = golden(f"g3_passes_{case}")
    p = problem_from(z)
    X, J = p.rollout(z["x0"], z["U0"])
    assert relerr(X, z["X_roll"]) < 1e-12 and abs(J - z["J_roll"]) < 1e-11 * abs(z["J_roll"])
    # tiles (the plugin contract) at the operating point
    T = int(z["T"])
    for t in (0, T // 2, T - 1):
        A, B = p.linearize(z["X"][t], z["U"][t])
        assert relerr(A, z["tile_A"][t]) < 1e-14 and relerr(B, z["tile_B"][t]) < 1e-14
        q = p.quadraticize(z["X"][t], z["U"][t])
        for got, nm in zip(q, ["Lx", "Lu", "Lxx", "Luu", "Lux"]):
            assert np.allclose(got, z[f"tile_{nm}"][t], rtol=1e-11, atol=1e-9), nm
    K, d = p.backward_pass(z["X"], z["U"], float(z["mu"]))
    assert relerr(K, z["K"]) < TOL_PASS and relerr(d, z["d"]) < TOL_PASS
    K2, d2 = orc.backward_pass_tiles(z["tile_A"], z["tile_B"], z["tile_Lx"], z["tile_Lu"], z["tile_Lxx"],
                                     z["tile_Luu"], z["tile_Lux"], float(z["mu"]))
    assert relerr(K2, z["K"]) < TOL_PASS and relerr(d2, z["d"]) < TOL_PASS
    assert np.array_equal(orc.alphas(), z["alphas"])
    for a in range(10):
        Xn, Un, Jn = p.forward_pass(z["X"], z["U"], z["K"], z["d"], z["alphas"][a])
        assert relerr(Xn, z["X_fwd"][a]) < TOL_PASS and relerr(Un, z["U_fwd"][a]) < TOL_PASS
        assert abs(Jn - z["J_fwd"][a]) < 1e-10 * abs(z["J_fwd"][a])


def check_solve(r, z, pre):
    nb = len(z[pre + "mu_trace"])
    assert r["n_bwd"] == nb, "number of backward passes differs"
    tr = r["trace"]
    np.testing.assert_array_equal(tr[:, 0], z[pre + "mu_trace"])
    np.testing.assert_array_equal(tr[:, 1].astype(int), z[pre + "acc_trace"])   # decision trace
    np.testing.assert_array_equal(tr[:, 4].astype(int), z[pre + "nfwd_trace"])
    assert relerr(tr[:, 2], z[pre + "Jlast_trace"]) < TOL_SOLVE
    assert relerr(r["X"], z[pre + "X"]) < TOL_SOLVE and relerr(r["U"], z[pre + "U"]) < TOL_SOLVE
    assert abs(r["J"] - z[pre + "J"]) < TOL_SOLVE * abs(z[pre + "J"])


@pytest.mark.parametrize("seed", CFG2_SEEDS)
def test_solve_cfg2(golden, seed):
    z = golden("g4_solves_cfg2"); c = cfg2_params()
    p = orc.Problem(c["model"], c["n_dims"], z[f"s{seed}_xf"], c["Q"], c["R"], c["Qf"], c["radius"], c["dt"], c["T"])
    r = p.solve(z[f"s{seed}_x0"], np.zeros((50, 10)))
    check_solve(r, z, f"s{seed}_")
    last = int(z[f"s{seed}_acc_trace"][-1])
    assert r["status"] == (2 if last < 0 else 1)


@pytest.mark.parametrize("tag", MISC_SOLVES)
def test_solve_misc(golden, tag):
    z = golden("g4_solves_misc")
    p = problem_from(z, tag + "_")
    r = p.solve(z[tag + "_x0"], z[tag + "_U0"])
    check_solve(r, z, tag + "_")


@pytest.mark.parametrize("tag", ["uni5", "quad10", "uni8"])
def test_dispatch(golden, tag):
    z = golden("g5_dispatch")
    p = problem_from(z, tag + "_")
    k, ns = p.k, p.n_s
    g = orc.define_inter_graph_threshold(z[tag + "_x0"].reshape(1, -1), 0.5, k, ns)
    adj = np.zeros((k, k), dtype=np.int32)
    for i, nb in g.items():
        adj[i, nb] = 1
    np.testing.assert_array_equal(adj, z[tag + "_adj_x0"])
    g2 = orc.define_inter_graph_threshold(z[tag + "_X_dec"], 0.5, k, ns)
    adj2 = np.zeros((k, k), dtype=np.int32)
    for i, nb in g2.items():
        adj2[i, nb] = 1
    np.testing.assert_array_equal(adj2, z[tag + "_adj_traj"])
    if tag == "quad10":
        return  # the solve itself is covered by the two unicycle cases (keeps the CPU suite short)
    Xd, Ud, Jf, _ = orc.solve_distributed(p, z[tag + "_x0"].reshape(1, -1), z[tag + "_U0"], 0.5)
    assert relerr(Xd, z[tag + "_X_dec"]) < TOL_SOLVE and relerr(Ud, z[tag + "_U_dec"]) < TOL_SOLVE
    assert abs(Jf - z[tag + "_J_full"]) < TOL_SOLVE * abs(z[tag + "_J_full"])


def test_solve_batch_matches_single(golden):
    z = golden("g4_solves_cfg2"); c = cfg2_params()
    seeds = [0, 17, 2]
    x0 = np.array([z[f"s{s}_x0"] for s in seeds]); xf = np.array([z[f"s{s}_xf"] for s in seeds])
    proto = orc.Problem(c["model"], c["n_dims"], xf[0], c["Q"], c["R"], c["Qf"], c["radius"], c["dt"], c["T"])
    r = orc.solve_batch(proto, x0, xf, np.zeros((3, 50, 10)), n_threads=2)
    for i, s in enumerate(seeds):
        assert relerr(r["X"][i], z[f"s{s}_X"]) < TOL_SOLVE
        assert r["n_bwd"][i] == len(z[f"s{s}_mu_trace"])


# ---------------------------------------------------------------- G8: the padded human model of BASELINE config 5
def test_padded_human_model_vs_reference_shim(golden):
    z = golden("g8_hetero_model")
    for i in range(len(z["HumanPad12D_x"])):
        x, u, dt = z["HumanPad12D_x"][i], z["HumanPad12D_u"][i], float(z["HumanPad12D_dt"][i])
        assert relerr(orc.model_f(8, x, u), z["HumanPad12D_f"][i]) < 1e-15
        np.testing.assert_array_equal(orc.model_integrate(8, x, u, dt)[6:], x[6:])       # the padding never moves
        assert relerr(orc.model_integrate(8, x, u, dt), z["HumanPad12D_integrate"][i]) < 1e-14
        A, B = orc.model_linearize(8, x, u, dt)
        assert relerr(A, z["HumanPad12D_A"][i]) < 1e-15 and relerr(B, z["HumanPad12D_B"][i]) < 1e-15


def test_hetero_k3_passes_and_solve(golden):
    z = golden("g8_hetero_k3_passes")
    p = problem_from(z)
    X, J = p.rollout(z["x0"], z["U0"])
    assert relerr(X, z["X_roll"]) < 1e-12 and abs(J - z["J_roll"]) < 1e-11 * abs(z["J_roll"])
    K, d = p.backward_pass(z["X"], z["U"], float(z["mu"]))
    assert relerr(K, z["K"]) < TOL_PASS and relerr(d, z["d"]) < TOL_PASS
    for a in range(10):
        Xn, Un, Jn = p.forward_pass(z["X"], z["U"], z["K"], z["d"], z["alphas"][a])
        assert relerr(Xn, z["X_fwd"][a]) < TOL_PASS and abs(Jn - z["J_fwd"][a]) < 1e-10 * abs(z["J_fwd"][a])
    z = golden("g8_hetero_k3_solve")
    p = problem_from(z)
    r = p.solve(z["x0"], z["U0"], n_lqr_iter=12)
    check_solve(r, z, "")


def test_hetero_k20_cfg5_size_pass(golden):
    """cfg5 at its stated size (20 agents, n_x = 240, n_u = 80, T = 150): one backward pass and the ten forward passes."""
    z = golden("g8_hetero_k20")
    p = problem_from(z)
    X, J = p.rollout(z["x0"], z["U0"])
    assert relerr(X[::10], z["X_roll_every10"]) < 1e-11 and abs(J - z["J_roll"]) < 1e-11 * abs(z["J_roll"])
    K, d = p.backward_pass(z["X"], z["U"], float(z["mu"]))
    assert relerr(K[z["K_steps"]], z["K_kept"]) < 1e-7 and relerr(d, z["d"]) < 1e-7
    acc = int(z["acc"])
    for a in range(10):
        Xn, Un, Jn = p.forward_pass(z["X"], z["U"], K, d, z["alphas"][a])
        if np.isnan(z["J_fwd"][a]):
            assert not (Jn < float(z["J_star"]))                      # rejected either way (control.py:183)
        else:
            assert abs(Jn - z["J_fwd"][a]) < 1e-6 * abs(z["J_fwd"][a])
        if a == acc:
            assert relerr(Xn[::10], z["X_fwd_acc_every10"]) < 1e-6 and relerr(Un[::10], z["U_fwd_acc_every10"]) < 1e-6


# ---------------------------------------------------------------- G7: the callers either side of the solve
def parse_row(row):
    import ast, csv
    f = next(csv.reader([row]))
    return dict(model=f[0], n_agents=int(f[1]), i_trial=f[2], centralized=f[3], last=f[4], t=float(f[5]), J=float(f[6]),
                N=int(f[7]), dt=float(f[8]), converged=f[9], ids=f[10], times=f[11],
                subgraphs=eval(f[12], {"np": np}), left=ast.literal_eval(f[13]))


@pytest.mark.parametrize("tag", ["rhc_c", "rhc_c2", "rhc_d"])
def test_rhc_vs_reference(golden, tag):
    z = golden("g7_callers")
    p = problem_from(z, tag + "_", T=z[tag + "_N"])
    kw = dict(step_size=int(z[tag + "_kw_step_size"]), dist_converge=float(z[tag + "_kw_dist_converge"]),
              t_diverge=float(z[tag + "_kw_t_diverge"]))
    Xf, Uf, Jf, rounds, conv = orc.solve_rhc(p, z[tag + "_x0"], int(z[tag + "_N"]), z[tag + "_U_warm"], radius=0.5,
                                             centralized=bool(z[tag + "_centralized"]), **kw)
    assert Xf.shape == z[tag + "_X_full"].shape
    assert relerr(Xf, z[tag + "_X_full"]) < TOL_SOLVE and relerr(Uf, z[tag + "_U_full"]) < TOL_SOLVE
    assert abs(Jf - z[tag + "_J_full"]) < TOL_SOLVE * abs(z[tag + "_J_full"])
    rows = [parse_row(r) for r in z[tag + "_rows"]]
    assert len(rows) == len(rounds) + 1
    ids = [int(i) for i in z[tag + "_ids"]]
    for row, (t, J, c, graphs, left) in zip(rows, rounds):
        assert abs(row["t"] - t) < 1e-12 and abs(row["J"] - J) < TOL_SOLVE * abs(J) and row["converged"] == str(c)
        assert [[int(v) for v in g] for g in row["subgraphs"]] == [[ids[i] for i in g] for g in graphs]
        assert np.allclose(row["left"], left, rtol=1e-6, atol=1e-9)
    assert rows[-1]["converged"] == str(conv) and rows[-1]["last"] == "True"


@pytest.mark.parametrize("tag", ["ws_uni", "ws_di"])
def test_selfish_warmstart_vs_reference(golden, tag):
    z = golden("g7_callers")
    p = problem_from(z, tag + "_", T=z[tag + "_N"])
    Uw = orc.selfish_warmstart(p, z[tag + "_x0"], int(z[tag + "_N"]))
    assert relerr(Uw, z[tag + "_U_warm"]) < TOL_SOLVE


def test_solve_subproblem_vs_reference(golden):
    z = golden("g7_callers")
    p = problem_from(z, "sub_")
    k, ns, nc, T = p.k, p.n_s, p.n_c, int(z["sub_T"])
    for i in range(k):
        idx = [int(j) for j in np.nonzero(z["sub_adj"][i])[0]]
        sub = p.with_T(T).subproblem(idx)
        r = sub.solve(np.concatenate([z["sub_x0"][a * ns:(a + 1) * ns] for a in idx]), np.zeros((T, nc * len(idx))))
        pos = idx.index(i)
        assert relerr(r["X"][:, pos * ns:(pos + 1) * ns], z[f"sub_X_{i}"]) < TOL_SOLVE
        assert relerr(r["U"][:, pos * nc:(pos + 1) * nc], z[f"sub_U_{i}"]) < TOL_SOLVE


# ---------------------------------------------------------------- the NumPy restatement (bench.py's second CPU baseline)
@pytest.mark.parametrize("seed", [0, 17, 2])
def test_numpy_port_solve_cfg2(golden, seed):
    """oracle/numpy_port.py (the reference's per-step Python structure, restated) against the reference's own solves:
    decision trace exactly, trajectory to 1e-5."""
    from oracle import numpy_port
    z = golden("g4_solves_cfg2"); c = cfg2_params()
    s = numpy_port.cfg_solver(c["model"], c["n_dims"], z[f"s{seed}_xf"], c["Q"], c["R"], c["Qf"], c["radius"], c["dt"], c["T"])
    r = s.solve(z[f"s{seed}_x0"], np.zeros((50, 10)))
    pre = f"s{seed}_"
    np.testing.assert_array_equal(r["trace"][:, 0], z[pre + "mu_trace"])
    np.testing.assert_array_equal(r["trace"][:, 1].astype(int), z[pre + "acc_trace"])
    assert relerr(r["X"], z[pre + "X"]) < TOL_SOLVE and relerr(r["U"], z[pre + "U"]) < TOL_SOLVE
    assert abs(r["J"] - z[pre + "J"]) < TOL_SOLVE * abs(z[pre + "J"])


@pytest.mark.parametrize("tag", ["uni_k3", "quad_k3", "mixed"])
def test_numpy_port_solve_misc(golden, tag):
    from oracle import numpy_port
    z = golden("g4_solves_misc")
    g = lambda k: z[tag + "_" + k]
    s = numpy_port.cfg_solver(g("model"), g("n_dims"), g("xf"), g("Q"), g("R"), g("Qf"), float(g("radius")), float(g("dt")), int(g("T")))
    r = s.solve(g("x0"), g("U0"))
    np.testing.assert_array_equal(r["trace"][:, 1].astype(int), g("acc_trace"))
    assert relerr(r["X"], g("X")) < TOL_SOLVE and relerr(r["U"], g("U")) < TOL_SOLVE


# ---------------------------------------------------------------- G9: does the reference determine its own result?
@pytest.mark.parametrize("fam", ["cfg2", "uni8", "quad10"])
def test_g9_the_reference_does_not_determine_its_own_result_where_the_gpu_flips(golden, fam):
    """The fixture's own content, as a fact about the REFERENCE: on every item the GPU decided differently from the oracle,
    the reference's 33 runs (x0 and 32 copies perturbed by at most 5e-13) do not agree on one decision trace -- typically
    they take some thirty different ones -- while on the control items they take exactly one."""
    from tests import chaos_util as cu
    z = golden(f"g9_chaos_{fam}")
    assert np.allclose(np.sort(np.abs(z["deltas"][1:])), np.sort(np.abs(np.array(__import__("oracle.parity", fromlist=["DELTAS"]).DELTAS))))
    flipped = z["flipped_on_gpu"].astype(bool)
    n_traces = np.array([len(set(cu.member_traces(z, a))) for a in range(len(flipped))])
    assert (n_traces[flipped] >= 2).all(), n_traces
    assert np.median(n_traces[flipped]) >= 10
    assert (n_traces[~flipped] == 1).sum() >= 1
    spread = z["dX_vs_base"].max(axis=1)
    assert np.median(spread[flipped]) > 1e-3          # the members' final trajectories differ in the leading digits


@pytest.mark.parametrize("fam", ["cfg2", "uni8", "quad10"])
def test_g9_oracle_against_the_reference_ensemble(golden, fam):
    """The C oracle on the same items (from the unperturbed x0): while all 33 reference members still agree on a decision
    the oracle takes it too; afterwards each of its decisions is one an alive member takes, or one the alive members do not
    agree on among themselves -- at most 1 % are taken against a unanimous group of >= 8 members (unanimity of n samples
    bounds the odds of another outcome only by ~3/n).  On items where the reference takes ONE trace the oracle reproduces
    it, the iteration costs and the final trajectory."""
    from tests import chaos_util as cu
    z = golden(f"g9_chaos_{fam}")
    model, n_dims, x0, xf, U0, Q, R, Qf, T = cu.problem_inputs(z)
    proto = orc.Problem(model, n_dims, xf[0], Q, R, Qf, 0.5, 0.1, T)
    o = orc.solve_batch(proto, x0, xf, U0, trace=True)
    total = dict(witnessed=0, undetermined=0, violation=0, inconclusive=0, exhausted=0)
    inside = 0
    for a in range(len(z["seeds"])):
        members = cu.member_traces(z, a)
        mine = cu.trace_of(o["n_bwd"][a], np.nan_to_num(o["trace"][a, :, 1], nan=-9))
        upto = cu.unanimous_prefix(members)
        assert mine[:upto] == members[0][:upto], (fam, int(z["seeds"][a]), upto)
        cat, _ = cu.classify(mine, members)
        for k_, v in cat.items():
            total[k_] += v
        lo, hi = z["J"][a].min(), z["J"][a].max()
        inside += bool(lo - 1e-9 * abs(lo) - (hi - lo) <= o["J"][a] <= hi + 1e-9 * abs(hi) + (hi - lo))
        if len(set(members)) == 1 and z["dX_vs_base"][a].max() < 1e-7:      # the reference determines this item
            nb = int(z["n_bwd"][a, 0])
            assert mine == members[0]
            assert np.allclose(o["trace"][a, :nb, 3], z["Jstar_trace"][a, 0, :nb], rtol=1e-6)      # stored as float32
            assert relerr(o["X"][a], z["X_base"][a]) < TOL_SOLVE
    n_dec = sum(total.values())
    assert total["violation"] <= max(1, n_dec // 100), total
    assert total["witnessed"] >= 0.8 * n_dec, total
    assert inside == len(z["seeds"]), inside            # the returned cost: within the reference ensemble's range, widened by its width
