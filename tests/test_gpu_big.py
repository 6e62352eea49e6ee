"""GPU parity of the large-cluster path (n_x > 60: BASELINE config 5), the padded human model, the fp32 arm and the
enqueue-only solve.  Run on the MI355X box with `-m gpu`.

Tolerances: fp64 single passes 1e-9 (1e-7 at n_x = 240, where cond(Q_uu) ~ 1e9 with the humans' R = 1e-9 entries),
whole solves 1e-5 with the decision trace required to match; fp32 is compared with fp64 at the tolerances written in
each test (they are what the tolerance study, scripts/fp32_study.py, measures)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from tests.golden_util import relerr

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

TOL_PASS = 1e-9
TOL_SOLVE = 1e-5


@pytest.fixture(scope="module")
def dp():
    import dpilqr_amd
    from dpilqr_amd import _lib
    _lib.require_gpu()
    return dpilqr_amd


def batch_from(dp, z, prefix="", B=1):
    g = lambda k: z[prefix + k]
    return dp.ProblemBatch(g("model"), g("n_dims"), np.tile(g("xf"), (B, 1)), g("Q"), g("R"), g("Qf"), float(g("radius")),
                           float(g("dt")), int(g("T")))


def oracle_problem(z, prefix=""):
    from oracle import oracle as orc
    g = lambda k: z[prefix + k]
    return orc.Problem(g("model"), g("n_dims"), g("xf"), g("Q"), g("R"), g("Qf"), float(g("radius")), float(g("dt")), int(g("T")))


def test_padded_human_model_ffi(dp, golden):
    """Model 8 (HumanDynamics6D zero-padded to 12 / 4) through the model FFI against the shim built from reference calls."""
    import torch
    from dpilqr_amd import _lib
    from dpilqr_amd.device import empty, ptr, stream_handle, to_dev
    z = golden("g8_hetero_model"); lib = _lib.load()
    x, u, dts = z["HumanPad12D_x"], z["HumanPad12D_u"], z["HumanPad12D_dt"]
    n = x.shape[0]
    model = to_dev(np.full(n, 8), torch.int32); xd, ud = to_dev(x), to_dev(u)
    f = empty((n, 12)); _lib.check(lib.dpilqr_model_f(n, 12, ptr(model), ptr(xd), ptr(ud), ptr(f), stream_handle()))
    assert relerr(f.cpu().numpy(), z["HumanPad12D_f"]) < 1e-13
    for dt in (0.05, 0.1):
        sel = np.where(dts == dt)[0]
        xn = empty((n, 12)); A = empty((n, 12, 12)); Bm = empty((n, 12, 4))
        _lib.check(lib.dpilqr_model_integrate(n, 12, ptr(model), ptr(xd), ptr(ud), dt, ptr(xn), stream_handle()))
        _lib.check(lib.dpilqr_model_linearize(n, 12, ptr(model), ptr(xd), ptr(ud), dt, ptr(A), ptr(Bm), stream_handle()))
        assert relerr(xn.cpu().numpy()[sel], z["HumanPad12D_integrate"][sel]) < 1e-12
        np.testing.assert_array_equal(xn.cpu().numpy()[:, 6:], x[:, 6:])            # the padding never moves
        assert relerr(A.cpu().numpy()[sel], z["HumanPad12D_A"][sel]) < 1e-13
        assert relerr(Bm.cpu().numpy()[sel], z["HumanPad12D_B"][sel]) < 1e-13
    m = dp.HumanDynamics6DPadded12(0.1, 77)
    assert (m.n_x, m.n_u, m.model.value) == (12, 4, 8)
    i1 = int(np.where(dts == 0.1)[0][0])
    assert relerr(m(x[i1], u[i1]), z["HumanPad12D_integrate"][i1]) < 1e-12                        # the plugin surface


def test_hetero_k3_vs_reference(dp, golden):
    """Quadcopter12D + Quadcopter12D + padded human (n_x = 36: the workgroup sweep and the generic forward pass)."""
    z = golden("g8_hetero_k3_passes")
    pb = batch_from(dp, z)
    X, J = pb.rollout(z["x0"][None], z["U0"][None])
    assert relerr(X[0].cpu().numpy(), z["X_roll"]) < 1e-10 and abs(float(J[0]) - z["J_roll"]) < 1e-10 * abs(z["J_roll"])
    K, d = pb.backward_pass(z["X"][None], z["U"][None], float(z["mu"]))
    assert relerr(K[0].cpu().numpy(), z["K"]) < TOL_PASS and relerr(d[0].cpu().numpy(), z["d"]) < TOL_PASS
    Xn, Un, Jn = pb.forward_pass(z["X"][None], z["U"][None], z["K"][None], z["d"][None], z["alphas"])
    assert relerr(Xn[0].cpu().numpy(), z["X_fwd"]) < TOL_PASS and relerr(Jn[0].cpu().numpy(), z["J_fwd"]) < TOL_PASS
    z = golden("g8_hetero_k3_solve")
    pb = batch_from(dp, z)
    r = pb.solve(z["x0"][None], z["U0"][None], n_lqr_iter=12, trace=True)
    nb = len(z["mu_trace"]); tr = r["trace"][0].cpu().numpy()[:nb]
    assert int(r["n_bwd"][0]) == nb
    np.testing.assert_array_equal(tr[:, 0], z["mu_trace"]); np.testing.assert_array_equal(tr[:, 1].astype(int), z["acc_trace"])
    assert relerr(r["X"][0].cpu().numpy(), z["X"]) < TOL_SOLVE and relerr(r["U"][0].cpu().numpy(), z["U"]) < TOL_SOLVE


def test_cfg5_size_pass_vs_reference(dp, golden):
    """BASELINE config 5 at its stated size: 14 Quadcopter12D + 6 padded humans, n_x = 240, n_u = 80, T = 150.
    One backward pass and the ten forward passes against the reference's own numbers (G8)."""
    z = golden("g8_hetero_k20")
    pb = batch_from(dp, z)
    assert pb.fused_sweep and (pb.n_x, pb.n_u, pb.T) == (240, 80, 150)
    X, J = pb.rollout(z["x0"][None], z["U0"][None])
    assert relerr(X[0].cpu().numpy()[::10], z["X_roll_every10"]) < 1e-9 and abs(float(J[0]) - z["J_roll"]) < 1e-9 * abs(z["J_roll"])
    K, d = pb.backward_pass(z["X"][None], z["U"][None], float(z["mu"]))
    Kh, dh = K[0].cpu().numpy(), d[0].cpu().numpy()
    assert np.isfinite(Kh).all() and np.isfinite(dh).all()
    assert relerr(Kh[z["K_steps"]], z["K_kept"]) < 1e-7 and relerr(dh, z["d"]) < 1e-7
    Xn, Un, Jn = pb.forward_pass(z["X"][None], z["U"][None], K, d, z["alphas"])
    Jn = Jn[0].cpu().numpy(); acc = int(z["acc"]); J_star = float(z["J_star"])
    for a in range(10):
        if np.isnan(z["J_fwd"][a]):
            assert not (Jn[a] < J_star)                       # rejected either way (control.py:183)
        else:
            assert abs(Jn[a] - z["J_fwd"][a]) < 1e-6 * abs(z["J_fwd"][a])
    assert relerr(Xn[0, acc].cpu().numpy()[::10], z["X_fwd_acc_every10"]) < 1e-6
    assert relerr(Un[0, acc].cpu().numpy()[::10], z["U_fwd_acc_every10"]) < 1e-6
    # the oracle on the GPU's own gains: the forward pass alone
    p = oracle_problem(z)
    Xo, Uo, Jo = p.forward_pass(z["X"], z["U"], Kh, dh, z["alphas"][acc])
    assert relerr(Xn[0, acc].cpu().numpy(), Xo) < 1e-9 and abs(Jn[acc] - Jo) < 1e-9 * abs(Jo)


def test_cfg5_quad12_homogeneous_pass_vs_oracle(dp):
    """20 x Quadcopter12D, T = 150 (the configuration the reference supports as stated): one backward + forward pass."""
    from oracle import oracle as orc
    from dpilqr_amd.util import random_setup
    k, T = 20, 150
    np.random.seed(77)
    a, b = random_setup(k, 12, is_rotation=False, rel_dist=k, var=k / 2, n_d=3, random=True, energy=100.0)
    x0, xf = a.ravel(), b.ravel()
    Q, R, Qf = np.eye(12), np.eye(4), 1000.0 * np.eye(12)
    U0 = np.zeros((T, 4 * k)); U0[:, 3::4] = 9.80665 * 63.0 / 2000.0
    pb = dp.ProblemBatch([7] * k, [3] * k, xf[None], Q, R, Qf, 0.5, 0.1, T)
    p = orc.Problem([7] * k, [3] * k, xf, Q, R, Qf, 0.5, 0.1, T)
    X, J = pb.rollout(x0[None], U0[None])
    Xo, Jo = p.rollout(x0, U0)
    assert relerr(X[0].cpu().numpy(), Xo) < 1e-9 and abs(float(J[0]) - Jo) < 1e-9 * abs(Jo)
    K, d = pb.backward_pass(Xo[None], U0[None], 1.0)
    Ko, do = p.backward_pass(Xo, U0, 1.0)
    assert relerr(K[0].cpu().numpy(), Ko) < 1e-8 and relerr(d[0].cpu().numpy(), do) < 1e-8
    al = orc.alphas()
    Xn, Un, Jn = pb.forward_pass(Xo[None], U0[None], Ko[None], do[None], al)
    for ai in (9, 6, 3):
        Xr, Ur, Jr = p.forward_pass(Xo, U0, Ko, do, al[ai])
        if np.isfinite(Jr):
            assert relerr(Xn[0, ai].cpu().numpy(), Xr) < 1e-8 and abs(float(Jn[0, ai]) - Jr) < 1e-8 * abs(Jr)


def _cfg5_batch(hetero, seeds, energy=100.0):
    """BASELINE config 5 inputs: 20 agents, T = 150; hetero: 14 Quadcopter12D + 6 zero-padded humans (the fixture G8's weights),
    else 20 x Quadcopter12D, the configuration the reference supports as stated (hover warm start)."""
    from dpilqr_amd.util import random_setup
    k, T = 20, 150
    models = [7] * 14 + [8] * 6 if hetero else [7] * 20
    nd = [3] * 14 + [2] * 6 if hetero else [3] * 20
    x0 = np.zeros((len(seeds), 240)); xf = np.zeros((len(seeds), 240))
    for j, s in enumerate(seeds):
        np.random.seed(s)
        a, b = random_setup(k, 12, is_rotation=False, rel_dist=k, var=k / 2, n_d=3, random=True, energy=energy)
        x0[j], xf[j] = a.ravel(), b.ravel()
    if hetero:
        Q = np.stack([np.eye(12)] * 14 + [np.diag([1.0, 1, 1, 0, 0, 0] + [0.0] * 6)] * 6)
        R = np.stack([np.eye(4)] * 14 + [np.diag([1.0, 1, 1e-9, 1e-9])] * 6)
    else:
        Q = np.stack([np.eye(12)] * 20); R = np.stack([np.eye(4)] * 20)
    Qf = np.stack([1000.0 * np.eye(12)] * k)
    U0 = np.zeros((len(seeds), T, 80))
    for i in range(k):
        if models[i] == 7:
            U0[:, :, 4 * i + 3] = 9.80665 * 63.0 / 2000.0
    return models, nd, x0, xf, Q, R, Qf, U0, T


@pytest.mark.parametrize("hetero,seeds", [(False, (6000, 6002, 6005, 6007)), (True, (6001, 6006, 6007, 6000))])
def test_cfg5_whole_solves_at_the_stated_size(dp, hetero, seeds):
    """BASELINE config 5 at its stated size -- 20 twelve-state agents, n_x = 240, n_u = 80, T = 150 -- as WHOLE solves (up
    to six iLQR iterations; the seeds were picked for what the oracle does with them: six full iterations, convergence at
    the fifth, line-search failures after two and three, NaN-cost candidates from tan() on the way): the large-cluster
    sweep, the KDIRECT forward pass and the device-side decision logic together, every item held to the ensemble
    envelope of oracle/parity.py (four members: a 240-state backward pass costs the oracle a second or two), and the
    items on which the reference's own ensemble stays within 1e-6 to the north star's 1e-5 with the oracle's own
    decision trace."""
    from oracle import oracle as orc, parity
    models, nd, x0, xf, Q, R, Qf, U0, T = _cfg5_batch(hetero, seeds)
    pb = dp.ProblemBatch(models, nd, xf, Q, R, Qf, 0.5, 0.1, T)
    assert pb.fused_sweep and (pb.n_x, pb.n_u, pb.T) == (240, 80, 150)
    r = {k: v.cpu().numpy() for k, v in pb.solve(x0, U0, n_lqr_iter=6, trace=True).items()}
    proto = orc.Problem(models, nd, xf[0], Q, R, Qf, 0.5, 0.1, T)
    o = orc.solve_batch(proto, x0, xf, U0, n_lqr_iter=6, trace=True)
    rep = parity.envelope(r, proto, x0, xf, U0, n_lqr_iter=6, natural=o, deltas=(1e-13, -1e-13, 5e-13, -5e-13))
    sm = rep["summary"]
    print(sm, "n_bwd", r["n_bwd"], "oracle", o["n_bwd"], "status", r["status"])
    assert sm["all_ok"], (sm, [w for w in rep["why"] if w])
    # (at this size half of the first iteration's candidates overflow to NaN in the reference itself, DESIGN.md section 5: a member
    # of the ensemble that is forced through such a candidate draws no bound for its item -- at most one of the four here)
    assert sm["unchecked_frac"] <= 0.25, sm
    assert r["n_bwd"].max() >= 5 and np.isfinite(r["X"]).all()
    tight = ~rep["flipped"] & (rep["spreadX"] < 1e-6)
    assert tight.sum() >= 2, (sm, rep["spreadX"])
    for i in np.where(tight)[0]:
        n = o["n_bwd"][i]
        assert (r["n_bwd"][i], r["n_fwd"][i], r["status"][i]) == (n, o["n_fwd"][i], o["status"][i]), i
        np.testing.assert_array_equal(r["trace"][i, :n, 0], o["trace"][i, :n, 0])                  # mu
        np.testing.assert_array_equal(r["trace"][i, :n, 1], o["trace"][i, :n, 1])                  # accepted alpha
        assert relerr(r["X"][i], o["X"][i]) < TOL_SOLVE and relerr(r["U"][i], o["U"][i]) < TOL_SOLVE, i
        assert abs(r["J"][i] - o["J"][i]) < TOL_SOLVE * abs(o["J"][i]) or (np.isnan(r["J"][i]) and np.isnan(o["J"][i]))


def test_fp32_tolerance_study_bounds(dp):
    """BASELINE config 5's "fp32 vs fp64 tolerance study" as assertions (the table itself: scripts/fp32_study.py,
    profiles/).  Every figure is fp32-GPU against fp64-GPU on the same items, with fp64-GPU against the fp64 oracle
    beside it as the noise floor of the comparison.  What the study found and this test holds:
      * a single fp32 pass is good to 1e-6..1e-5 on well-conditioned items (cfg2: median gain error < 1e-5, median
        forward-pass state error < 1e-5) but not uniformly (the worst cfg2 gain is off by O(1));
      * the decision logic turns that into different iterates: fp32 flips decisions on 10..40 % of cfg2 solves, an order
        of magnitude more often than fp64 does against the oracle, and fewer than 80 % of the fp32 solves end within
        1e-5 of the fp64 ones -- fp32 cannot meet the north star's bar;
      * at config 5's size and conditioning (R = 1e-9 entries, cond(Q_uu) ~ 1e9) the fp32 gains are wrong in the second
        digit (median > 1e-4) although a forward pass on given gains still tracks to 1e-3."""
    sys.path.insert(0, str(ROOT / "scripts"))
    import fp32_study
    c2 = fp32_study.study("cfg2", [0] * 5, [2] * 5, 50, 512, 0, 10.0, 50, 128)
    p, s, so = c2["pass"], c2["solve_fp32_vs_fp64"], c2["solve_fp64_vs_oracle"]
    assert p["rollout_X"] < 1e-5 and p["K_median"] < 1e-5 and p["forward_X_median"] < 1e-5 and p["K_max"] > 1e-3, p
    assert 0.10 < s["decision_flip_rate"] < 0.40 and s["frac_within_1e5"] < 0.80, s
    assert so["decision_flip_rate"] < 0.05 and s["decision_flip_rate"] > 3 * so["decision_flip_rate"], (s, so)
    assert so["frac_within_1e5"] > 0.93, so
    c5 = fp32_study.study("cfg5", [7] * 14 + [8] * 6, [3] * 14 + [2] * 6, 150, 4, 6000, 100.0, 6, 0)
    p5, s5 = c5["pass"], c5["solve_fp32_vs_fp64"]
    assert 1e-4 < p5["K_median"] < 10.0 and p5["forward_X_median"] < 1e-3, p5
    assert s5["frac_within_1e5"] <= 0.5, s5


@pytest.mark.parametrize("model,k,T", [(3, 16, 20), (0, 18, 12), (4, 11, 15), (7, 6, 8), (1, 12, 10)])
def test_large_cluster_passes_vs_oracle(dp, model, k, T):
    """n_x just beyond the workgroup sweep (64 .. 72) for every state-dimension family: rollout, backward and forward
    pass of the large-cluster kernels against the oracle, several items per launch."""
    from oracle import oracle as orc
    from dpilqr_amd.device import to_dev
    ns, nc = {0: (4, 2), 3: (4, 2), 4: (6, 3), 1: (6, 3), 7: (12, 4)}[model]
    nd = 3 if ns >= 6 else 2
    B = 3
    rng = np.random.default_rng(4000 + model * 31 + k)
    xf = rng.normal(size=(B, k * ns)) * 2.0; x0 = rng.normal(size=(B, k * ns)) * 2.0
    x0.reshape(B, k, ns)[:, :, nd:] *= 0.1; xf.reshape(B, k, ns)[:, :, nd:] = 0.0
    U0 = rng.normal(size=(B, T, k * nc)) * 0.05
    if model == 4:
        U0[:, :, 0::3] += 9.80665
    if model == 7:
        U0 = U0 * 1e-4; U0[:, :, 3::4] += 9.80665 * 63.0 / 2000.0
        x0.reshape(B, k, ns)[:, :, 3:] *= 0.02
    Q = np.eye(ns) * 1.3; R = np.eye(nc); Qf = 100.0 * np.eye(ns)
    pb = dp.ProblemBatch([model] * k, [nd] * k, xf, Q, R, Qf, 0.7, 0.1, T)
    assert pb.fused_sweep
    X, J = pb.rollout(x0, U0)
    mu = rng.uniform(0, 1, size=B)
    K, d = pb.backward_pass(X, U0, to_dev(mu))
    al = orc.alphas()
    Xn, Un, Jn = pb.forward_pass(X, U0, K, d, al)
    tol_roll = 1e-7 if model == 7 else 1e-11
    for i in range(B):
        p = orc.Problem([model] * k, [nd] * k, xf[i], Q, R, Qf, 0.7, 0.1, T)
        Xo, Jo = p.rollout(x0[i], U0[i])
        assert relerr(X[i].cpu().numpy(), Xo) < tol_roll and abs(float(J[i]) - Jo) <= tol_roll * abs(Jo), i
        Xi = X[i].cpu().numpy()
        Ko, do = p.backward_pass(Xi, U0[i], mu[i])
        assert relerr(K[i].cpu().numpy(), Ko) < TOL_PASS and relerr(d[i].cpu().numpy(), do) < TOL_PASS, i
        for ai in (0, 4, 9):
            Xr, Ur, Jr = p.forward_pass(Xi, U0[i], K[i].cpu().numpy(), d[i].cpu().numpy(), al[ai])
            if np.isfinite(Jr) and abs(Jr) < 1e12:
                assert relerr(Xn[i, ai].cpu().numpy(), Xr) < 1e-8 and abs(float(Jn[i, ai]) - Jr) < 1e-8 * abs(Jr), (i, ai)


@pytest.mark.parametrize("model,k,T", [(3, 16, 20), (0, 17, 15)])
def test_large_cluster_solve_vs_oracle(dp, model, k, T):
    """Whole solves beyond n_x = 60 through the device-resident loop (windowed admission included)."""
    from oracle import oracle as orc
    from dpilqr_amd.util import random_setup
    B = 5
    x0 = np.zeros((B, k * 4)); xf = np.zeros((B, k * 4))
    for s in range(B):
        np.random.seed(700 + s)
        a, b = random_setup(k, 4, is_rotation=False, rel_dist=k, var=k / 2, n_d=2, random=True, energy=10.0)
        x0[s], xf[s] = a.ravel(), b.ravel()
    Q, R, Qf = np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4)
    U0 = np.zeros((B, T, k * 2))
    pb = dp.ProblemBatch([model] * k, [2] * k, xf, Q, R, Qf, 0.5, 0.1, T)
    r = pb.solve(x0, U0, n_lqr_iter=5, window=3)
    proto = orc.Problem([model] * k, [2] * k, xf[0], Q, R, Qf, 0.5, 0.1, T)
    o = orc.solve_batch(proto, x0, xf, U0, n_lqr_iter=5)
    op = orc.solve_batch(proto, x0 * (1 + 1e-13), xf, U0, n_lqr_iter=5)
    X = r["X"].cpu().numpy(); nb = r["n_bwd"].cpu().numpy(); nf = r["n_fwd"].cpu().numpy(); st = r["status"].cpu().numpy()
    n_well = 0
    for i in range(B):
        sens = relerr(op["X"][i], o["X"][i])
        if op["n_fwd"][i] == o["n_fwd"][i] and sens < 1e-7:
            n_well += 1
            assert (nb[i], nf[i], st[i]) == (o["n_bwd"][i], o["n_fwd"][i], o["status"][i]), i
            assert relerr(X[i], o["X"][i]) < TOL_SOLVE, i
    assert n_well >= 3 and np.isfinite(X).all()


def test_forced_big_path_equals_golden_in_a_fresh_process():
    """The large-cluster kernels on SMALL problems whose reference answers are committed: a child process with
    DPILQR_FORCE_BIG=1 (the switch is read once per process) checks G3 / G8 passes and G4 solves at fp64 tolerances."""
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
import dpilqr_amd as dp
from tests.golden_util import relerr
G = %r
def batch_from(z, prefix=""):
    g = lambda k: z[prefix + k]
    return dp.ProblemBatch(g("model"), g("n_dims"), g("xf")[None], g("Q"), g("R"), g("Qf"), float(g("radius")), float(g("dt")), int(g("T")))
for name in ("g3_passes_cfg2_di4d_k5", "g3_passes_quad6d_k3", "g3_passes_mixed_q6h6", "g3_passes_uni4d_k3", "g3_passes_car3d_k2",
             "g3_passes_quad12d_k2", "g8_hetero_k3_passes", "g3_passes_di4d_k1"):
    z = dict(np.load(G + "/" + name + ".npz"))
    pb = batch_from(z)
    K, d = pb.backward_pass(z["X"][None], z["U"][None], float(z["mu"]))
    assert relerr(K[0].cpu().numpy(), z["K"]) < 1e-9 and relerr(d[0].cpu().numpy(), z["d"]) < 1e-9, name
    Xn, Un, Jn = pb.forward_pass(z["X"][None], z["U"][None], z["K"][None], z["d"][None], z["alphas"])
    assert relerr(Xn[0].cpu().numpy(), z["X_fwd"]) < 1e-9 and relerr(Jn[0].cpu().numpy(), z["J_fwd"]) < 1e-9, name
    X, J = pb.rollout(z["x0"][None], z["U0"][None])
    assert relerr(X[0].cpu().numpy(), z["X_roll"]) < 1e-10, name
z = dict(np.load(G + "/g4_solves_cfg2.npz"))
from tests.golden_util import cfg2_params
c = cfg2_params()
for s in (0, 17, 2, 29):
    pb = dp.ProblemBatch(c["model"], c["n_dims"], z["s%%d_xf" %% s][None], c["Q"], c["R"], c["Qf"], c["radius"], c["dt"], c["T"])
    r = pb.solve(z["s%%d_x0" %% s][None], np.zeros((1, 50, 10)), trace=True)
    nb = len(z["s%%d_mu_trace" %% s]); tr = r["trace"][0].cpu().numpy()[:nb]
    assert int(r["n_bwd"][0]) == nb, s
    assert np.array_equal(tr[:, 1].astype(int), z["s%%d_acc_trace" %% s]) and np.array_equal(tr[:, 0], z["s%%d_mu_trace" %% s]), s
    assert relerr(r["X"][0].cpu().numpy(), z["s%%d_X" %% s]) < 1e-5, s
print("forced-big ok")
''' % (str(ROOT), str(ROOT / "tests" / "golden"))
    env = dict(os.environ, DPILQR_FORCE_BIG="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "forced-big ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.parametrize("case,tol_fwd", [("g3_passes_cfg2_di4d_k5", 1e-4), ("g3_passes_quad6d_k3", 0.2), ("g8_hetero_k3_passes", 1e-4)])
def test_fp32_passes_near_fp64(dp, golden, case, tol_fwd):
    """The fp32 arm against the reference's fp64 numbers, pass by pass: gains to 2e-3, forward-pass states to 1e-4 --
    except Quadcopter6D, whose closed loop through tan() amplifies single-precision rounding to several per cent over
    30 steps (a finding of the tolerance study; scripts/fp32_study.py reports the figures per configuration)."""
    import torch
    z = golden(case)
    pb = batch_from(dp, z)
    X, J = pb.rollout(z["x0"][None], z["U0"][None], dtype=torch.float32)
    assert X.dtype == torch.float32 and relerr(X[0].cpu().numpy(), z["X_roll"]) < 1e-4
    assert abs(float(J[0]) - z["J_roll"]) < 1e-4 * abs(z["J_roll"])
    K, d = pb.backward_pass(z["X"][None], z["U"][None], float(z["mu"]), dtype=torch.float32)
    assert K.dtype == torch.float32
    assert relerr(K[0].cpu().numpy(), z["K"]) < 2e-3 and relerr(d[0].cpu().numpy(), z["d"]) < 2e-3
    Xn, Un, Jn = pb.forward_pass(z["X"][None], z["U"][None], z["K"][None], z["d"][None], z["alphas"], dtype=torch.float32)
    assert relerr(Xn[0].cpu().numpy(), z["X_fwd"]) < tol_fwd and relerr(Jn[0].cpu().numpy(), z["J_fwd"]) < tol_fwd


def test_fp32_solve_runs_the_same_state_machine(dp, golden):
    """fp32 whole solves of cfg2 scenarios: same statuses / iteration structure as fp64 on most items, every trajectory
    finite, final cost within 1e-2 of the fp64 solve where the decision traces agree."""
    import torch
    from tests.golden_util import cfg2_params
    from dpilqr_amd.util import random_setup
    c = cfg2_params(); B = 64
    x0 = np.zeros((B, 20)); xf = np.zeros((B, 20))
    for s in range(B):
        np.random.seed(9000 + s)
        a, b = random_setup(5, 4, is_rotation=False, rel_dist=5, var=2.5, n_d=2, random=True, energy=10.0)
        x0[s], xf[s] = a.ravel(), b.ravel()
    pb = dp.ProblemBatch(c["model"], c["n_dims"], xf, c["Q"], c["R"], c["Qf"], c["radius"], c["dt"], c["T"])
    r64 = pb.solve(x0, np.zeros((B, 50, 10)))
    r32 = pb.solve(x0, np.zeros((B, 50, 10)), dtype=torch.float32, window=16)
    assert r32["X"].dtype == torch.float32 and torch.isfinite(r32["X"]).all()
    same = (r32["n_bwd"] == r64["n_bwd"]) & (r32["n_fwd"] == r64["n_fwd"]) & (r32["status"] == r64["status"])
    assert float(same.double().mean()) > 0.5
    J32, J64 = r32["J"].cpu().numpy(), r64["J"].cpu().numpy()
    sel = same.cpu().numpy()
    assert np.all(np.abs(J32[sel] - J64[sel]) < 1e-2 * np.abs(J64[sel]))
    assert (r32["status"] > 0).all()


def test_enqueue_only_solve_equals_synchronous_solve(dp):
    """dpilqr_solve_enqueue: a fixed number of global iterations enqueued without any host read; the bound always
    suffices, a smaller number plus a resumed call continues where the first stopped, and the answers are bit-identical
    to the synchronous solve (scheduling cannot change an item's arithmetic)."""
    import torch
    from tests.golden_util import cfg2_params
    from dpilqr_amd.util import random_setup
    c = cfg2_params(); B = 96
    x0 = np.zeros((B, 20)); xf = np.zeros((B, 20))
    for s in range(B):
        np.random.seed(5000 + s)
        a, b = random_setup(5, 4, is_rotation=False, rel_dist=5, var=2.5, n_d=2, random=True, energy=10.0)
        x0[s], xf[s] = a.ravel(), b.ravel()
    pb = dp.ProblemBatch(c["model"], c["n_dims"], xf, c["Q"], c["R"], c["Qf"], c["radius"], c["dt"], c["T"])
    U0 = np.zeros((B, 50, 10))
    ref = pb.solve(x0, U0, window=32)
    bound = pb.iterations_bound(50, window=32)
    assert bound == 3 * 50 + 1
    # (a) one call with the bound, on a side stream, nothing waited for until we look
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        r, _ = pb.solve_enqueue(x0, U0, bound, window=32)
    side.synchronize()
    for key in ("X", "U", "J", "status", "n_bwd", "n_fwd"):
        assert torch.equal(r[key], ref[key]), key
    # (b) too few iterations: some items still active; a resumed call finishes them
    r, state = pb.solve_enqueue(x0, U0, 4, window=32)
    torch.cuda.synchronize()
    assert int((r["status"] == 0).sum()) > 0
    for _ in range(40):
        r, state = pb.solve_enqueue(None, None, 5, state=state)
    torch.cuda.synchronize()
    assert int((r["status"] == 0).sum()) == 0
    for key in ("X", "U", "J", "status", "n_bwd", "n_fwd"):
        assert torch.equal(r[key], ref[key]), key


def test_enqueue_only_solve_can_be_captured_in_a_hip_graph(dp):
    """dpilqr_solve_enqueue neither allocates, nor reads anything on the host, nor synchronises: the whole iteration
    loop of a solve is captured into a HIP graph (torch.cuda.graph drives hipStreamBeginCapture / EndCapture) and
    replayed; the replayed solve gives the synchronous solve's answer bit for bit."""
    import torch
    from tests.golden_util import cfg2_params
    from dpilqr_amd.util import random_setup_batch
    c = cfg2_params(); B = 64
    x0, xf = random_setup_batch((7000, B), 5, 4, var=2.5, n_d=2, energy=10.0)
    pb = dp.ProblemBatch(c["model"], c["n_dims"], xf, c["Q"], c["R"], c["Qf"], c["radius"], c["dt"], c["T"])
    U0 = torch.zeros((B, 50, 10), dtype=torch.float64, device="cuda")
    ref = pb.solve(x0, U0, window=32)
    bound = pb.iterations_bound(50, window=32)
    r, state = pb.solve_enqueue(x0, U0, 0, window=32)          # initialise + rollout, eagerly (also warms every kernel)
    r, state = pb.solve_enqueue(None, None, 2, state=state)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        pb.solve_enqueue(None, None, bound, state=state)         # resume = 1: continues from the state left above
    graph.replay()
    torch.cuda.synchronize()
    for key in ("X", "U", "J", "status", "n_bwd", "n_fwd"):
        assert torch.equal(r[key], ref[key]), key


@pytest.mark.parametrize("B,dtype_name", [(1, "float64"), (3, "float64"), (2, "float32")])
def test_team_of_workgroups_gives_the_single_workgroups_gains(dp, monkeypatch, B, dtype_name):
    """Few items: helper workgroups take their share of S5 + S6's tile pairs and (twelve-state fp64) of S1's block pairs
    (csrc/riccati_big.hpp, BigTeam).  The same tiles,
    computed by the same instructions wherever they run: the gains of the team (the default; of two parts only; of as many as fit)
    are those of the single workgroup (DPILQR_BIG_TEAM=0) bit for bit -- and so are two consecutive team passes on one scratch."""
    import torch
    dtype = getattr(torch, dtype_name)
    models, nd, x0, xf, Q, R, Qf, U0, T = _cfg5_batch(True, tuple(range(6100, 6100 + B)))
    pb = dp.ProblemBatch(models, nd, xf, Q, R, Qf, 0.5, 0.1, T)
    X, J = pb.rollout(x0, U0, dtype=dtype)
    Ud = torch.as_tensor(U0, dtype=dtype, device="cuda")
    mu = torch.ones(B, dtype=torch.float64, device="cuda")

    def run(team):
        if team is None:
            monkeypatch.delenv("DPILQR_BIG_TEAM", raising=False)
        else:
            monkeypatch.setenv("DPILQR_BIG_TEAM", team)
        K, d = pb.backward_pass(X, Ud, mu, dtype=dtype)
        return K.cpu().numpy(), d.cpu().numpy()

    K0, d0 = run("0")
    assert np.isfinite(K0).all() and np.isfinite(d0).all()
    for team in (None, "2", None):
        K1, d1 = run(team)
        assert np.array_equal(K0, K1) and np.array_equal(d0, d1), team


@pytest.mark.parametrize("dtype_name", ["float64", "float32"])
def test_teams_that_fill_every_xcd_and_both_kinds_of_release(dp, monkeypatch, dtype_name):
    """32 items = four teams of eight workgroups on each XCD, every CU taken.  A team whose parts all report the main workgroup's
    XCD releases without writing the L2 back (csrc/riccati_big.hpp, big_arrive); DPILQR_BIG_TEAM_AGENT=1 keeps the agent-scope
    release.  Both give the single workgroup's gains bit for bit, pass after pass (round 6: a first form of the local hand-over
    that also weakened the ACQUIRE passed fp64 everywhere and failed here, in fp32, sporadically -- stale vector-cache lines)."""
    import torch
    dtype = getattr(torch, dtype_name)
    B = 32
    models, nd, x0, xf, Q, R, Qf, U0, T = _cfg5_batch(True, tuple(range(6200, 6200 + B)))
    pb = dp.ProblemBatch(models, nd, xf, Q, R, Qf, 0.5, 0.1, T)
    X, J = pb.rollout(x0, U0, dtype=dtype)
    Ud = torch.as_tensor(U0, dtype=dtype, device="cuda")
    mu = torch.ones(B, dtype=torch.float64, device="cuda")

    def run(**env):
        for key in ("DPILQR_BIG_TEAM", "DPILQR_BIG_TEAM_AGENT"):
            monkeypatch.delenv(key, raising=False)
        for key, v in env.items():
            monkeypatch.setenv(key, v)
        K, d = pb.backward_pass(X, Ud, mu, dtype=dtype)
        return K.cpu().numpy(), d.cpu().numpy()

    K0, d0 = run(DPILQR_BIG_TEAM="0")
    assert np.isfinite(K0).all() and np.isfinite(d0).all()
    for rep in range(4):
        for env in ({}, {"DPILQR_BIG_TEAM_AGENT": "1"}):
            K1, d1 = run(**env)
            assert np.array_equal(K0, K1) and np.array_equal(d0, d1), (rep, env)


def test_team_helpers_that_start_late_leave_the_pass_to_the_main_workgroup(dp, monkeypatch):
    """What a chip busy with other work does to the team: the helpers report after the main workgroup has looked for them
    (DPILQR_BIG_TEAM_LATE delays them by a few milliseconds).  The main workgroup then runs the whole pass alone -- the
    single-workgroup gains, bit for bit, no hang -- and the late helpers leave at once; the next pass, with the helpers on
    time again, is a team's."""
    import time
    import torch
    models, nd, x0, xf, Q, R, Qf, U0, T = _cfg5_batch(True, (6200,))
    pb = dp.ProblemBatch(models, nd, xf, Q, R, Qf, 0.5, 0.1, T)
    X, J = pb.rollout(x0, U0)
    Ud = torch.as_tensor(U0, dtype=torch.float64, device="cuda")
    mu = torch.ones(1, dtype=torch.float64, device="cuda")

    def run(**env):
        for key in ("DPILQR_BIG_TEAM", "DPILQR_BIG_TEAM_LATE"):
            monkeypatch.delenv(key, raising=False)
        for key, v in env.items():
            monkeypatch.setenv(key, v)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K, d = pb.backward_pass(X, Ud, mu)
        torch.cuda.synchronize()
        return K.cpu().numpy(), d.cpu().numpy(), time.perf_counter() - t0

    run()                                         # (warm)
    K0, d0, t_alone = run(DPILQR_BIG_TEAM="0")
    K1, d1, t_late = run(DPILQR_BIG_TEAM_LATE="1")
    K2, d2, t_team = run()
    assert np.array_equal(K0, K1) and np.array_equal(d0, d1)
    assert np.array_equal(K0, K2) and np.array_equal(d0, d2)
    assert t_late > 0.9 * t_alone and t_team < 0.85 * t_alone, (t_alone, t_late, t_team)    # alone again / a team again


@pytest.mark.parametrize("model,k,ns,nc,nd,T", [(3, 30, 4, 2, 2, 30), (4, 20, 6, 3, 3, 25), (7, 10, 12, 4, 3, 20), (0, 24, 4, 2, 2, 20)])
def test_team_at_other_sizes_and_families(dp, monkeypatch, model, k, ns, nc, nd, T):
    """The team with two or three parts (n_x = 96, 120) and with the vector forms of S1 (four- and six-state agents, fp32): the
    single workgroup's gains bit for bit."""
    import torch
    from dpilqr_amd.util import random_setup
    np.random.seed(5)
    a, b = random_setup(k, ns, is_rotation=False, rel_dist=k, var=k / 2, n_d=nd, random=True, energy=10.0 * k)
    x0, xf = a.ravel()[None], b.ravel()[None]
    U0 = np.zeros((1, T, k * nc))
    if model == 4:
        U0[:, :, 0::3] = 9.80665
    if model == 7:
        U0[:, :, 3::4] = 9.80665 * 63.0 / 2000.0
    pb = dp.ProblemBatch([model] * k, [nd] * k, xf, np.eye(ns), np.eye(nc), 100.0 * np.eye(ns), 0.5, 0.1, T)
    for dtype in (torch.float64, torch.float32):
        X, J = pb.rollout(x0, U0, dtype=dtype)
        Ud = torch.as_tensor(U0, dtype=dtype, device="cuda")
        mu = torch.ones(1, dtype=torch.float64, device="cuda")
        monkeypatch.setenv("DPILQR_BIG_TEAM", "0")
        K0, d0 = pb.backward_pass(X, Ud, mu, dtype=dtype)
        monkeypatch.delenv("DPILQR_BIG_TEAM")
        K1, d1 = pb.backward_pass(X, Ud, mu, dtype=dtype)
        assert torch.isfinite(K0).all() and torch.equal(K0, K1) and torch.equal(d0, d1), dtype


def test_team_whose_helpers_stall_ends_in_a_status_not_a_dead_context(dp, monkeypatch):
    """The team kernel's failure path is a return code (include/dpilqr_hip.h: "never throws or aborts"; it used to be five
    __builtin_trap sites, which poison the HIP context of the whole process).  Fault injection: DPILQR_BIG_TEAM_LATE=2 makes the
    helpers join the team and then never work; DPILQR_BIG_SPIN_LOG2 bounds every wait at 2^13 polls so that the main workgroup's
    first wait for them expires in milliseconds.  Expected: the bare pass returns (its first gain offset NaN), the solve
    retires the item with DPILQR_STATUS_FAULT after its first backward pass with X, U the initial rollout, dpilqr_solve_batch
    reports DPILQR_EHIP with a message -- and the SAME process then runs the pass and the solve normally, with the results a
    process that never saw the fault gets."""
    import torch
    from dpilqr_amd import _lib
    models, nd, x0, xf, Q, R, Qf, U0, T = _cfg5_batch(True, (6300,))
    pb = dp.ProblemBatch(models, nd, xf, Q, R, Qf, 0.5, 0.1, T)
    X, J = pb.rollout(x0, U0)
    Ud = torch.as_tensor(U0, dtype=torch.float64, device="cuda")
    mu = torch.ones(1, dtype=torch.float64, device="cuda")
    K_ok, d_ok = pb.backward_pass(X, Ud, mu)
    good = pb.solve(x0, U0, n_lqr_iter=2)
    assert torch.isfinite(K_ok).all() and int(good["status"][0]) != _lib.STATUS_FAULT

    monkeypatch.setenv("DPILQR_BIG_TEAM_LATE", "2")
    monkeypatch.setenv("DPILQR_BIG_SPIN_LOG2", "13")
    K_bad, d_bad = pb.backward_pass(X, Ud, mu)
    torch.cuda.synchronize()                                     # the launch ENDS: no trap, no hang
    assert torch.isnan(d_bad[0, 0, 0])
    with pytest.raises(_lib.DpilqrError) as ei:
        pb.solve(x0, U0, n_lqr_iter=2)
    assert ei.value.code == _lib.EHIP and "gave up 1 of 1 items" in str(ei.value)
    r = ei.value.results
    assert int(r["status"][0]) == _lib.STATUS_FAULT and int(r["n_bwd"][0]) == 1 and int(r["n_fwd"][0]) == 0
    assert torch.equal(r["X"], X) and torch.equal(r["U"], Ud)    # the last accepted iterate = the initial rollout
    # the enqueue-only form: the status alone tells
    r2, _ = pb.solve_enqueue(x0, U0, 3, n_lqr_iter=2)
    torch.cuda.synchronize()
    assert int(r2["status"][0]) == _lib.STATUS_FAULT

    monkeypatch.delenv("DPILQR_BIG_TEAM_LATE")
    monkeypatch.delenv("DPILQR_BIG_SPIN_LOG2")
    K2, d2 = pb.backward_pass(X, Ud, mu)                          # the context is alive, the team works again
    assert torch.equal(K2, K_ok) and torch.equal(d2, d_ok)
    again = pb.solve(x0, U0, n_lqr_iter=2)
    for key in ("X", "U", "J", "status", "n_bwd", "n_fwd"):       # (J: the last EVALUATED cost, NaN where tan() overflowed a candidate)
        assert np.array_equal(again[key].cpu().numpy(), good[key].cpu().numpy(), equal_nan=True), key


def test_team_hand_overs_under_uneven_load(dp, monkeypatch):
    """The team's hand-overs (one relaxed poll, ONE agent-scope acquire by the polling wavefront, the workgroup barrier; arrivals
    behind every wavefront's drained stores and an agent-scope release) with the chip busy and uneven: a copy stream of a few
    hundred MB runs beside every team pass (other workgroups take CUs and L2, the parts start at different times, the caches are
    warm with the previous pass's lines of the same addresses).  Ten passes of one item and of three, every word of the gains
    compared with the single workgroup's."""
    import torch
    for B in (1, 3):
        models, nd, x0, xf, Q, R, Qf, U0, T = _cfg5_batch(True, tuple(range(6400, 6400 + B)))
        pb = dp.ProblemBatch(models, nd, xf, Q, R, Qf, 0.5, 0.1, T)
        X, J = pb.rollout(x0, U0)
        Ud = torch.as_tensor(U0, dtype=torch.float64, device="cuda")
        mu = torch.ones(B, dtype=torch.float64, device="cuda")
        monkeypatch.setenv("DPILQR_BIG_TEAM", "0")
        K0, d0 = pb.backward_pass(X, Ud, mu)
        monkeypatch.delenv("DPILQR_BIG_TEAM")
        src = torch.empty(96 << 20, dtype=torch.float32, device="cuda").normal_()      # 384 MB: past L2 and the Infinity Cache
        dst = torch.empty_like(src)
        side = torch.cuda.Stream()
        for rep in range(10):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(1 + rep % 3):
                    dst.copy_(src); src.mul_(1.0000001)
            K1, d1 = pb.backward_pass(X, Ud, mu)
            torch.cuda.current_stream().wait_stream(side)
            assert torch.equal(K0, K1) and torch.equal(d0, d1), (B, rep)
