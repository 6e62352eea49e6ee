"""Calibration of the all-items parity envelope (oracle/parity.py) on the CPU: what it lets through and what it catches.

The GPU tests hold every item of a batch of whole solves to that envelope; this file shows, without a GPU, that
  * a legitimately different implementation -- the oracle itself compiled with fused multiply-adds, i.e. every a*b+c
    rounded once instead of twice -- passes on every item, decision flips and chaotic items included;
  * an implementation error of 1e-9 (relative, on the inputs) fails on (nearly) every item and 1e-11 on many;
  * a wrong convergence test (tol 1.5e-3 instead of 1e-3) is caught as unexplained convergence flips;
  * a wrong acceptance rule (accept a candidate that does not lower the cost) is caught;
  * the replay of the oracle's own decisions IS the oracle's solve, bit for bit (the replay adds nothing of its own).
"""
import ctypes as C
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc, parity

B = 256


@pytest.fixture(scope="module")
def batch():
    from dpilqr_amd.util import random_setup
    x0 = np.zeros((B, 20)); xf = np.zeros((B, 20))
    for s in range(B):
        np.random.seed(1000 + s)
        a, b = random_setup(5, 4, is_rotation=False, rel_dist=5, var=2.5, n_d=2, random=True, energy=10.0)
        x0[s], xf[s] = a.ravel(), b.ravel()
    Q, R, Qf = np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4)
    proto = orc.Problem([0] * 5, [2] * 5, xf[0], Q, R, Qf, 0.5, 0.1, 50)
    U0 = np.zeros((B, 50, 10))
    o = orc.solve_batch(proto, x0, xf, U0, trace=True)
    return proto, x0, xf, U0, o


def _solve_with_library(path, fn):
    """Run fn() with oracle.lib() bound to another build of ilqr_oracle.c."""
    real_build = orc.build
    orc._LIB = None
    orc.build = lambda force=False: path
    try:
        orc.lib()
        return fn()
    finally:
        orc.build = real_build
        orc._LIB = None
        orc.lib()


def test_replay_of_the_oracles_own_decisions_is_the_oracles_solve(batch):
    proto, x0, xf, U0, o = batch
    r = orc.replay_batch(proto, x0, xf, U0, o)
    assert np.array_equal(r["X"], o["X"]) and np.array_equal(r["U"], o["U"]) and np.array_equal(r["J"], o["J"])
    for i in range(B):
        n = o["n_bwd"][i]
        assert np.array_equal(r["rtrace"][i, :n, 1], o["trace"][i, :n, 1])          # own verdict = the decision taken
        assert np.array_equal(r["rtrace"][i, :n, 3], o["trace"][i, :n, 3])
        assert (r["rtrace"][i, :n, 4] == 0).all() and (r["rtrace"][i, :n, 5] == 0).all()   # no margins: nothing flipped
    rep = parity.envelope(o, proto, x0, xf, U0, natural=o)
    assert rep["summary"]["all_ok"] and rep["summary"]["flipped_frac"] == 0.0


def test_fused_multiply_add_build_of_the_oracle_passes_on_every_item(batch, tmp_path):
    if "fma" not in open("/proc/cpuinfo").read():
        pytest.skip("host CPU has no FMA")
    so = tmp_path / "liboracle_fma.so"
    src = orc.HERE / "ilqr_oracle.c"
    subprocess.run(["gcc", "-O3", "-fPIC", "-shared", "-fopenmp", "-mfma", "-mavx2", "-ffp-contract=fast", "-std=c99",
                    "-o", str(so), str(src), "-lm"], check=True)
    proto, x0, xf, U0, o = batch
    g = _solve_with_library(str(so), lambda: orc.solve_batch(proto, x0, xf, U0, trace=True))
    assert not np.array_equal(g["X"], o["X"])            # it IS a different rounding
    rep = parity.envelope(g, proto, x0, xf, U0, natural=o)
    sm = rep["summary"]
    assert sm["all_ok"], (sm, [w for w in rep["why"] if w][:5])
    assert sm["identical_decision_trace_frac"] > 0.95
    tight = rep["spreadX"] < 1e-6
    assert tight.mean() > 0.85 and (rep["errX"][tight] < 1e-5).all()


@pytest.mark.parametrize("delta,min_fail_frac", [(1e-9, 0.95), (1e-11, 0.15)])
def test_input_error_is_caught(batch, delta, min_fail_frac):
    proto, x0, xf, U0, o = batch
    g = orc.solve_batch(proto, x0 * (1 + delta), xf, U0, trace=True)
    rep = parity.envelope(g, proto, x0, xf, U0)
    assert rep["summary"]["violations"] >= min_fail_frac * B, rep["summary"]


def test_perturbations_inside_the_ensembles_range_pass(batch):
    proto, x0, xf, U0, o = batch
    for delta in (3e-14, -7e-14):
        g = orc.solve_batch(proto, x0 * (1 + delta), xf, U0, trace=True)
        rep = parity.envelope(g, proto, x0, xf, U0)
        assert rep["summary"]["all_ok"], (delta, rep["summary"], [w for w in rep["why"] if w][:3])


def test_wrong_convergence_tolerance_is_caught(batch):
    proto, x0, xf, U0, o = batch
    g = orc.solve_batch(proto, x0, xf, U0, tol=1.5e-3, trace=True)
    rep = parity.envelope(g, proto, x0, xf, U0)
    bad = [w for w in rep["why"] if w]
    assert len(bad) >= 5 and all("convergence flip" in w for w in bad), bad[:3]


def test_wrong_acceptance_is_caught(batch):
    """Forge a result whose iteration 0 'accepted' alpha_1 although alpha_0 lowers the cost: the oracle's numbers are used
    for everything else, so only the decision is wrong."""
    proto, x0, xf, U0, o = batch
    forged = {k: np.array(v, copy=True) for k, v in o.items()}
    sel = np.where((o["trace"][:, 0, 1] == 0) & (o["n_bwd"] >= 2))[0][:16]
    assert len(sel) >= 8
    forged["trace"][sel, 0, 1] = 1
    forged["trace"][sel, 0, 4] = 2
    r = orc.replay_batch(proto, x0, xf, U0, forged)       # what following that decision gives (the oracle's own arithmetic)
    for k in ("X", "U", "J"):
        forged[k] = r[k]
    forged["trace"][:, :, 3] = r["rtrace"][:, :, 3]; forged["trace"][:, :, 2] = r["rtrace"][:, :, 2]
    rep = parity.envelope(forged, proto, x0, xf, U0)
    assert not rep["ok"][sel].any(), rep["summary"]
    assert all("acceptance flip at iteration 0" in rep["why"][i] for i in sel)
    others = np.setdiff1d(np.arange(B), sel)
    assert rep["ok"][others].all()
