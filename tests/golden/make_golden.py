#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ from the REAL reference.

This script is the only place where the reference (labicon/dp-ilqr, mounted
read-only at /root/reference) is executed.  It
  1. copies the reference package to a scratch directory OUTSIDE the repo
     (/tmp/oracle_ref), builds its Cython extension there, and imports it;
  2. runs the reference's own functions on seeded inputs;
  3. stores inputs AND outputs as small .npz files next to this script.

Nothing of the reference (source, bytecode, .so) is written into the repo: the
.npz files hold arrays only.  The GPU box has no /root/reference, so the tests
only ever read the .npz files.

Manifest:
  g1_models.npz      f / integrate / linearize for Model 0..7
  g2_costs.npz       ReferenceCost, quadraticize_distance, ProximityCost, GameCost
  g3_passes_*.npz    rollout, backward pass (K,d), forward pass (10 alphas)
  g4_solves_*.npz    full ilqrSolver.solve incl. decision trace
  g5_dispatch.npz    define_inter_graph_threshold, split_graph, solve_distributed
  g6_scenarios.npz   np.random.seed(s); random_setup(...)
  g7_callers.npz     solve_rhc (centralized and distributed branch, with the logging rows), selfish_warmstart,
                     solve_subproblem
  g9_chaos_*.npz     the reference on items the GPU decides differently from the oracle: x0 and 32 perturbed copies
                     (python tests/golden/make_golden.py g9; needs gpurun_out/flips/ from scripts/find_flips.py)
  g10_harness.npz    the Monte-Carlo harness itself: the reference's scripts/analysis.py::multi_agent_run run as it is on a
                     seeded stream (both branches; the distributed one through ignore_ids=[] as in G7): the draws and the rows
  g8_hetero_*.npz    the zero-padded human model (12 states / 4 controls) mixed with Quadcopter12D: a shim class
                     built from reference calls (the reference itself cannot mix 12- and 6-state agents)

Run:  python tests/golden/make_golden.py
"""
import io
import os
import shutil
import subprocess
import sys
from contextlib import redirect_stdout
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
SCRATCH = Path("/tmp/oracle_ref")
OUT = Path(__file__).resolve().parent


def import_reference():
    if not (REF / "dpilqr" / "control.py").exists():
        raise SystemExit("reference not mounted at /root/reference - nothing to do")
    so = list((SCRATCH / "dpilqr").glob("bbdynamicswrap*.so")) if SCRATCH.exists() else []
    if not so:
        if SCRATCH.exists():
            shutil.rmtree(SCRATCH)
        SCRATCH.mkdir(parents=True)
        shutil.copytree(REF / "dpilqr", SCRATCH / "dpilqr")
        shutil.copy(REF / "setup.py", SCRATCH / "setup.py")
        subprocess.run(["chmod", "-R", "u+w", str(SCRATCH)], check=True)
        subprocess.run([sys.executable, "setup.py", "build_ext", "--inplace"],
                       cwd=SCRATCH, check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.path.insert(0, str(SCRATCH))
    import dpilqr  # noqa
    return dpilqr


dp = import_reference()

MODEL_CLASSES = [
    ("DoubleInt4D", dp.DoubleIntDynamics4D),
    ("DoubleInt6D", dp.DoubleIntDynamics6D),
    ("Car3D", dp.CarDynamics3D),
    ("Unicycle4D", dp.UnicycleDynamics4D),
    ("Quadcopter6D", dp.QuadcopterDynamics6D),
    ("Human6D", dp.HumanDynamics6D),
    ("HumanLin6D", dp.HumanDynamicsLin6D),
    ("Quadcopter12D", dp.QuadcopterDynamics12D),
]
G = 9.80665


class HumanDynamics6DPadded12(dp.DynamicalModel):
    """BASELINE config 5's "zero-padded" human: HumanDynamics6D carried in a 12-state / 4-control slot so that it
    can be stacked with QuadcopterDynamics12D (MultiDynamicalModel needs uniform dims, dynamics.py:165-170).
    Built from reference calls only: the first 6 states / 3 controls are the reference's HumanDynamics6D, the
    padded states never move (A = 1 on their diagonal, B = 0) and the padded control does nothing."""

    def __init__(self, dt, id=None):
        super().__init__(12, 4, dt, id)
        self.human = dp.HumanDynamics6D(dt, self.id)

    def __call__(self, x, u):
        return np.r_[self.human(np.ascontiguousarray(x[:6]), np.ascontiguousarray(u[:3])), x[6:]]

    def f(self, x, u):
        return np.r_[self.human.f(np.ascontiguousarray(x[:6]), np.ascontiguousarray(u[:3])), np.zeros(6)]

    def linearize(self, x, u):
        Ah, Bh = self.human.linearize(np.ascontiguousarray(x[:6]), np.ascontiguousarray(u[:3]))
        A = np.eye(12); A[:6, :6] = Ah
        B = np.zeros((12, 4)); B[:6, :3] = Bh
        return A, B


MODEL_ENUM = {c: dp.Model[n].value for n, c in MODEL_CLASSES}
MODEL_ENUM[HumanDynamics6DPadded12] = 8      # the build's own enum value for the padded model


# --------------------------------------------------------------------------- G1
def g1_models():
    out = {}
    rng = np.random.default_rng(1001)
    for enum_val, (name, cls) in enumerate(MODEL_CLASSES):
        assert dp.Model[name].value == enum_val
        n_s, n_c = cls(0.1).n_x, cls(0.1).n_u
        n = 8
        x = rng.normal(size=(n, n_s))
        u = rng.normal(size=(n, n_c)) * 0.5
        if name == "Quadcopter12D":
            x[:, 3:6] *= 0.4  # keep Euler angles away from cos(theta)=0
            u[:, :3] *= 1e-3  # torques act through 1/inertia ~ 5e4: keep the step finite
        dt = np.where(np.arange(n) % 2 == 0, 0.05, 0.1)
        f = np.zeros((n, n_s)); xn = np.zeros((n, n_s))
        A = np.zeros((n, n_s, n_s)); B = np.zeros((n, n_s, n_c))
        for i in range(n):
            m = cls(float(dt[i]))
            f[i] = m.f(x[i].copy(), u[i].copy())
            xn[i] = m(x[i].copy(), u[i].copy())
            A[i], B[i] = m.linearize(x[i].copy(), u[i].copy())
        out.update({f"{name}_x": x, f"{name}_u": u, f"{name}_dt": dt, f"{name}_f": f,
                    f"{name}_xn": xn, f"{name}_A": A, f"{name}_B": B,
                    f"{name}_enum": np.array(enum_val)})
    # the DoubleInt4D hand-computed trajectory the reference's own test holds
    # (tests/test_dynamics.py:31-37), run through the reference model (dt=0.5)
    m = dp.DoubleIntDynamics4D(0.5)
    x = np.array([0.0, 2, 0, -2]); u = np.array([0.0, 2.0])
    traj = [x.copy()]
    for _ in range(4):
        x = m(x, u); traj.append(x.copy())
    out["DoubleInt4D_reftest_traj"] = np.array(traj)
    np.savez_compressed(OUT / "g1_models.npz", **out)


# --------------------------------------------------------------------------- G2
def g2_costs():
    out = {}
    rng = np.random.default_rng(2002)
    # ReferenceCost with NON-symmetric Q, R (exercises Q+Q^T), cost.py:79-101
    n_s, n_c = 4, 2
    Q = rng.normal(size=(n_s, n_s)); R = rng.normal(size=(n_c, n_c)); Qf = rng.normal(size=(n_s, n_s))
    xf = rng.normal(size=n_s)
    rc = dp.ReferenceCost(xf, Q, R, Qf, 100)
    x = rng.normal(size=(4, n_s)); u = rng.normal(size=(4, n_c))
    out.update(ref_Q=Q, ref_R=R, ref_Qf=Qf, ref_xf=xf, ref_x=x, ref_u=u)
    for term in (False, True):
        tag = "T" if term else "S"
        out[f"ref_cost_{tag}"] = np.array([np.asarray(rc(x[i], u[i], term)).item() for i in range(4)])
        qs = [rc.quadraticize(x[i], u[i], term) for i in range(4)]
        for j, nm in enumerate(["Lx", "Lu", "Lxx", "Luu", "Lux"]):
            out[f"ref_{nm}_{tag}"] = np.array([q[j] for q in qs])
    # the reference's own passing test (tests/test_cost.py:39-78): Q=R=I, Qf=diag(1,1,0)
    xi = np.array([3.0, 7.0, 2.0]); ui = np.array([4.0, 1.0])
    rc2 = dp.ReferenceCost(np.zeros(3), np.eye(3), np.eye(2), np.diag([1.0, 1, 0]), 101)
    out["reftest_x"] = xi; out["reftest_u"] = ui
    out["reftest_cost"] = np.array([np.asarray(rc2(xi, ui)).item(),
                                    np.asarray(rc2(xi, ui, terminal=True)).item()])

    # quadraticize_distance inside / outside / at the radius, 2-D and 3-D (cost.py:269-315)
    pa = np.array([[0.1, 0.2, 0.0], [0.1, 0.2, 0.0], [0.0, 0.0, 0.0], [1.3, -0.4, 0.7], [1.3, -0.4, 0.7], [2.5, 1.0, 0.3]])
    pb = np.array([[0.3, 0.5, 0.0], [2.3, 0.5, 0.0], [0.5, 0.0, 0.0], [1.1, -0.1, 0.9], [3.1, -0.1, 0.9], [2.25, 1.1, 0.1]])
    nd = np.array([2, 2, 2, 3, 3, 3]); rad = 0.5
    Lx = np.zeros((6, 3)); Lxx = np.zeros((6, 3, 3))
    for i in range(6):
        a = dp.Point(*pa[i, :nd[i]]); b = dp.Point(*pb[i, :nd[i]])
        g, H = dp.quadraticize_distance(a, b, rad, int(nd[i]))
        Lx[i, :nd[i]] = g; Lxx[i, :nd[i], :nd[i]] = H
    out.update(qd_pa=pa, qd_pb=pb, qd_nd=nd, qd_radius=np.array(rad), qd_Lx=Lx, qd_Lxx=Lxx)

    # ProximityCost: n_dims [2]*3, [3]*3 (planar-cost quirk Q5), mixed [3,3,2]
    for tag, n_s_, n_dims in (("p2", 4, [2, 2, 2]), ("p3", 6, [3, 3, 3]), ("pm", 6, [3, 3, 2])):
        k = 3
        pc = dp.ProximityCost([n_s_] * k, 0.5, n_dims)
        xs = rng.normal(size=(4, k * n_s_)) * 0.3
        xs[:, 2::n_s_] = np.abs(xs[:, 2::n_s_]) + 0.05  # nonzero z (Q7)
        out[f"{tag}_x"] = xs; out[f"{tag}_ndims"] = np.array(n_dims)
        out[f"{tag}_cost"] = np.array([float(pc(xs[i])) for i in range(4)])
        qs = [pc.quadraticize(xs[i]) for i in range(4)]
        out[f"{tag}_Lx"] = np.array([q[0] for q in qs]); out[f"{tag}_Lxx"] = np.array([q[1] for q in qs])

    # GameCost k=1,3,5 on DoubleInt4D-shaped agents (cost.py:197-239)
    for k in (1, 3, 5):
        n_s, n_c = 4, 2
        xf = rng.normal(size=k * n_s)
        Q = np.diag([1.0, 1, 0, 0]); R = np.eye(2); Qf = 1000.0 * np.eye(4)
        refs = [dp.ReferenceCost(xf[i * n_s:(i + 1) * n_s], Q.copy(), R.copy(), Qf.copy(), 100 + i) for i in range(k)]
        gc = dp.GameCost(refs, dp.ProximityCost([n_s] * k, 0.5, [2] * k))
        xs = rng.normal(size=(3, k * n_s)) * 0.35; us = rng.normal(size=(3, k * n_c))
        out[f"gc{k}_xf"] = xf; out[f"gc{k}_x"] = xs; out[f"gc{k}_u"] = us
        for term in (False, True):
            tag = "T" if term else "S"
            out[f"gc{k}_cost_{tag}"] = np.array([np.asarray(gc(xs[i], us[i], term)).item() for i in range(3)])
            qs = [gc.quadraticize(xs[i], us[i], term) for i in range(3)]
            for j, nm in enumerate(["Lx", "Lu", "Lxx", "Luu", "Lux"]):
                out[f"gc{k}_{nm}_{tag}"] = np.array([q[j] for q in qs])
    np.savez_compressed(OUT / "g2_costs.npz", **out)


# ----------------------------------------------------------------- problem setup
def analysis_problem(model_cls, k, seed, radius=0.5, dt=0.1, n_d=None):
    """The scenario distribution of scripts/analysis.py:45-69,140-143."""
    n_s = model_cls(dt).n_x
    if n_d is None:
        n_d = 3 if n_s == 6 else 2
    np.random.seed(seed)
    x0, xf = dp.random_setup(k, n_s, is_rotation=False, rel_dist=k, var=k / 2,
                             n_d=n_d, random=True, energy=10.0)
    return build_problem([model_cls] * k, x0, xf, radius, dt, [n_d] * k)


def build_problem(model_classes, x0, xf, radius, dt, n_dims, ids=None):
    k = len(model_classes)
    ids = ids or [100 + i for i in range(k)]
    models = [c(dt, id_) for c, id_ in zip(model_classes, ids)]
    n_s, n_c = models[0].n_x, models[0].n_u
    if n_s == 4:
        Q = np.diag([1.0, 1.0, 0, 0]); R = np.eye(2)
    elif n_s == 6:
        Q = np.eye(6) * 50; R = np.eye(3)
    elif n_s == 3:
        Q = np.eye(3); R = np.eye(2)
    else:
        Q = np.eye(n_s); R = np.eye(n_c)
    Qf = 1000.0 * np.eye(n_s)
    Qs, Rs, Qfs = [Q.copy() for _ in range(k)], [R.copy() for _ in range(k)], [Qf.copy() for _ in range(k)]
    for i, c in enumerate(model_classes):
        if c is HumanDynamics6DPadded12:
            # the human's weights of examples.py:89-93 (R entry 1e-9 for its unused third control), and the same
            # 1e-9 for the padded fourth control so that Q_uu stays non-singular; padded states carry no weight
            Qs[i] = np.diag([1.0, 1, 1, 0, 0, 0] + [0.0] * 6)
            Rs[i] = np.diag([1.0, 1, 1e-9, 1e-9])
    xf = np.asarray(xf).reshape(-1)
    refs = [dp.ReferenceCost(xf[i * n_s:(i + 1) * n_s], Qs[i], Rs[i], Qfs[i], id_)
            for i, id_ in enumerate(ids)]
    prox = dp.ProximityCost([n_s] * k, radius, list(n_dims))
    prob = dp.ilqrProblem(dp.MultiDynamicalModel(models), dp.GameCost(refs, prox))
    meta = dict(x0=np.asarray(x0).reshape(-1), xf=xf, Q=np.array(Qs), R=np.array(Rs),
                Qf=np.array(Qfs), radius=np.array(radius), dt=np.array(dt),
                n_dims=np.array(n_dims), k=np.array(k), n_s=np.array(n_s), n_c=np.array(n_c),
                model=np.array([MODEL_ENUM[mc] for mc in model_classes]),
                ids=np.array(ids))
    return prob, meta


def warm_U(model_classes, T):
    """U0 = 0, except hover thrust for the quadcopter models (examples.py:122)."""
    cols = []
    for c in model_classes:
        if c is dp.QuadcopterDynamics6D:
            cols.append(np.tile([G, 0, 0], (T, 1)))
        elif c is dp.QuadcopterDynamics12D:
            cols.append(np.tile([0, 0, 0, G * 63.0 / 2000.0], (T, 1)))
        elif c is HumanDynamics6DPadded12:
            cols.append(np.zeros((T, 4)))
        else:
            cols.append(np.zeros((T, c(0.1).n_u)))
    return np.hstack(cols)


class TracingSolver(dp.ilqrSolver):
    """ilqrSolver that records the decision trace; numerics untouched."""

    def reset_trace(self):
        self.bwd_mu = []; self.fwd = []; self.last_Kd = None

    def _backward_pass(self, X, U):
        self.bwd_mu.append(self.μ)
        K, d = super()._backward_pass(X, U)
        self.last_Kd = (K.copy(), d.copy())
        self.fwd.append([])
        return K, d

    def _forward_pass(self, X, U, K, d, α):
        r = super()._forward_pass(X, U, K, d, α)
        self.fwd[-1].append(r[2])
        return r


def traced_solve(prob, x0, U0, T, tol=1e-3, n_lqr_iter=50):
    s = TracingSolver(prob, T); s.reset_trace()
    _, J0 = s._rollout(x0.reshape(-1, 1), U0)
    s.reset_trace()
    X, U, J = s.solve(x0.copy(), U0.copy(), n_lqr_iter=n_lqr_iter, tol=tol, verbose=False)
    n_b = len(s.bwd_mu)
    mu = np.array(s.bwd_mu); nf = np.array([len(f) for f in s.fwd])
    Jl = np.array([f[-1] for f in s.fwd]); acc = np.zeros(n_b, dtype=np.int32)
    Js = J0; Jstar = np.zeros(n_b)
    for i in range(n_b):
        if Jl[i] < Js:
            acc[i] = nf[i] - 1; Js = Jl[i]
        else:
            acc[i] = -1
        Jstar[i] = Js
    return dict(X=X, U=U, J=np.array(J), J0=np.array(J0), mu_trace=mu, nfwd_trace=nf,
                Jlast_trace=Jl, acc_trace=acc, Jstar_trace=Jstar,
                K_last=s.last_Kd[0], d_last=s.last_Kd[1])


# --------------------------------------------------------------------------- G3
def g3_case(tag, model_classes, T, seed, n_dims, radius=0.5, warm_iters=2, x0=None, xf=None):
    k = len(model_classes)
    n_s = model_classes[0](0.1).n_x
    if x0 is None:
        np.random.seed(seed)
        x0, xf = dp.random_setup(k, n_s, is_rotation=False, rel_dist=k, var=k / 2,
                                 n_d=max(n_dims), random=True, energy=10.0)
    prob, meta = build_problem(model_classes, x0, xf, radius, 0.1, n_dims)
    U0 = warm_U(model_classes, T)
    s = dp.ilqrSolver(prob, T)
    x0v = meta["x0"]
    Xr, Jr = s._rollout(x0v.reshape(-1, 1), U0)
    out = dict(meta); out.update(T=np.array(T), U0=U0, X_roll=Xr, J_roll=np.array(Jr))
    # operating point: after `warm_iters` iLQR iterations so that agents interact
    if warm_iters:
        X, U, _ = s.solve(x0v.copy(), U0.copy(), n_lqr_iter=warm_iters, verbose=False)
    else:
        X, U = Xr, U0
    mu = s.μ if warm_iters else 1.0
    s.μ = mu
    K, d = s._backward_pass(X, U)
    alphas = 1.1 ** (-np.arange(10, dtype=np.float32) ** 2)
    Xs, Us, Js = [], [], []
    for a in alphas:
        Xn, Un, Jn = s._forward_pass(X, U, K, d, a)
        Xs.append(Xn); Us.append(Un); Js.append(Jn)
    # tiles at the operating point (what a plugin's linearize/quadraticize return)
    tA = np.array([prob.dynamics.linearize(X[t], U[t])[0] for t in range(T)])
    tB = np.array([prob.dynamics.linearize(X[t], U[t])[1] for t in range(T)])
    q = [prob.game_cost.quadraticize(X[t], U[t]) for t in range(T)]
    q.append(prob.game_cost.quadraticize(X[T], np.zeros(U.shape[1]), terminal=True))
    out.update(X=X, U=U, mu=np.array(mu), K=K, d=d, alphas=alphas.astype(np.float64),
               X_fwd=np.array(Xs), U_fwd=np.array(Us), J_fwd=np.array(Js),
               tile_A=tA, tile_B=tB,
               tile_Lx=np.array([x[0] for x in q]), tile_Lu=np.array([x[1] for x in q]),
               tile_Lxx=np.array([x[2] for x in q]), tile_Luu=np.array([x[3] for x in q]),
               tile_Lux=np.array([x[4] for x in q]))
    np.savez_compressed(OUT / f"g3_passes_{tag}.npz", **out)


def g3_passes():
    DI, UNI, Q6, H6, HL6, DI6, CAR, Q12 = (dp.DoubleIntDynamics4D, dp.UnicycleDynamics4D, dp.QuadcopterDynamics6D,
                                            dp.HumanDynamics6D, dp.HumanDynamicsLin6D, dp.DoubleIntDynamics6D,
                                            dp.CarDynamics3D, dp.QuadcopterDynamics12D)
    g3_case("cfg1_di4d_k3", [DI] * 3, 50, 3, [2] * 3)
    g3_case("cfg2_di4d_k5", [DI] * 5, 50, 0, [2] * 5)
    g3_case("uni4d_k3", [UNI] * 3, 30, 5, [2] * 3)
    g3_case("quad6d_k3", [Q6] * 3, 30, 7, [3] * 3)
    # mixed Quad6D + Human6D (examples.py:74-131 pattern): n_dims [3,3,2]
    x0 = np.array([[0.5, 1.5, 1, 0, 0, 0], [2.4, 1.2, 1.1, 0, 0, 0], [1.2, 0.4, 1.2, 0.1, 0, 0]], float)
    xf = np.array([[2.5, 1.5, 1, 0, 0, 0], [0.3, 1.4, 1.2, 0, 0, 0], [1.3, 2.4, 1.2, 0, 0, 0]], float)
    g3_case("mixed_q6h6", [Q6, Q6, H6], 30, 0, [3, 3, 2], x0=x0.reshape(-1), xf=xf.reshape(-1))
    # small cases of the remaining models, so every Model value appears in a pass
    g3_case("di6d_hlin6d_k2", [DI6, HL6], 20, 11, [3, 3], warm_iters=1)
    g3_case("car3d_k2", [CAR] * 2, 20, 12, [2, 2], warm_iters=1)
    g3_case("quad12d_k2", [Q12] * 2, 10, 13, [3, 3], warm_iters=1)
    # k=1: random_setup would centre the single agent on the origin and divide by zero
    g3_case("di4d_k1", [DI], 20, 14, [2], warm_iters=1,
            x0=np.array([1.5, -0.7, 0.2, 0.1]), xf=np.array([-0.8, 1.1, 0.0, 0.0]))


# --------------------------------------------------------------------------- G4
def g4_solves():
    DI, UNI, Q6, H6 = dp.DoubleIntDynamics4D, dp.UnicycleDynamics4D, dp.QuadcopterDynamics6D, dp.HumanDynamics6D
    # cfg2 seeds: 17,19,36 end by line-search failure; 0,26,29 take ~10 iterations
    out = {}
    seeds = [0, 1, 2, 3, 17, 19, 26, 29, 36, 5, 8, 13]
    T = 50
    for s in seeds:
        prob, meta = analysis_problem(DI, 5, s)
        U0 = np.zeros((T, 10))
        r = traced_solve(prob, meta["x0"], U0, T)
        for k_, v in r.items():
            if k_ in ("K_last", "d_last") and s not in (0, 17):
                continue
            out[f"s{s}_{k_}"] = v
        out[f"s{s}_x0"] = meta["x0"]; out[f"s{s}_xf"] = meta["xf"]
    out["seeds"] = np.array(seeds); out["T"] = np.array(T)
    np.savez_compressed(OUT / "g4_solves_cfg2.npz", **out)

    out = {}
    cases = [("cfg1", [DI] * 3, 50, 4, [2] * 3), ("uni_k3", [UNI] * 3, 40, 6, [2] * 3),
             ("uni_k4", [UNI] * 4, 40, 2, [2] * 4),
             ("quad_k3", [Q6] * 3, 40, 9, [3] * 3), ("quad_k5", [Q6] * 5, 30, 1, [3] * 5),
             ("di_k1", [DI], 50, 10, [2])]
    for tag, mcs, T, seed, n_dims in cases:
        k = len(mcs); n_s = mcs[0](0.1).n_x
        np.random.seed(seed)
        if k == 1:  # random_setup degenerates for one agent (energy normalisation 0/0)
            x0, xf = np.array([1.5, -0.7, 0.2, 0.1]), np.array([-0.8, 1.1, 0.0, 0.0])
        else:
            x0, xf = dp.random_setup(k, n_s, is_rotation=False, rel_dist=k, var=k / 2, n_d=max(n_dims),
                                     random=True, energy=10.0)
        prob, meta = build_problem(mcs, x0, xf, 0.5, 0.1, n_dims)
        U0 = warm_U(mcs, T)
        r = traced_solve(prob, meta["x0"], U0, T)
        r.pop("K_last"); r.pop("d_last")
        for k_, v in {**meta, **r, "U0": U0, "T": np.array(T)}.items():
            out[f"{tag}_{k_}"] = v
    # mixed
    x0 = np.array([[0.5, 1.5, 1, 0, 0, 0], [2.4, 1.2, 1.1, 0, 0, 0], [1.2, 0.4, 1.2, 0.1, 0, 0]], float)
    xf = np.array([[2.5, 1.5, 1, 0, 0, 0], [0.3, 1.4, 1.2, 0, 0, 0], [1.3, 2.4, 1.2, 0, 0, 0]], float)
    mcs = [Q6, Q6, H6]; T = 40
    prob, meta = build_problem(mcs, x0.reshape(-1), xf.reshape(-1), 0.5, 0.1, [3, 3, 2])
    U0 = warm_U(mcs, T)
    r = traced_solve(prob, meta["x0"], U0, T); r.pop("K_last"); r.pop("d_last")
    for k_, v in {**meta, **r, "U0": U0, "T": np.array(T)}.items():
        out[f"mixed_{k_}"] = v
    out["tags"] = np.array(["cfg1", "uni_k3", "uni_k4", "quad_k3", "quad_k5", "di_k1", "mixed"])
    np.savez_compressed(OUT / "g4_solves_misc.npz", **out)


# --------------------------------------------------------------------------- G5
def graph_to_arrays(graph, ids):
    k = len(ids)
    adj = np.zeros((k, k), dtype=np.int32)
    for i, id_ in enumerate(ids):
        for j in graph[id_]:
            adj[i, ids.index(int(j))] = 1
    return adj


def g5_dispatch():
    out = {}
    UNI, Q6 = dp.UnicycleDynamics4D, dp.QuadcopterDynamics6D
    for tag, cls, k, T, seed in (("uni5", UNI, 5, 40, 3), ("quad10", Q6, 10, 75, 0), ("uni8", UNI, 8, 30, 1)):
        prob, meta = analysis_problem(cls, k, seed)
        ids = [int(i) for i in meta["ids"]]
        U0 = warm_U([cls] * k, T)
        X0row = meta["x0"].reshape(1, -1)
        g1 = dp.define_inter_graph_threshold(X0row, 0.5, prob.game_cost.x_dims, ids)
        with redirect_stdout(io.StringIO()):
            Xd, Ud, Jf, info = dp.solve_distributed(prob, X0row, U0, 0.5, ignore_ids=[], verbose=False)
        g2 = dp.define_inter_graph_threshold(Xd, 0.5, prob.game_cost.x_dims, ids)
        xs = dp.split_graph(X0row, prob.game_cost.x_dims, g1)
        for k_, v in meta.items():
            out[f"{tag}_{k_}"] = v
        out.update({f"{tag}_T": np.array(T), f"{tag}_U0": U0, f"{tag}_adj_x0": graph_to_arrays(g1, ids),
                    f"{tag}_adj_traj": graph_to_arrays(g2, ids), f"{tag}_X_dec": Xd, f"{tag}_U_dec": Ud,
                    f"{tag}_J_full": np.array(Jf)})
        for i, x in enumerate(xs):
            out[f"{tag}_x0split_{i}"] = x
        if tag != "quad10":
            # second call seeded with the first result (the receding-horizon pattern)
            with redirect_stdout(io.StringIO()):
                Xd2, Ud2, Jf2, _ = dp.solve_distributed(prob, Xd, Ud, 0.5, ignore_ids=[], verbose=False)
            out.update({f"{tag}_X_dec2": Xd2, f"{tag}_U_dec2": Ud2, f"{tag}_J_full2": np.array(Jf2)})
    np.savez_compressed(OUT / "g5_dispatch.npz", **out)


# --------------------------------------------------------------------------- G6
def g6_scenarios():
    out = {}
    for tag, k, n_s, n_d in (("k5s4", 5, 4, 2), ("k3s4", 3, 4, 2), ("k10s6", 10, 6, 3), ("k15s4", 15, 4, 2)):
        x0s, xfs = [], []
        for s in range(8):
            np.random.seed(s)
            x0, xf = dp.random_setup(k, n_s, is_rotation=False, rel_dist=k, var=k / 2, n_d=n_d,
                                     random=True, energy=10.0)
            x0s.append(x0.reshape(-1)); xfs.append(xf.reshape(-1))
        out[f"{tag}_x0"] = np.array(x0s); out[f"{tag}_xf"] = np.array(xfs)
    np.savez_compressed(OUT / "g6_scenarios.npz", **out)


# --------------------------------------------------------------------------- G7
class _Rows(__import__("logging").Handler):
    def __init__(self):
        super().__init__()
        self.rows = []

    def emit(self, record):
        self.rows.append(record.getMessage())


def g7_callers():
    """The callers either side of the solve: solve_rhc (distributed.py:106-221), ilqrProblem.selfish_warmstart
    (problem.py:66-91), solve_subproblem (problem.py:97-105)."""
    import logging
    out = {}
    DI, UNI = dp.DoubleIntDynamics4D, dp.UnicycleDynamics4D
    log = logging.getLogger(); log.setLevel(logging.INFO)
    for tag, cls, k, N, seed, centralized, kw in (
            ("rhc_c", DI, 3, 15, 4, True, dict(step_size=5, dist_converge=0.5, t_diverge=3.0)),
            # (J_converge cannot be pinned: the reference raises NameError there -- n_agents / n_states are only bound
            #  in the dist_converge branch, distributed.py:134-144 vs :131,189)
            ("rhc_c2", DI, 5, 20, 11, True, dict(step_size=1, dist_converge=1.5, t_diverge=0.6)),
            ("rhc_d", UNI, 6, 15, 1, False, dict(step_size=4, dist_converge=0.4, t_diverge=2.0))):
        prob, meta = analysis_problem(cls, k, seed)
        np.random.seed(1000 + seed)
        U_warm = np.random.rand(N, k * int(meta["n_c"])) * 0.01       # what solve_rhc is about to draw (:152)
        np.random.seed(1000 + seed)
        h = _Rows(); log.addHandler(h)
        args = () if centralized else (0.5, [])                      # radius, ignore_ids=[] (quirk Q9)
        with redirect_stdout(io.StringIO()):
            Xf, Uf, Jf = dp.solve_rhc(prob, meta["x0"], N, *args, centralized=centralized, i_trial=7, **kw)
        log.removeHandler(h)
        for k_, v in meta.items():
            out[f"{tag}_{k_}"] = v
        out.update({f"{tag}_N": np.array(N), f"{tag}_np_seed": np.array(1000 + seed), f"{tag}_U_warm": U_warm,
                    f"{tag}_X_full": Xf, f"{tag}_U_full": Uf, f"{tag}_J_full": np.array(Jf),
                    f"{tag}_rows": np.array(h.rows), f"{tag}_centralized": np.array(centralized)})
        for k_, v in kw.items():
            out[f"{tag}_kw_{k_}"] = np.array(v)
    # selfish_warmstart
    for tag, cls, k, N, seed in (("ws_uni", UNI, 3, 20, 5), ("ws_di", DI, 4, 25, 9)):
        prob, meta = analysis_problem(cls, k, seed)
        with redirect_stdout(io.StringIO()):
            Uw = prob.selfish_warmstart(meta["x0"], N)
        for k_, v in meta.items():
            out[f"{tag}_{k_}"] = v
        out.update({f"{tag}_N": np.array(N), f"{tag}_U_warm": Uw})
    # solve_subproblem on the sub-problems of a proximity graph
    prob, meta = analysis_problem(UNI, 5, 3)
    ids = [int(i) for i in meta["ids"]]
    T = 30
    U0 = np.zeros((T, 10))
    graph = dp.define_inter_graph_threshold(meta["x0"].reshape(1, -1), 0.5, prob.game_cost.x_dims, ids)
    subs = prob.split(graph)
    x0s = dp.split_graph(meta["x0"].reshape(1, -1), prob.game_cost.x_dims, graph)
    U0s = dp.split_graph(U0, prob.game_cost.u_dims, graph)
    for k_, v in meta.items():
        out[f"sub_{k_}"] = v
    out["sub_T"] = np.array(T); out["sub_adj"] = graph_to_arrays(graph, ids)
    for i, id_ in enumerate(ids):
        Xi, Ui, rid = dp.problem.solve_subproblem((subs[i], x0s[i], U0s[i], id_, False))
        assert rid == id_
        out[f"sub_X_{i}"] = Xi; out[f"sub_U_{i}"] = Ui
    np.savez_compressed(OUT / "g7_callers.npz", **out)


# --------------------------------------------------------------------------- G8
def g8_hetero():
    """BASELINE config 5's heterogeneous team (QuadcopterDynamics12D + the zero-padded human)."""
    Q12, HP = dp.QuadcopterDynamics12D, HumanDynamics6DPadded12
    # the padded model alone, like G1
    out = {}
    rng = np.random.default_rng(1008)
    m = HP(0.1, 500)
    xs, us, dts, fs, its, As, Bs = [], [], [], [], [], [], []
    for i in range(8):
        x = np.r_[rng.uniform(-2, 2, 6), np.zeros(6)]; u = rng.uniform(-1, 1, 4)
        if i >= 4:
            x[6:] = rng.uniform(-1, 1, 6)      # the padding is carried through untouched whatever it holds
        dt = (0.05, 0.1)[i % 2]
        m = HP(dt, 500)
        A, B = m.linearize(x, u)
        xs.append(x); us.append(u); dts.append(dt); fs.append(m.f(x, u)); its.append(m(x, u)); As.append(A); Bs.append(B)
    out.update(HumanPad12D_x=np.array(xs), HumanPad12D_u=np.array(us), HumanPad12D_dt=np.array(dts),
               HumanPad12D_f=np.array(fs), HumanPad12D_integrate=np.array(its), HumanPad12D_A=np.array(As),
               HumanPad12D_B=np.array(Bs))
    np.savez_compressed(OUT / "g8_hetero_model.npz", **out)

    # k = 3: passes (G3 form, with tiles) and a traced solve (G4 form)
    mcs = [Q12, Q12, HP]
    np.random.seed(21)
    x0, xf = dp.random_setup(3, 12, is_rotation=False, rel_dist=3, var=1.5, n_d=3, random=True, energy=3.0)
    g3_case("hetero_k3", mcs, 20, 21, [3, 3, 2], x0=x0.reshape(-1), xf=xf.reshape(-1), warm_iters=2)
    (OUT / "g3_passes_hetero_k3.npz").rename(OUT / "g8_hetero_k3_passes.npz")
    prob, meta = build_problem(mcs, x0.reshape(-1), xf.reshape(-1), 0.5, 0.1, [3, 3, 2])
    U0 = warm_U(mcs, 20)
    r = traced_solve(prob, meta["x0"], U0, 20, n_lqr_iter=12); r.pop("K_last"); r.pop("d_last")
    np.savez_compressed(OUT / "g8_hetero_k3_solve.npz", **{**meta, **r, "U0": U0, "T": np.array(20)})

    # cfg5 at its stated size: 20 agents (14 quadcopters + 6 humans), T = 150, n_x = 240, n_u = 80.  One backward pass and
    # the ten forward passes at an operating point one iLQR iteration away from the hover rollout; the gains are 23 MB,
    # so the fixture keeps K at three steps, all of d, every J and the accepted candidate's trajectory at every tenth step.
    k, T = 20, 150
    mcs = [Q12] * 14 + [HP] * 6
    nd = [3] * 14 + [2] * 6
    np.random.seed(55)
    # energy 100 (5 per agent): with analysis.py's 10 the twenty agents start within 0.5 of each other and every
    # line-search candidate of the reference's first iteration overflows (J = nan)
    x0, xf = dp.random_setup(k, 12, is_rotation=False, rel_dist=k, var=k / 2, n_d=3, random=True, energy=100.0)
    prob, meta = build_problem(mcs, x0.reshape(-1), xf.reshape(-1), 0.5, 0.1, nd)
    U0 = warm_U(mcs, T)
    s = dp.ilqrSolver(prob, T)
    Xr, Jr = s._rollout(meta["x0"].reshape(-1, 1), U0)
    X, U, _ = s.solve(meta["x0"].copy(), U0.copy(), n_lqr_iter=1, verbose=False)
    mu = s.μ
    K, d = s._backward_pass(X, U)
    alphas = 1.1 ** (-np.arange(10, dtype=np.float32) ** 2)
    _, J_star = s._rollout(meta["x0"].reshape(-1, 1), U)
    Js, X0f, acc = [], None, -1
    for i, a in enumerate(alphas):
        Xn, Un, Jn = s._forward_pass(X, U, K, d, a)
        Js.append(Jn)
        if acc < 0 and Jn < J_star:     # the candidate the line search accepts (control.py:183)
            X0f, U0f, acc = Xn, Un, i
    keep = np.array([0, 75, 149])
    out = dict(meta)
    out.update(T=np.array(T), U0=U0, X_roll_every10=Xr[::10], J_roll=np.array(Jr), X=X, U=U, mu=np.array(mu),
               K_steps=keep, K_kept=K[keep], d=d, alphas=alphas.astype(np.float64), J_fwd=np.array(Js),
               J_star=np.array(J_star), acc=np.array(acc), X_fwd_acc_every10=X0f[::10], U_fwd_acc_every10=U0f[::10])
    np.savez_compressed(OUT / "g8_hetero_k20.npz", **out)


# --------------------------------------------------------------------------- G9
# Does the reference determine its own result?  For items the GPU decides differently from the CPU oracle (seeds found on
# the GPU box by scripts/find_flips.py; the lists are copied into the fixture) the REAL reference solves x0 and 32 copies
# of x0 perturbed by 1e-14 .. 5e-13 (oracle/parity.py's ensemble): per member the decision trace, the accepted costs, the
# returned J and how far its final trajectory lies from the unperturbed member's.
G9_FAMILIES = {   # name: (model class name, k, T, hover warm start, how many flipped items, how many controls)
    "cfg2": ("DoubleIntDynamics4D", 5, 50, False, 48, 4),
    "uni8": ("UnicycleDynamics4D", 8, 100, False, 8, 2),
    "quad10": ("QuadcopterDynamics6D", 10, 75, True, 5, 1),
}
G9_DELTAS = tuple(float(sg * v) for v in np.geomspace(1e-14, 5e-13, 16) for sg in (1.0, -1.0))   # = oracle/parity.py DELTAS


def _g9_member(task):
    fam, seed, delta = task
    cls_name, k, T, hover, _, _ = G9_FAMILIES[fam]
    cls = getattr(dp, cls_name)
    prob, meta = analysis_problem(cls, k, seed)
    U0 = warm_U([cls] * k, T) if hover else np.zeros((T, k * cls(0.1).n_u))
    x0 = meta["x0"] * (1.0 + delta)
    with redirect_stdout(io.StringIO()):
        r = traced_solve(prob, x0, U0, T)
    return fam, seed, delta, r, meta


def g9_chaos(flips_dir=None):
    import multiprocessing as mp
    flips_dir = Path(flips_dir) if flips_dir else OUT.parent.parent / "gpurun_out" / "flips"
    tasks, chosen = [], {}
    for fam, (_, k, T, hover, n_flip, n_ctl) in G9_FAMILIES.items():
        z = np.load(flips_dir / f"{fam}.npz")
        fl = np.nonzero(z["flipped"])[0]; ct = np.nonzero(~z["flipped"])[0]
        # spread the flipped items over the seed range instead of taking the first ones
        take = np.concatenate([fl[np.linspace(0, len(fl) - 1, min(n_flip, len(fl))).round().astype(int)], ct[:n_ctl]])
        chosen[fam] = (z, take)
        for i in take:
            for delta in (0.0,) + G9_DELTAS:
                tasks.append((fam, int(z["seeds"][i]), delta))
    with mp.Pool(8) as pool:
        results = pool.map(_g9_member, tasks, chunksize=1)
    by = {}
    for fam, seed, delta, r, meta in results:
        by.setdefault((fam, seed), {})[delta] = (r, meta)
    ROWS = 50
    for fam, (z, take) in chosen.items():
        seeds = [int(z["seeds"][i]) for i in take]
        M = 1 + len(G9_DELTAS)
        n_bwd = np.zeros((len(seeds), M), dtype=np.int16); acc = np.full((len(seeds), M, ROWS), -9, dtype=np.int8)
        Jstar = np.full((len(seeds), M, ROWS), np.nan); Jlast = np.full((len(seeds), M, ROWS), np.nan)
        J = np.zeros((len(seeds), M)); J0 = np.zeros((len(seeds), M)); dX = np.zeros((len(seeds), M)); dU = np.zeros((len(seeds), M))
        Xb, Ub, x0s, xfs = [], [], [], []
        for a, seed in enumerate(seeds):
            base, meta = by[(fam, seed)][0.0]
            Xb.append(base["X"]); Ub.append(base["U"]); x0s.append(meta["x0"]); xfs.append(meta["xf"])
            for b, delta in enumerate((0.0,) + G9_DELTAS):
                r, _ = by[(fam, seed)][delta]
                nb = len(r["mu_trace"])
                n_bwd[a, b] = nb; acc[a, b, :nb] = r["acc_trace"]; Jstar[a, b, :nb] = r["Jstar_trace"]; Jlast[a, b, :nb] = r["Jlast_trace"]
                J[a, b] = r["J"]; J0[a, b] = r["J0"]
                dX[a, b] = np.max(np.abs(r["X"] - base["X"])) / np.max(np.abs(base["X"]))
                dU[a, b] = np.max(np.abs(r["U"] - base["U"])) / max(np.max(np.abs(base["U"])), 1e-300)
        cls_name, k, T, hover, _, _ = G9_FAMILIES[fam]
        np.savez_compressed(OUT / f"g9_chaos_{fam}.npz", seeds=np.array(seeds), flipped_on_gpu=z["flipped"][take],
                            deltas=np.array((0.0,) + G9_DELTAS), n_bwd=n_bwd, acc_trace=acc, Jstar_trace=Jstar.astype(np.float32),
                            J=J, J0=J0, dX_vs_base=dX, dU_vs_base=dU, X_base=np.array(Xb), x0=np.array(x0s),
                            xf=np.array(xfs), model=np.array(cls_name), k=np.array(k), T=np.array(T), hover=np.array(hover),
                            gpu_n_bwd_when_found=z["gpu_n_bwd"][take], gpu_accept_when_found=z["gpu_accept"][take])
        same = (n_bwd == n_bwd[:, :1]).all(axis=1) & (acc == acc[:, :1]).all(axis=(1, 2))
        print(f"g9 {fam}: {len(seeds)} items; the reference's own 33 members take ONE decision trace on {int(same.sum())} of them; "
              f"max trajectory spread among members {dX.max():.2e}", flush=True)


# --------------------------------------------------------------------------- G10
def g10_harness():
    """The Monte-Carlo harness itself (scripts/analysis.py:35-107): the REAL multi_agent_run, imported from where it lies and run
    unmodified on a seeded global stream, for a few (model, team size, trial) cells.  Stored per trial: the seed, every draw of
    the trial in the order the reference makes it -- (x0, xf) of random_setup, the centralized branch's warm start, the
    distributed branch's (captured from np.random.rand itself, not re-derived) -- both branches' returned trajectories, and the
    rows the reference logs.

    One shim, the same as G7's: the distributed branch of the reference as shipped raises (quirk Q9: multi_agent_run passes no
    ignore_ids and solve_distributed's default None is not iterable), so `solve_rhc` as seen by the analysis module appends
    ignore_ids=[] to the positional arguments of the distributed call.  Nothing else is touched: loop order, RNG consumption,
    weights, ids, STEP_SIZE and the logging are the reference's own code running."""
    import importlib.util
    import logging
    spec = importlib.util.spec_from_file_location("ref_scripts_analysis", REF / "scripts" / "analysis.py")
    ana = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ana)
    real_rhc = ana.solve_rhc
    multi_agent_run_ret = {}

    def rhc_q9(problem, x0, N, *args, centralized=True, **kw):
        a = tuple(args) if centralized else tuple(args) + ([],)
        r = real_rhc(problem, x0, N, *a, centralized=centralized, **kw)
        multi_agent_run_ret[bool(centralized)] = r
        return r

    ana.solve_rhc = rhc_q9
    models = [dp.DoubleIntDynamics4D, dp.UnicycleDynamics4D, dp.QuadcopterDynamics6D]        # analysis.py:133-137
    log = logging.getLogger(); log.setLevel(logging.INFO)
    out = {}
    dt, N, radius, energy, seed0 = 0.1, 20, 0.5, 10.0, 4
    kw = dict(t_kill=None, dist_converge=0.1, t_diverge=3.0)
    cells = [(dp.DoubleIntDynamics4D, 3, (0, 1)), (dp.UnicycleDynamics4D, 4, (0, 2)), (dp.QuadcopterDynamics6D, 3, (1,))]
    tags = []
    real_rand = np.random.rand
    for model, n_agents, trials in cells:
        n_d = 3 if model is dp.QuadcopterDynamics6D else 2
        n_states = model(-1).n_x
        for i_trial in trials:
            # the harness's documented seed of a trial (dpilqr_amd/analysis.py: seed_of), restated here
            seed = seed0 + 100003 * models.index(model) + 1009 * n_agents + i_trial
            tag = f"{model.__name__}_{n_agents}_{i_trial}"
            drawn = []

            def recording_rand(*shape):
                v = real_rand(*shape)
                drawn.append(np.array(v))
                return v

            h = _Rows(); log.addHandler(h)
            np.random.seed(seed)
            np.random.rand = recording_rand
            try:
                with redirect_stdout(io.StringIO()):
                    ana.multi_agent_run(model, [n_states] * n_agents, dt, N, radius, n_d=n_d, energy=energy, i_trial=i_trial,
                                        verbose=False, **kw)
            finally:
                np.random.rand = real_rand
                log.removeHandler(h)
            after = np.random.get_state()[1][:4].copy()        # where the trial left the stream
            warm = [v for v in drawn if v.shape == (N, n_agents * model(-1).n_u)]
            assert len(warm) == 2, [v.shape for v in drawn]
            np.random.seed(seed)
            x0, xf = dp.random_setup(n_agents, n_states, is_rotation=False, rel_dist=n_agents, var=n_agents / 2, n_d=n_d,
                                     random=True, energy=energy)        # (the trial's first draw, made again for the record)
            (Xc, Uc, Jc), (Xd, Ud, Jd) = multi_agent_run_ret[True], multi_agent_run_ret[False]
            assert np.array_equal(Xc[0], x0.ravel()) and np.array_equal(Xd[0], x0.ravel())
            out.update({f"{tag}_seed": np.array(seed), f"{tag}_x0": x0, f"{tag}_xf": xf, f"{tag}_U_c": warm[0] * 0.01,
                        f"{tag}_U_d": warm[1] * 0.01, f"{tag}_rows": np.array(h.rows), f"{tag}_Xc": Xc, f"{tag}_Uc": Uc,
                        f"{tag}_Jc": np.array(Jc), f"{tag}_Xd": Xd, f"{tag}_Ud": Ud, f"{tag}_Jd": np.array(Jd),
                        f"{tag}_stream_after": after})
            tags.append(tag)
            print(f"g10 {tag}: seed {seed}, {len(h.rows)} rows, Jc {Jc:.6g}, Jd {Jd:.6g}", flush=True)
    out.update(tags=np.array(tags), dt=np.array(dt), N=np.array(N), radius=np.array(radius), energy=np.array(energy),
               seed0=np.array(seed0), dist_converge=np.array(kw["dist_converge"]), t_diverge=np.array(kw["t_diverge"]))
    np.savez_compressed(OUT / "g10_harness.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8"]
    for w in which:
        print("generating", w, flush=True)
        {"g1": g1_models, "g2": g2_costs, "g3": g3_passes, "g4": g4_solves,
         "g5": g5_dispatch, "g6": g6_scenarios, "g7": g7_callers, "g8": g8_hetero, "g9": g9_chaos, "g10": g10_harness}[w]()
    for f in sorted(OUT.glob("*.npz")):
        print(f"{f.name:40s} {f.stat().st_size/1024:8.1f} KiB")
