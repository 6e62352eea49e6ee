"""NumPy restatement of the reference's iLQR hot path -- TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and
tests/test_oracle_golden.py; the product never imports it).

SURVEY.md 8(d) asks for two CPU baselines beside the GPU number: the C restatement on all usable host cores
(oracle/ilqr_oracle.c) and THIS one -- the same per-step Python structure as the reference (a Python loop over the horizon
calling per-agent dynamics through an FFI, NumPy for the cost derivatives, np.linalg.solve twice per step), hence the
reference's own speed class, single process.  The reference itself cannot travel to the GPU box; this file is written
from the mathematics of SURVEY.md Appendix A and pinned to the reference's outputs by the golden solves G4
(tests/test_oracle_golden.py::test_numpy_port_*).

    control.py:80-93    rollout            -> Solver.rollout
    control.py:95-114   _forward_pass      -> Solver.forward_pass
    control.py:116-148  _backward_pass     -> Solver.backward_pass
    control.py:150-225  solve              -> Solver.solve
    cost.py:79-101, 117-171, 197-239, 269-315 -> GameCost
    dynamics.py:159-186 MultiDynamicalModel -> JointModel (per-agent calls into oracle/liboracle.so, as the reference's
                        CppModel calls bbdynamicswrap: bbdynamicswrap.pyx:93-164)
"""
import itertools

import numpy as np

from . import oracle as orc

ALPHAS = 1.1 ** (-np.arange(10, dtype=np.float32) ** 2)      # control.py:162 (float32: quirk Q1)


class JointModel:
    def __init__(self, models, dt):
        self.models = [int(m) for m in models]
        self.dt = float(dt)
        self.n_s, self.n_c = orc.MODEL_DIMS[self.models[0]]
        self.k = len(self.models)
        self.n_x, self.n_u = self.k * self.n_s, self.k * self.n_c

    def __call__(self, x, u):
        ns, nc = self.n_s, self.n_c
        return np.concatenate([orc.model_integrate(m, x[i * ns:(i + 1) * ns], u[i * nc:(i + 1) * nc], self.dt)
                               for i, m in enumerate(self.models)])

    def linearize(self, x, u):
        ns, nc = self.n_s, self.n_c
        A = np.zeros((self.n_x, self.n_x)); B = np.zeros((self.n_x, self.n_u))
        for i, m in enumerate(self.models):
            Ai, Bi = orc.model_linearize(m, x[i * ns:(i + 1) * ns], u[i * nc:(i + 1) * nc], self.dt)
            A[i * ns:(i + 1) * ns, i * ns:(i + 1) * ns] = Ai
            B[i * ns:(i + 1) * ns, i * nc:(i + 1) * nc] = Bi
        return A, B


class GameCost:
    REF_WEIGHT, PROX_WEIGHT = 1.0, 200.0

    def __init__(self, xf, Q, R, Qf, radius, n_dims, n_s, n_c):
        self.k = len(n_dims)
        self.n_s, self.n_c = n_s, n_c
        bc = lambda M, n: np.broadcast_to(np.asarray(M, dtype=np.float64), (self.k, n, n))
        self.xf = np.asarray(xf, dtype=np.float64).reshape(self.k, n_s)
        self.Q, self.R, self.Qf = bc(Q, n_s), bc(R, n_c), bc(Qf, n_s)
        self.radius = float(radius)
        self.n_dims = [int(v) for v in n_dims]
        self.pairs = list(itertools.combinations(range(self.k), 2))
        self.uniform = len(set(self.n_dims)) == 1

    def _prox(self, x):
        if self.k == 1:
            return 0.0
        xs = x.reshape(self.k, self.n_s)
        tot = 0.0
        for i, j in self.pairs:
            nd = 2 if self.uniform else min(self.n_dims[i], self.n_dims[j])        # quirk Q5: planar when homogeneous
            d = np.linalg.norm(xs[i, :nd] - xs[j, :nd])
            tot += np.fmin(0.0, d - self.radius) ** 2
        return tot

    def __call__(self, x, u, terminal=False):
        xs, us = x.reshape(self.k, self.n_s), u.reshape(self.k, self.n_c)
        ref = 0.0
        for i in range(self.k):
            e = xs[i] - self.xf[i]
            ref += e @ self.Qf[i] @ e if terminal else e @ self.Q[i] @ e + us[i] @ self.R[i] @ us[i]
        return self.PROX_WEIGHT * self._prox(x) + self.REF_WEIGHT * ref

    def quadraticize(self, x, u, terminal=False):
        k, ns, nc = self.k, self.n_s, self.n_c
        n, m = k * ns, k * nc
        xs, us = x.reshape(k, ns), u.reshape(k, nc)
        Lx, Lu = np.zeros(n), np.zeros(m)
        Lxx, Luu, Lux = np.zeros((n, n)), np.zeros((m, m)), np.zeros((m, n))
        for i in range(k):
            M = self.Qf[i] if terminal else self.Q[i]
            e = xs[i] - self.xf[i]
            Lx[i * ns:(i + 1) * ns] = e @ (M + M.T)
            Lxx[i * ns:(i + 1) * ns, i * ns:(i + 1) * ns] = M + M.T
            if not terminal:
                Lu[i * nc:(i + 1) * nc] = us[i] @ (self.R[i] + self.R[i].T)
                Luu[i * nc:(i + 1) * nc, i * nc:(i + 1) * nc] = self.R[i] + self.R[i].T
        Lx *= self.REF_WEIGHT; Lu *= self.REF_WEIGHT; Lxx *= self.REF_WEIGHT; Luu *= self.REF_WEIGHT
        if k > 1:
            gx, gxx = np.zeros(n), np.zeros((n, n))
            r = self.radius
            for i, j in self.pairs:
                nd = min(self.n_dims[i], self.n_dims[j])
                dl = xs[i, :nd] - xs[j, :nd]
                d = np.sqrt(np.sum(dl ** 2))
                if d > r:
                    continue
                g = 2.0 * (d - r) / d * dl
                H = 2.0 * r * np.outer(dl, dl) / d ** 3
                H[np.diag_indices(nd)] = 2.0 * r * dl ** 2 / d ** 3 - 2.0 * r / d + 2.0
                si, sj = slice(i * ns, i * ns + nd), slice(j * ns, j * ns + nd)
                gx[si] += g; gx[sj] -= g
                gxx[si, si] += H; gxx[sj, sj] += H; gxx[si, sj] -= H; gxx[sj, si] -= H
            Lx += self.PROX_WEIGHT * gx; Lxx += self.PROX_WEIGHT * gxx
        return Lx, Lu, Lxx, Luu, Lux


class Solver:
    """ilqrSolver of control.py:53-242 on a JointModel and a GameCost."""

    def __init__(self, dynamics, cost, N):
        self.dynamics, self.cost, self.N = dynamics, cost, int(N)
        self.n_x, self.n_u = dynamics.n_x, dynamics.n_u
        self.mu, self.delta = 1.0, 2.0

    def rollout(self, x0, U):
        X = np.zeros((self.N + 1, self.n_x)); X[0] = x0
        J = 0.0
        for t in range(self.N):
            X[t + 1] = self.dynamics(X[t], U[t])
            J += self.cost(X[t], U[t])
        return X, J + self.cost(X[-1], np.zeros(self.n_u), terminal=True)

    def forward_pass(self, X, U, K, d, alpha):
        Xn, Un = np.zeros_like(X), np.zeros_like(U)
        Xn[0] = X[0]
        J = 0.0
        for t in range(self.N):
            Un[t] = U[t] + K[t] @ (Xn[t] - X[t]) + alpha * d[t]
            Xn[t + 1] = self.dynamics(Xn[t], Un[t])
            J += self.cost(Xn[t], Un[t])
        return Xn, Un, J + self.cost(Xn[-1], np.zeros(self.n_u), terminal=True)

    def backward_pass(self, X, U):
        n, m = self.n_x, self.n_u
        K = np.zeros((self.N, m, n)); d = np.zeros((self.N, m))
        reg = self.mu * np.eye(n)
        p, _, P, _, _ = self.cost.quadraticize(X[-1], np.zeros(m), terminal=True)
        for t in range(self.N - 1, -1, -1):
            Lx, Lu, Lxx, Luu, Lux = self.cost.quadraticize(X[t], U[t])
            A, B = self.dynamics.linearize(X[t], U[t])
            Qx = Lx + A.T @ p
            Qu = Lu + B.T @ p
            Qxx = Lxx + A.T @ P @ A
            Quu = Luu + B.T @ (P + reg) @ B
            Qux = Lux + B.T @ (P + reg) @ A
            K[t] = -np.linalg.solve(Quu, Qux)
            d[t] = -np.linalg.solve(Quu, Qu)
            p = Qx + K[t].T @ Quu @ d[t] + K[t].T @ Qu + Qux.T @ d[t]
            P = Qxx + K[t].T @ Quu @ K[t] + K[t].T @ Qux + Qux.T @ K[t]
            P = 0.5 * (P + P.T)
        return K, d

    def solve(self, x0, U=None, n_lqr_iter=50, tol=1e-3):
        U = np.zeros((self.N, self.n_u)) if U is None else np.array(U, dtype=np.float64)
        self.mu, self.delta = 1.0, 2.0
        X, J_star = self.rollout(np.asarray(x0, dtype=np.float64).reshape(-1), U)
        J = J_star
        trace = []
        status = 3
        for _ in range(n_lqr_iter):
            mu_before = self.mu
            K, d = self.backward_pass(X, U)
            accept = converged = False
            acc = -1
            for a, alpha in enumerate(ALPHAS):
                Xn, Un, J = self.forward_pass(X, U, K, d, alpha)
                if J < J_star:
                    converged = abs((J_star - J) / J_star) < tol
                    X, U, J_star = Xn, Un, J
                    self.delta = min(1.0, self.delta) / 2.0
                    self.mu *= self.delta
                    if self.mu <= 1e-6:
                        self.mu = 0.0
                    accept, acc = True, a
                    break
            trace.append((mu_before, acc, J, J_star))
            if not accept:
                status = 2
                break
            if converged:
                status = 1
                break
        return dict(X=X, U=U, J=J, status=status, trace=np.array(trace))


def cfg_solver(models, n_dims, xf, Q, R, Qf, radius, dt, T):
    dyn = JointModel(models, dt)
    return Solver(dyn, GameCost(xf, Q, R, Qf, radius, n_dims, dyn.n_s, dyn.n_c), T)
