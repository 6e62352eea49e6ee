"""ilqrSolver: the reference's solver class (dpilqr/control.py:15-249) over the HIP library.

Same constructor, attributes (N, n_x, n_u, dt, cost, dynamics, mu, Delta) and methods (solve, _rollout,
_forward_pass, _backward_pass).  For problems built from the recognised plugin types the whole solve runs
device-resident in one call (dpilqr_solve_batch).  For problems that contain host plugins (user subclasses
of DynamicalModel / Cost) the iteration loop below runs on the host, calls the user's
__call__ / linearize / quadraticize itself, and sends the resulting tile records through the GPU Riccati
sweep -- the plugin surface is unchanged, only the recursion moved.
"""
from time import perf_counter

import numpy as np

from . import _lib
from .batch import backward_pass_tiles, pack_tiles
from .lowering import is_lowerable, lower_problems


class ilqrSolver:
    DELTA_0 = 2.0      # control.py:48-51
    MU_MIN = 1e-6
    MU_MAX = 1e3
    N_LS_ITER = 10

    def __init__(self, problem, N=10):
        self.problem = problem
        self.N = N
        self._batch = None
        self._reset_regularization()

    # ---- the attribute surface of the reference
    cost = property(lambda self: self.problem.game_cost)
    dynamics = property(lambda self: self.problem.dynamics)
    n_x = property(lambda self: self.problem.dynamics.n_x)
    n_u = property(lambda self: self.problem.dynamics.n_u)
    dt = property(lambda self: self.problem.dynamics.dt)

    def _reset_regularization(self):
        self.μ = 1.0
        self.Δ = self.DELTA_0

    def _decrease_regularization(self):
        self.Δ = min(1.0, self.Δ) / self.DELTA_0
        self.μ *= self.Δ
        if self.μ <= self.MU_MIN:
            self.μ = 0.0

    def _increase_regularization(self):
        self.Δ = max(1.0, self.Δ) * self.DELTA_0
        self.μ = max(self.MU_MIN, self.μ * self.Δ)

    # ---- device lowering
    @property
    def on_device(self):
        return is_lowerable(self.problem)

    def _pb(self, T):
        if self._batch is None or self._batch.T != T:
            self._batch = lower_problems([self.problem], T)
        return self._batch

    # ---- passes (each one launch on the device path)
    def _rollout(self, x0, U):
        U = np.asarray(U, dtype=np.float64); x0 = np.asarray(x0, dtype=np.float64).reshape(-1)
        if self.on_device:
            X, J = self._pb(U.shape[0]).rollout(x0[None], U[None])
            return X[0].cpu().numpy(), float(J.item())
        X = np.zeros((U.shape[0] + 1, self.n_x)); X[0] = x0
        J = 0.0
        for t in range(U.shape[0]):
            X[t + 1] = self.dynamics(X[t], U[t])
            J += float(np.asarray(self.cost(X[t], U[t])).item())
        J += float(np.asarray(self.cost(X[-1], np.zeros(self.n_u), terminal=True)).item())
        return X, J

    def _forward_pass(self, X, U, K, d, α):
        if self.on_device:
            Xn, Un, Jn = self._pb(self.N).forward_pass(X[None], U[None], K[None], d[None], [float(α)])
            return Xn[0, 0].cpu().numpy(), Un[0, 0].cpu().numpy(), float(Jn.item())
        Xn = np.zeros((self.N + 1, self.n_x)); Un = np.zeros((self.N, self.n_u))
        Xn[0] = X[0]
        J = 0.0
        for t in range(self.N):
            Un[t] = U[t] + (K[t] @ (Xn[t] - X[t]) + α * d[t])
            Xn[t + 1] = self.dynamics(Xn[t], Un[t])
            J += float(np.asarray(self.cost(Xn[t], Un[t])).item())
        J += float(np.asarray(self.cost(Xn[-1], np.zeros(self.n_u), terminal=True)).item())
        return Xn, Un, J

    def _host_tiles(self, X, U):
        """Call the plugins' linearize / quadraticize at every step: the tile contract of the GPU sweep."""
        T, n, m = self.N, self.n_x, self.n_u
        A = np.zeros((T, n, n)); B = np.zeros((T, n, m))
        Lx = np.zeros((T + 1, n)); Lu = np.zeros((T + 1, m)); Lxx = np.zeros((T + 1, n, n))
        Luu = np.zeros((T + 1, m, m)); Lux = np.zeros((T + 1, m, n))
        for t in range(T):
            A[t], B[t] = self.dynamics.linearize(X[t], U[t])
            Lx[t], Lu[t], Lxx[t], Luu[t], Lux[t] = self.cost.quadraticize(X[t], U[t])
        Lx[T], _, Lxx[T], _, _ = self.cost.quadraticize(X[T], np.zeros(m), terminal=True)
        return pack_tiles(A[None], B[None], Lx[None], Lu[None], Lxx[None], Luu[None], Lux[None])

    def _backward_pass(self, X, U):
        X = np.asarray(X, dtype=np.float64); U = np.asarray(U, dtype=np.float64)
        if self.on_device:
            K, d = self._pb(self.N).backward_pass(X[None], U[None], self.μ)
        else:
            K, d = backward_pass_tiles(self._host_tiles(X, U), 1, self.N, self.n_x, self.n_u, self.μ)
        return K[0].cpu().numpy(), d[0].cpu().numpy()

    # ---- the solve
    def solve(self, x0, U=None, n_lqr_iter=50, tol=1e-3, t_kill=None, verbose=True):
        if U is None:
            U = np.zeros((self.N, self.n_u))
        U = np.asarray(U, dtype=np.float64)
        if U.shape != (self.N, self.n_u):
            raise ValueError
        x0 = np.asarray(x0, dtype=np.float64).reshape(-1)
        self._reset_regularization()
        if self.on_device:      # t_kill is honoured on the device (its own clock, include/dpilqr_hip.h section 5)
            return self._solve_device(x0, U, n_lqr_iter, tol, verbose, t_kill)
        return self._solve_host_loop(x0, U, n_lqr_iter, tol, t_kill, verbose)

    def _solve_device(self, x0, U, n_lqr_iter, tol, verbose, t_kill=None):
        r = self._pb(self.N).solve(x0[None], U[None], n_lqr_iter=n_lqr_iter, tol=tol, trace=True, t_kill=t_kill)
        tr = r["trace"][0].cpu().numpy(); nb = int(r["n_bwd"][0])
        for i in range(nb):  # leave the solver object in the state the reference would be in
            if tr[i, 1] >= 0:
                self._decrease_regularization()
        if verbose:
            for i in range(nb):
                if tr[i, 1] >= 0:
                    print(f"{i + 1}/{n_lqr_iter}\tJ: {tr[i, 3]:g}")
        self.status = int(r["status"][0]); self.n_bwd = nb; self.n_fwd = int(r["n_fwd"][0])
        if verbose and self.status == _lib.STATUS_KILLED:
            print(f"Killing due to exceeded computation time > {t_kill} s.")
        if self.status == _lib.STATUS_SINGULAR:
            raise np.linalg.LinAlgError("Singular matrix")     # what np.linalg.solve raises in the reference
        return r["X"][0].cpu().numpy(), r["U"][0].cpu().numpy(), float(r["J"][0])

    def _solve_host_loop(self, x0, U, n_lqr_iter, tol, t_kill, verbose):
        """The iteration logic of control.py:164-225, with passes on the GPU wherever the plugins allow."""
        alphas = np.array(_lib.alphas())
        X, J_star = self._rollout(x0, U)
        J = J_star
        if verbose:
            print(f"0/{n_lqr_iter}\tJ: {J_star:g}")
        t0 = perf_counter()
        for i in range(n_lqr_iter):
            K, d = self._backward_pass(X, U)
            accepted = converged = False
            for α in alphas:
                Xn, Un, J = self._forward_pass(X, U, K, d, α)
                if J < J_star:
                    converged = abs((J_star - J) / J_star) < tol
                    X, U, J_star = Xn, Un, J
                    self._decrease_regularization()
                    accepted = True
                    break
            if not accepted or converged:
                break
            if t_kill and perf_counter() - t0 > t_kill:
                break
            if verbose:
                print(f"{i + 1}/{n_lqr_iter}\tJ: {J_star:g}\tμ: {self.μ:g}\tΔ: {self.Δ:g}")
        return X, U, J

    def __repr__(self):
        return f"iLQR(\n\tdynamics: {self.dynamics},\n\tcost: {self.cost},\n\tN: {self.N},\n\tdt: {self.dt},\n\tμ: {self.μ},\n\tΔ: {self.Δ}\n)"
