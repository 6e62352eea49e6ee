#!/usr/bin/env python
"""CPU study behind the pivot threshold of the blocked S3 (csrc/riccati_wg.hpp, gj_blocked): on the Q_uu of real iterates
(15 unicycles T = 100, 10 quadcopters T = 75, after two iLQR iterations, mu = 0.125; oracle/numpy_port.py) how often Gauss-Jordan
with a diagonal-preferring threshold declines, and how far its solution is from LAPACK's (numpy.linalg.solve), per threshold.
    python scripts/threshold_study.py   (a few minutes, no GPU; output: profiles/r03_threshold_study.txt)"""
import sys, numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import numpy_port as npp
from dpilqr_amd.util import random_setup

def gj_threshold(A, B, tau):
    """Gauss-Jordan with diagonal-preferring threshold pivoting; returns (declined, X)"""
    A = A.copy(); B = B.copy(); m = A.shape[0]
    for k in range(m):
        pv = A[k, k]
        below = np.abs(A[k+1:, k]).max() if k + 1 < m else 0.0
        if not (abs(pv) > 0) or below > tau * abs(pv):
            return True, None
        inv = 1.0 / pv
        for r in range(m):
            if r == k: continue
            l = A[r, k] * inv
            A[r] -= l * A[k]; B[r] -= l * B[k]
        A[k] *= inv; B[k] *= inv
    return False, B

def collect(k, mdl, seeds, iters=2):
    if mdl == 'uni4':
        models, ns, nc, nd, T = [3] * k, 4, 2, 2, 100
        Q = np.diag([1.0, 1, 0, 0])
    else:
        models, ns, nc, nd, T = [4] * k, 6, 3, 3, 75
        Q = 50.0 * np.eye(6)
    out = []
    for s in seeds:
        np.random.seed(500 + s)
        a, b = random_setup(k, ns, is_rotation=False, rel_dist=k, var=k / 2, n_d=nd, random=True, energy=10.0)
        sol = npp.cfg_solver(models, [nd] * k, b.ravel(), Q, np.eye(nc), 1000.0 * np.eye(ns), 0.5, 0.1, T)
        U0 = np.zeros((T, k * nc))
        if mdl != 'uni4': U0[:, 0::3] = 9.80665
        r = sol.solve(a.ravel(), U0, n_lqr_iter=iters)
        X, U = r['X'], r['U']
        sol.mu = 0.125
        # capture Quu, rhs per step
        n, m = sol.n_x, sol.n_u
        reg = sol.mu * np.eye(n)
        p, _, P, _, _ = sol.cost.quadraticize(X[-1], np.zeros(m), terminal=True)
        for t in range(sol.N - 1, -1, -1):
            Lx, Lu, Lxx, Luu, Lux = sol.cost.quadraticize(X[t], U[t])
            A, B = sol.dynamics.linearize(X[t], U[t])
            Qx = Lx + A.T @ p; Qu = Lu + B.T @ p
            Qxx = Lxx + A.T @ P @ A; Quu = Luu + B.T @ (P + reg) @ B; Qux = Lux + B.T @ (P + reg) @ A
            out.append((Quu.copy(), np.hstack([Qux, Qu[:, None]])))
            K = -np.linalg.solve(Quu, Qux); d = -np.linalg.solve(Quu, Qu)
            p = Qx + K.T @ Quu @ d + K.T @ Qu + Qux.T @ d
            P = Qxx + K.T @ Quu @ K + K.T @ Qux + Qux.T @ K
            P = 0.5 * (P + P.T)
    return out

for mdl, k in (('uni4', 15), ('quad6', 10)):
    data = collect(k, mdl, range(6))
    print(mdl, k, len(data), 'steps')
    for tau in (8, 16, 32, 64, 256, 1024, 1e6):
        dec = 0; errs = []
        for Quu, R in data:
            bad, Xs = gj_threshold(Quu, R, tau)
            if bad: dec += 1; continue
            ref = np.linalg.solve(Quu, R)
            errs.append(np.abs(Xs - ref).max() / max(np.abs(ref).max(), 1e-300))
        errs = np.array(errs)
        print(f"  tau {tau:>8}: declined {100 * dec / len(data):5.1f} %   rel err vs LAPACK: median {np.median(errs):.1e} p99 {np.quantile(errs, 0.99):.1e} max {errs.max():.1e}")
