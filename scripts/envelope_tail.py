#!/usr/bin/env python
"""How close to the envelope's bounds the implementation sits on cfg4's Monte-Carlo audit (tests/test_gpu_configs.py, 1024 seeds
of 10 Quadcopter6D, ~8000 distinct sub-problem solves): per cluster size the worst cost error in units of the ensemble's spread
(oracle/parity.py rule (1): must stay <= C = 10) and the worst state error over its bound.  Used to compare builds of the library
(DPILQR_LIB=...): a change of the kernels' rounding should move these tails no more than a different seed would.
    python scripts/envelope_tail.py [n_scenarios=1024]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import dpilqr_amd as dp  # noqa: E402
from dpilqr_amd.dispatch import solve_scenarios_distributed  # noqa: E402
from dpilqr_amd.util import random_setup  # noqa: E402
from oracle import oracle as orc, parity  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
k, T, ns, nc = 10, 75, 6, 3
x0 = np.zeros((S, ns * k)); xf = np.zeros((S, ns * k))
for s in range(S):
    np.random.seed(s)
    a, b = random_setup(k, ns, is_rotation=False, rel_dist=k, var=k / 2, n_d=3, random=True, energy=10.0)
    x0[s], xf[s] = a.ravel(), b.ravel()
Q, R, Qf = 50.0 * np.eye(ns), np.eye(nc), 1000.0 * np.eye(ns)
ids = [100 + i for i in range(k)]
dyn = dp.MultiDynamicalModel([dp.QuadcopterDynamics6D(0.1, id_) for id_ in ids])
refs = [dp.ReferenceCost(xf[0][i * ns:(i + 1) * ns], Q.copy(), R.copy(), Qf.copy(), id_) for i, id_ in enumerate(ids)]
prob = dp.ilqrProblem(dyn, dp.GameCost(refs, dp.ProximityCost([ns] * k, 0.5, [3] * k)))
U0 = np.zeros((S, T, nc * k)); U0[:, :, 0::3] = 9.80665
Xd, Ud, J, info = solve_scenarios_distributed(prob, x0[:, None, :], U0, 0.5, xf=xf, audit=True)
worst = []
for kc, a in sorted(info["audit"].items()):
    proto = orc.Problem([4] * kc, [3] * kc, a["xf"][0], Q, R, Qf, 0.5, 0.1, T)
    rep = parity.envelope(a, proto, a["x0"], a["xf"], a["U0"])
    sm = rep["summary"]
    top = np.sort(rep["cost_ratio"])[::-1][:3]
    print(f"k={kc:2d}: {sm['items']:5d} solves, violations {sm['violations']}, cost error / spread: top {np.array2string(top, precision=2)}, "
          f"state error / bound max {sm['max_err_over_bound']:.3f}, flipped {100 * sm['flipped_frac']:.1f} %", flush=True)
    worst.append(top[0])
print(f"worst cost error / spread over all sizes: {max(worst):.2f} (C = {parity.C_ENV:g})")
