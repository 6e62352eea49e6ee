"""cfg1 of BASELINE.json in the style of the reference's scripts/examples.py: three DoubleIntDynamics4D agents,
T = 50, one centralised Potential-iLQR solve, then the same scenario through solve_distributed -- written against
the package exactly as a dp-ilqr user would write it against `dpilqr` (only the import differs)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import dpilqr_amd as dpilqr

n_agents, n_states, n_controls, N, dt, radius = 3, 4, 2, 50, 0.1, 0.5
x0 = np.array([[0.5, 1.5, 0, 0, 2.5, 1.5, 0, 0, 1.5, 1.3, 0, 0]]).T          # scenarios.py:12-22 style table
xf = np.array([[2.5, 1.5, 0, 0, 0.5, 1.5, 0, 0, 1.5, 2.2, 0, 0]]).T
ids = [100 + i for i in range(n_agents)]
Q, R, Qf = np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4)
dynamics = dpilqr.MultiDynamicalModel([dpilqr.DoubleIntDynamics4D(dt, id_) for id_ in ids])
goal_costs = [dpilqr.ReferenceCost(xf_i, Q.copy(), R.copy(), Qf.copy(), id_)
              for xf_i, id_ in zip(dpilqr.split_agents_gen(xf, [n_states] * n_agents), ids)]
prox_cost = dpilqr.ProximityCost([n_states] * n_agents, radius, [2] * n_agents)
problem = dpilqr.ilqrProblem(dynamics, dpilqr.GameCost(goal_costs, prox_cost))
solver = dpilqr.ilqrSolver(problem, N)
X, U, J = solver.solve(x0, np.zeros((N, n_agents * n_controls)), verbose=False)
print(f"centralised: J = {J:.6f}, final distance to goal {np.linalg.norm(X[-1] - xf.ravel()):.3e}")
Xd, Ud, Jd, info = dpilqr.solve_distributed(problem, x0.T, np.zeros((N, n_agents * n_controls)), radius, ignore_ids=[], verbose=False)
print(f"distributed: J_full = {Jd:.6f}, neighbourhoods {[info[i][1] for i in ids]}")
