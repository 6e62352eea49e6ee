"""cfg3 / cfg4 of BASELINE.json as Monte-Carlo runs through the many-scenario front end
(dispatch.solve_scenarios_distributed): S random-goal scenarios of one k-agent problem, proximity-graph split,
one windowed device solve per cluster size, stitched trajectories.

    python scripts/montecarlo.py cfg3 [S]     15 x Unicycle4D, T=100   (scripts/analysis.py distribution)
    python scripts/montecarlo.py cfg4 [S]     10 x Quadcopter6D, T=75, hover warm start
"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import dpilqr_amd as dp
from dpilqr_amd.dispatch import solve_scenarios_distributed
from dpilqr_amd.util import random_setup

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
if cfg == "cfg3":
    k, ns, nc, nd, T, Model = 15, 4, 2, 2, 100, dp.UnicycleDynamics4D
    Q, R = np.diag([1.0, 1, 0, 0]), np.eye(2)
else:
    k, ns, nc, nd, T, Model = 10, 6, 3, 3, 75, dp.QuadcopterDynamics6D
    Q, R = 50.0 * np.eye(6), np.eye(3)
Qf, dt, radius = 1000.0 * np.eye(ns), 0.1, 0.5
x0 = np.zeros((S, k * ns)); xf = np.zeros((S, k * ns))
for s in range(S):
    np.random.seed(s)
    a, b = random_setup(k, ns, is_rotation=False, rel_dist=k, var=k / 2, n_d=nd, random=True, energy=10.0)
    x0[s], xf[s] = a.ravel(), b.ravel()
ids = [f"a{i}" for i in range(k)]
dyn = dp.MultiDynamicalModel([Model(dt, id_) for id_ in ids])
costs = [dp.ReferenceCost(xf[0, i * ns:(i + 1) * ns], Q, R, Qf, id_) for i, id_ in enumerate(ids)]
prob = dp.ilqrProblem(dyn, dp.GameCost(costs, dp.ProximityCost([ns] * k, radius, [nd] * k)))
U0 = np.zeros((S, T, k * nc))
if cfg == "cfg4":
    U0[:, :, 0::3] = 9.80665
solve_scenarios_distributed(prob, x0[:2, None, :], U0[:2], radius, xf=xf[:2])          # warm (kernels, allocator)
torch.cuda.synchronize()
t0 = time.perf_counter()
Xd, Ud, J, info = solve_scenarios_distributed(prob, x0[:, None, :], U0, radius, xf=xf)
torch.cuda.synchronize()
dt1 = time.perf_counter() - t0
sec = info["seconds"]
host_frac = 1.0 - sec["solves"] / dt1
print(f"      stages: front end (uploads, graph, de-duplication, buckets) {sec['front_end']:.3f} s, bucket solves {sec['solves']:.3f} s, "
      f"stitch + J_full rollout {sec['stitch_and_rollout']:.3f} s, device->host copies of the results {dt1 - sum(sec.values()):.3f} s; "
      f"outside the solves: {100 * host_frac:.1f} % of the call")
print(f"{cfg}: {S} scenarios x {k} agents, T={T}: first call (X = x0) {dt1:.2f} s  -> {info['n_unique']} distinct sub-problems of "
      f"{info['n_subproblems']} ({info['n_unique'] / dt1:.0f} sub-problems/s, {S / dt1:.1f} scenarios/s), sizes {info['sizes']}, "
      f"finite {np.isfinite(J).mean():.3f}, median J {np.median(J):.1f}")
t0 = time.perf_counter()
Xd2, Ud2, J2, info2 = solve_scenarios_distributed(prob, Xd, Ud, radius, xf=xf)       # the receding-horizon pattern
torch.cuda.synchronize()
dt2 = time.perf_counter() - t0
print(f"      second call (X = first result) {dt2:.2f} s -> {info2['n_unique']} distinct sub-problems ({info2['n_unique'] / dt2:.0f}/s), "
      f"sizes {info2['sizes']}, median J {np.median(J2):.1f}, stages {info2['seconds']}")
if len(sys.argv) > 3:
    import json
    Path(sys.argv[3]).write_text(json.dumps(dict(config=cfg, scenarios=S, agents=k, T=T, first_call_s=dt1, first_call_stages=sec,
                                                 first_call_outside_solves_frac=host_frac, n_unique=info["n_unique"], sizes=info["sizes"],
                                                 second_call_s=dt2, second_call_stages=info2["seconds"], second_sizes=info2["sizes"]), indent=1))
