import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import dpilqr_amd as dp
from oracle import oracle as orc
from tests.test_host_logic import problem_from
from tests.golden_util import relerr
z = dict(np.load("tests/golden/g5_dispatch.npz")); tag = "uni5"
prob = problem_from(z, tag + "_")
ids = prob.ids
graph = dp.define_inter_graph_threshold(z[tag + "_x0"].reshape(1, -1), 0.5, prob.game_cost.x_dims, ids)
print(graph)
subs = prob.split(graph)
T = int(z[tag + "_T"])
for i, sub in enumerate(subs):
    x0 = z[tag + "_x0"][i * 4:(i + 1) * 4]; U0 = z[tag + "_U0"][:, i * 2:(i + 1) * 2]
    s = dp.ilqrSolver(sub, T)
    X, U, J = s.solve(x0, U0, verbose=False)
    p = orc.Problem([3], [2], z[tag + "_xf"][i * 4:(i + 1) * 4], z[tag + "_Q"][i], z[tag + "_R"][i], z[tag + "_Qf"][i], 0.5, 0.1, T)
    o = p.solve(x0, U0)
    print(i, "gpu n_bwd", s.n_bwd, "oracle", o["n_bwd"], "relX", relerr(X, o["X"]), "vs golden", relerr(X, z[tag + "_X_dec"][:, i * 4:(i + 1) * 4]), "J", J, o["J"])
    # per-pass check at U0
    Xr, Jr = s._rollout(x0, U0); Xo, Jo = p.rollout(x0, U0)
    s.μ = 1.0
    K, d = s._backward_pass(Xr, U0); Ko, do = p.backward_pass(Xo, U0, 1.0)
    print("    rollout", relerr(Xr, Xo), "K", relerr(K, Ko), "d", relerr(d, do))

print("---- batched")
from dpilqr_amd.dispatch import solve_problem_list
x0s = [z[tag + "_x0"][i * 4:(i + 1) * 4] for i in range(5)]; U0s = [z[tag + "_U0"][:, i * 2:(i + 1) * 2] for i in range(5)]
res = solve_problem_list(subs, x0s, U0s)
for i in range(5):
    print(i, res[i][3], relerr(res[i][0], z[tag + "_X_dec"][:, i * 4:(i + 1) * 4]))
from dpilqr_amd.lowering import lower_problems
pb = lower_problems(subs, T)
print(pb.desc.B, pb.desc.k, pb.desc.model_bstride, pb.desc.xf_bstride, pb.desc.Q_bstride, pb.desc.radius_bstride, pb._xf.shape, pb._model)
X, J = pb.rollout(np.stack(x0s), np.stack(U0s))
for i in range(5):
    print("rollout", i, relerr(X[i].cpu().numpy(), dp.ilqrSolver(subs[i], T)._rollout(x0s[i], U0s[i])[0]))
