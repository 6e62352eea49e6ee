import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import dpilqr_amd as dp
from oracle import oracle as orc
from tests.test_host_logic import problem_from
from tests.golden_util import relerr
z = dict(np.load("tests/golden/g5_dispatch.npz")); tag = "uni8"
prob = problem_from(z, tag + "_"); ids = prob.ids; T = int(z[tag + "_T"])
X0 = z[tag + "_x0"].reshape(1, -1); U0 = z[tag + "_U0"]
graph = dp.define_inter_graph_threshold(X0, 0.5, prob.game_cost.x_dims, ids)
print(graph)
Xd, Ud, Jf, info = dp.solve_distributed(prob, X0, U0, 0.5, ignore_ids=[], verbose=False)
for i in range(8):
    print(i, graph[ids[i]], relerr(Xd[:, i*4:(i+1)*4], z[tag + "_X_dec"][:, i*4:(i+1)*4]))
subs = prob.split(graph)
x0s = dp.split_graph(X0, prob.game_cost.x_dims, graph); Us = dp.split_graph(U0, prob.game_cost.u_dims, graph)
for i in range(8):
    s = dp.ilqrSolver(subs[i], T)
    X, U, J = s.solve(x0s[i], Us[i], verbose=False)
    Xa, Ua = subs[i].extract(X, U, ids[i])
    print("single", i, s.n_bwd, relerr(Xa, z[tag + "_X_dec"][:, i*4:(i+1)*4]))
