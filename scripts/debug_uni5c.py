import os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import dpilqr_amd as dp
from dpilqr_amd.lowering import lower_problems
from tests.test_host_logic import problem_from
from tests.golden_util import relerr
from oracle import oracle as orc
z = dict(np.load("tests/golden/g5_dispatch.npz")); tag = "uni5"
prob = problem_from(z, tag + "_")
subs = prob.split({i: [i] for i in prob.ids}); T = int(z[tag + "_T"])
x0s = np.stack([z[tag + "_x0"][i * 4:(i + 1) * 4] for i in range(5)]); U0s = np.stack([z[tag + "_U0"][:, i * 2:(i + 1) * 2] for i in range(5)])
pb = lower_problems(subs, T)
X, J = pb.rollout(x0s, U0s)
tiles = pb.make_tiles(X, U0s)
mu = np.array([1.0, 0.5, 0.125, 1.0, 0.5])
K, d = dp.backward_pass_tiles(tiles, 5, T, 4, 2, dp.device.to_dev(mu))
for i in range(5):
    p = orc.Problem([3], [2], z[tag + "_xf"][i * 4:(i + 1) * 4], z[tag + "_Q"][i], z[tag + "_R"][i], z[tag + "_Qf"][i], 0.5, 0.1, T)
    Ko, do = p.backward_pass(X[i].cpu().numpy(), U0s[i], mu[i])
    print(i, "K", relerr(K[i].cpu().numpy(), Ko), "d", relerr(d[i].cpu().numpy(), do))
