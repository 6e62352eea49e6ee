import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import dpilqr_amd as dp
from dpilqr_amd.lowering import lower_problems
from tests.test_host_logic import problem_from
from tests.golden_util import relerr
z = dict(np.load("tests/golden/g5_dispatch.npz")); tag = "uni5"
prob = problem_from(z, tag + "_")
graph = {i: [i] for i in prob.ids}
subs = prob.split(graph); T = int(z[tag + "_T"])
x0s = np.stack([z[tag + "_x0"][i * 4:(i + 1) * 4] for i in range(5)]); U0s = np.stack([z[tag + "_U0"][:, i * 2:(i + 1) * 2] for i in range(5)])
pb = lower_problems(subs, T)
for w in (1, 2, 5):
    r = pb.solve(x0s, U0s, window=w)
    print("window", w, r["n_bwd"].tolist(), [f"{relerr(r['X'][i].cpu().numpy(), z[tag + '_X_dec'][:, i * 4:(i + 1) * 4]):.1e}" for i in range(5)])
