import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import dpilqr_amd as dp
from oracle import oracle as orc
from tests.golden_util import relerr
rng = np.random.default_rng(0)
for k in (1, 2, 3, 4, 5):
    B, T = 9, 12
    xf = rng.normal(size=(B, 4 * k)); x0 = rng.normal(size=(B, 4 * k)); U = rng.normal(size=(B, T, 2 * k)) * 0.1
    Q, R, Qf = np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4)
    pb = dp.ProblemBatch([3] * k, [2] * k, xf, Q, R, Qf, 0.5, 0.1, T)
    X, J = pb.rollout(x0, U)
    mu = rng.uniform(0, 1, size=B)
    K, d = pb.backward_pass(X, U, dp.device.to_dev(mu))
    errs = []
    for i in range(B):
        p = orc.Problem([3] * k, [2] * k, xf[i], Q, R, Qf, 0.5, 0.1, T)
        Ko, do = p.backward_pass(X[i].cpu().numpy(), U[i], mu[i])
        errs.append(max(relerr(K[i].cpu().numpy(), Ko), relerr(d[i].cpu().numpy(), do)))
    print("k", k, ["%.1e" % e for e in errs])
