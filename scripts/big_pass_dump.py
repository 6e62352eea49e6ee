"""One config-5 backward pass (4 items of 14 Quadcopter12D + 6 padded humans, T = 150, from an iterate two iLQR iterations in) -> K, d
as .npy: python scripts/big_pass_dump.py out_prefix [f32].  For A/B builds of the large-cluster sweep (DPILQR_LIB): are the gains
the same bit for bit?"""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import dpilqr_amd as dp
from dpilqr_amd.util import random_setup
k, T, B = 20, 150, 4
models = [7] * 14 + [8] * 6; n_dims = [3] * 14 + [2] * 6
Q = np.stack([np.eye(12)] * 14 + [np.diag([1.0, 1, 1, 0, 0, 0] + [0.0] * 6)] * 6)
R = np.stack([np.eye(4)] * 14 + [np.diag([1.0, 1, 1e-9, 1e-9])] * 6)
Qf = np.stack([1000.0 * np.eye(12)] * k)
x0 = np.zeros((B, 240)); xf = np.zeros((B, 240))
for s in range(B):
    np.random.seed(100 + s)
    a, b = random_setup(k, 12, is_rotation=False, rel_dist=k, var=k / 2, n_d=3, random=True, energy=100.0)
    x0[s], xf[s] = a.ravel(), b.ravel()
U0 = np.zeros((B, T, 80)); U0[:, :, [4 * i + 3 for i in range(14)]] = 9.80665 * 63.0 / 2000.0
dtype = torch.float32 if "f32" in sys.argv[2:] else torch.float64
pb = dp.ProblemBatch(models, n_dims, xf, Q, R, Qf, 0.5, 0.1, T)
X, J = pb.rollout(x0, U0, dtype=dtype)
Ud = torch.as_tensor(U0, dtype=dtype, device="cuda")
mu = torch.full((B,), 0.25, dtype=torch.float64, device="cuda")
K, d = pb.backward_pass(X, Ud, mu, dtype=dtype)
np.save(sys.argv[1] + "_K.npy", K.cpu().numpy()); np.save(sys.argv[1] + "_d.npy", d.cpu().numpy())
print("dumped", sys.argv[1], float(K.abs().max()), bool(torch.isfinite(K).all()))
