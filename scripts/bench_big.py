#!/usr/bin/env python
"""Time the large-cluster passes (tu_big.hip) at BASELINE config 5's size: B items of 20 twelve-state agents
(14 Quadcopter12D + 6 padded humans), n_x = 240, n_u = 80, T = 150, fp64 and fp32.
    python scripts/bench_big.py [B ...]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import dpilqr_amd as dp  # noqa: E402
from dpilqr_amd.util import random_setup  # noqa: E402

k, T = 20, 150
models = [7] * 14 + [8] * 6
n_dims = [3] * 14 + [2] * 6
Q = np.stack([np.eye(12)] * 14 + [np.diag([1.0, 1, 1, 0, 0, 0] + [0.0] * 6)] * 6)
R = np.stack([np.eye(4)] * 14 + [np.diag([1.0, 1, 1e-9, 1e-9])] * 6)
Qf = np.stack([1000.0 * np.eye(12)] * k)
n, m = 12 * k, 4 * k
# dense fp64 flops of one backward pass with the block structure of A, B exploited
flops_pass = T * 2.0 * (k * k * 16 * (144 + 16 * 12) + m ** 3 / 3 + (n + 1) * m * m + m * m * n + 3 * n * (n + 1) * m)
# SURVEY 8(d)'s dense count (what bench.py uses for cfg2): T (4 n^3 + 8 n^2 m + 6 n m^2 + 2/3 m^3) = 15.3 Gflop
flops_dense = T * (4.0 * n ** 3 + 8.0 * n * n * m + 6.0 * n * m * m + 2.0 * m ** 3 / 3.0)


def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for B in [int(a) for a in sys.argv[1:]] or [1, 32, 256]:
    x0 = np.zeros((B, n)); xf = np.zeros((B, n))
    for s in range(B):
        np.random.seed(100 + s)
        a, b = random_setup(k, 12, is_rotation=False, rel_dist=k, var=k / 2, n_d=3, random=True, energy=100.0)
        x0[s], xf[s] = a.ravel(), b.ravel()
    U0 = np.zeros((B, T, m)); U0[:, :, [4 * i + 3 for i in range(14)]] = 9.80665 * 63.0 / 2000.0
    pb = dp.ProblemBatch(models, n_dims, xf, Q, R, Qf, 0.5, 0.1, T)
    for dtype in (torch.float64, torch.float32):
        X, J = pb.rollout(x0, U0, dtype=dtype)
        Ud = torch.as_tensor(U0, dtype=dtype, device="cuda")
        mu = torch.ones(B, dtype=torch.float64, device="cuda")
        K, d = pb.backward_pass(X, Ud, mu, dtype=dtype)
        t_b = timeit(lambda: pb.backward_pass(X, Ud, mu, dtype=dtype))
        al = np.array(dp._lib.alphas())
        t_f = timeit(lambda: pb.forward_pass(X, Ud, K, d, al, dtype=dtype))
        t_r = timeit(lambda: pb.rollout(x0, U0, dtype=dtype))
        name = "fp64" if dtype == torch.float64 else "fp32"
        print(f"B={B:4d} {name}: backward {t_b * 1e3:8.2f} ms ({B * flops_pass / t_b / 1e12:6.2f} TFLOP/s of the flops left with the block structure, "
              f"{B * flops_dense / t_b / 1e12:6.2f} by SURVEY 8(d)'s dense count), "
              f"10 forward passes {t_f * 1e3:8.2f} ms, rollout {t_r * 1e3:7.2f} ms", flush=True)
