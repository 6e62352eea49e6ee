#!/usr/bin/env python
"""Time the mid-size Riccati sweep (k_riccati_wg, n_x 24..60: cfg3 / cfg4 clusters) and its tile producer per cluster size.
For each k: B clusters of k Quadcopter6D (T = 75, cfg4's model) or k Unicycle4D (T = 100, cfg3's) at the iterate reached after
two iLQR iterations; prints the launch time of make_tiles and of the sweep, the sweep's time per item-step, its dense-count
fp64 rate and the record bytes per second it consumes.
    python scripts/bench_wg.py [--model quad6|uni4] [--B 2048] [k ...]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import dpilqr_amd as dp  # noqa: E402
from dpilqr_amd.batch import backward_pass_tiles  # noqa: E402
from dpilqr_amd.util import random_setup  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="quad6")
ap.add_argument("--B", type=int, default=2048)
ap.add_argument("ks", nargs="*", type=int)
args = ap.parse_args()
mdl, ns, nc, nd, T = {"quad6": (4, 6, 3, 3, 75), "uni4": (3, 4, 2, 2, 100), "car3": (2, 3, 2, 2, 50), "quad12": (7, 12, 4, 3, 150),
                       "di6": (1, 6, 3, 3, 75)}[args.model]
ks = args.ks or ([4, 5, 6, 7, 8, 9, 10] if mdl == 4 else [7, 9, 11, 13, 15])


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for k in ks:
    B, n, m = args.B, k * ns, k * nc
    x0 = np.zeros((B, n)); xf = np.zeros((B, n))
    for s in range(B):
        np.random.seed(500 + s)
        a, b = random_setup(k, ns, is_rotation=False, rel_dist=k, var=k / 2, n_d=nd, random=True, energy=10.0)
        x0[s], xf[s] = a.ravel(), b.ravel()
    Q = (50.0 * np.eye(6)) if ns == 6 else (np.diag([1.0, 1, 0, 0]) if ns == 4 else np.eye(ns))
    U0 = np.zeros((B, T, m))
    if mdl == 4:
        U0[:, :, 0::3] = 9.80665
    if mdl == 7:
        U0[:, :, 3::4] = 9.80665 * 63 / 2000
    pb = dp.ProblemBatch([mdl] * k, [nd] * k, xf, Q, np.eye(nc), 1000.0 * np.eye(ns), 0.5, 0.1, T)
    r = pb.solve(x0, U0, n_lqr_iter=2)
    X, U = r["X"], r["U"]
    mu = torch.full((B,), 0.125, dtype=torch.float64, device="cuda")
    tiles = pb.make_tiles(X, U)
    t_t = timeit(lambda: pb.make_tiles(X, U, tiles))
    t_s = timeit(lambda: backward_pass_tiles(tiles, B, T, n, m, mu, blocks=(ns, nc)))
    try:       # the record-free form the solve loop uses (k_riccati_wg<..., FUSED = true>)
        t_f = timeit(lambda: pb.backward_pass_fused(X, U, mu))
    except Exception:
        t_f = float("nan")
    flops = T * (4.0 * n ** 3 + 8.0 * n * n * m + 6.0 * n * m * m + 2.0 * m ** 3 / 3.0)
    rec = 8.0 * (T + 1) * (2 * n * n + 2 * n * m + m * m + n + m)
    print(f"{args.model} k={k:2d} n_x={n:2d} n_u={m:2d} B={B}: make_tiles {t_t * 1e3:7.2f} ms | fused sweep {t_f * 1e3:7.2f} ms | record-fed sweep {t_s * 1e3:7.2f} ms = "
          f"{t_s / B / T * 1e9 * min(B, 256):7.0f} ns per item-step per CU slot, {B * flops / t_s / 1e12:5.2f} TFLOP/s dense count, "
          f"records {B * rec / t_s / 1e9:6.0f} GB/s", flush=True)
