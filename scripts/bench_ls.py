#!/usr/bin/env python
"""Launch time of ONE full-window line search (k_linesearch_wave<0,5>) and ONE fused sweep on cfg2 items: a solve with
n_lqr_iter = 1 and the whole batch in flight, the library's profiler on every kernel class.
    python scripts/bench_ls.py [--B 6144] [--reps 7]"""
import argparse, sys, statistics
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import dpilqr_amd as dp  # noqa: E402
from dpilqr_amd import _lib  # noqa: E402
from dpilqr_amd.util import random_setup_batch  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--B", type=int, default=6144); ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--iters", type=int, default=1)
a = ap.parse_args()
B = a.B
x0, xf = random_setup_batch((0, B), 5, 4, var=2.5, n_d=2, energy=10.0)
pb = dp.ProblemBatch([0] * 5, [2] * 5, xf, np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4), 0.5, 0.1, 50)
U0 = torch.zeros((B, 50, 10), dtype=torch.float64, device="cuda")
pb.solve(x0, U0, n_lqr_iter=a.iters, window=B)
_lib.profile_enable(True)
rows = {}
for _ in range(a.reps):
    _lib.profile_read(reset=True)
    pb.solve(x0, U0, n_lqr_iter=a.iters, window=B); torch.cuda.synchronize()
    for c, v in _lib.profile_read(reset=True).items():
        if v["launches"]:
            rows.setdefault(c, []).append(v["ms"] / v["launches"])
_lib.profile_enable(False)
print(f"cfg2, {B} items in one window, n_lqr_iter = {a.iters}: median launch time over {a.reps} solves  " +
      "  ".join(f"{c} {statistics.median(v) * 1e3:7.1f} us" for c, v in rows.items()), flush=True)
