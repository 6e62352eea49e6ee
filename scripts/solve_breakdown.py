#!/usr/bin/env python
"""Where a windowed device solve of one cluster size spends its GPU time: B clusters of k agents solved to convergence with the
library's own profiler on (HIP events around every launch, by kernel class).  One bucket, one stream: nothing overlaps.
    python scripts/solve_breakdown.py [--model quad6|uni4|di4] [--B 2048] [k ...]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import dpilqr_amd as dp  # noqa: E402
from dpilqr_amd import _lib  # noqa: E402
from dpilqr_amd.util import random_setup  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="quad6")
ap.add_argument("--B", type=int, default=2048)
ap.add_argument("ks", nargs="*", type=int)
args = ap.parse_args()
mdl, ns, nc, nd, T = {"quad6": (4, 6, 3, 3, 75), "uni4": (3, 4, 2, 2, 100), "di4": (0, 4, 2, 2, 50)}[args.model]
ks = args.ks or ([4, 6, 8, 10] if mdl == 4 else [6, 9, 12, 15])
for k in ks:
    B, n, m = args.B, k * ns, k * nc
    x0 = np.zeros((B, n)); xf = np.zeros((B, n))
    for s in range(B):
        np.random.seed(500 + s)
        a, b = random_setup(k, ns, is_rotation=False, rel_dist=k, var=k / 2, n_d=nd, random=True, energy=10.0)
        x0[s], xf[s] = a.ravel(), b.ravel()
    Q = (50.0 * np.eye(6)) if mdl == 4 else np.diag([1.0, 1, 0, 0])
    U0 = np.zeros((B, T, m))
    if mdl == 4:
        U0[:, :, 0::3] = 9.80665
    pb = dp.ProblemBatch([mdl] * k, [nd] * k, xf, Q, np.eye(nc), 1000.0 * np.eye(ns), 0.5, 0.1, T)
    pb.solve(x0, U0, n_lqr_iter=2)                      # warm
    torch.cuda.synchronize()
    t0 = time.perf_counter(); r = pb.solve(x0, U0); torch.cuda.synchronize(); wall = time.perf_counter() - t0
    _lib.profile_enable(True)
    r = pb.solve(x0, U0); torch.cuda.synchronize()
    pr = _lib.profile_read()
    _lib.profile_enable(False)
    tot = sum(v["ms"] for v in pr.values())
    it = float(r["n_bwd"].double().mean())
    print(f"{args.model} k={k:2d} n_x={n:2d} B={B}: wall {wall * 1e3:7.1f} ms ({B / wall:8.0f} sub-problems/s), mean iterations {it:4.1f} | "
          + ", ".join(f"{c} {v['ms']:6.1f} ms ({100 * v['ms'] / tot:4.1f} %, {v['launches']} launches)" for c, v in pr.items()), flush=True)
