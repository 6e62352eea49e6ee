#!/usr/bin/env python
"""Small clusters of twelve-state agents (Quadcopter12D, cfg5's model, T = 150): what the distributed solve of config 5 dispatches.
B clusters of k agents, a few solver iterations with the library's profiler on: time per kernel class.
    python scripts/bench_q12.py [--B 512] [--iters 4] [k ...]"""
import argparse
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import dpilqr_amd as dp  # noqa: E402
from dpilqr_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=512)
ap.add_argument("--iters", type=int, default=4)
ap.add_argument("ks", nargs="*", type=int)
a = ap.parse_args()
ns, nc, nd, T, model = 12, 4, 3, 150, 7
for k in (a.ks or [1, 2, 3, 5]):
    B = a.B
    rng = np.random.default_rng(4000 + k)
    xf = rng.normal(size=(B, k * ns)) * 2.0; x0 = rng.normal(size=(B, k * ns)) * 2.0
    x0.reshape(B, k, ns)[:, :, nd:] *= 0.1; xf.reshape(B, k, ns)[:, :, nd:] = 0.0
    x0.reshape(B, k, ns)[:, :, 3:] *= 0.02
    U0 = rng.normal(size=(B, T, k * nc)) * 0.05 * 1e-4; U0[:, :, 3::4] += 9.80665 * 63.0 / 2000.0
    pb = dp.ProblemBatch([model] * k, [nd] * k, xf, np.eye(ns) * 1.3, np.eye(nc), 100.0 * np.eye(ns), 0.7, 0.1, T)
    pb.solve(x0, U0, n_lqr_iter=a.iters)
    _lib.profile_enable(True); _lib.profile_read(reset=True)
    pb.solve(x0, U0, n_lqr_iter=a.iters); torch.cuda.synchronize()
    prof = _lib.profile_read(reset=True); _lib.profile_enable(False)
    print(f"quad12 k={k} n_x={k * ns} T={T} B={B}, {a.iters} iterations: " +
          ", ".join(f"{c} {v['ms']:.2f} ms / {v['launches']}" for c, v in prof.items() if v["launches"]), flush=True)
