#!/usr/bin/env python
"""Launch time of the line search (k_linesearch_wave<model, k>) per cluster size, alone on the chip: B clusters of k agents, three
iLQR iterations with the whole batch in one window, the library's profiler on the line-search class.
    python scripts/bench_ls_sizes.py [--B 2048] uni4:15 uni4:14 quad6:8 quad6:6 ..."""
import argparse
import statistics
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import dpilqr_amd as dp  # noqa: E402
from dpilqr_amd import _lib  # noqa: E402
from dpilqr_amd.util import random_setup_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=2048)
ap.add_argument("cases", nargs="*", default=["uni4:15", "uni4:14", "quad6:8", "quad6:7", "quad6:6"])
a = ap.parse_args()
FAM = {"uni4": (3, 4, 2, 2, 100), "quad6": (4, 6, 3, 3, 75), "di4": (0, 4, 2, 2, 50)}
for case in a.cases:
    fam, k = case.split(":"); k = int(k)
    mdl, ns, nc, nd, T = FAM[fam]
    B = a.B
    x0, xf = random_setup_batch((500, B), k, ns, var=k / 2, n_d=nd, energy=10.0)
    Q = (50.0 * np.eye(6)) if ns == 6 else np.diag([1.0, 1, 0, 0])
    pb = dp.ProblemBatch([mdl] * k, [nd] * k, xf, Q, np.eye(nc), 1000.0 * np.eye(ns), 0.5, 0.1, T)
    U0 = torch.zeros((B, T, k * nc), dtype=torch.float64, device="cuda")
    if mdl == 4:
        U0[:, :, 0::3] = 9.80665
    pb.solve(x0, U0, n_lqr_iter=3, window=B)
    _lib.profile_enable(True)
    ms, sw = [], []
    for _ in range(5):
        _lib.profile_read(reset=True)
        pb.solve(x0, U0, n_lqr_iter=3, window=B); torch.cuda.synchronize()
        p = _lib.profile_read(reset=True)
        ls = p["forward"]
        ms.append(ls["ms"] / max(ls["launches"], 1))
        sw.append(p["riccati"]["ms"] / max(p["riccati"]["launches"], 1))
    _lib.profile_enable(False)
    print(f"{fam} k={k:2d} B={B}: line search {statistics.median(ms):7.3f} ms per launch, sweep {statistics.median(sw):7.3f} ms per launch "
          f"(median of 5 three-iteration solves; classes {sorted(p)})", flush=True)
