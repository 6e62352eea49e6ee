"""Per-shape timings of the three kernels (which sweep / line-search instantiation each shape gets, and how far
the size-generic fallbacks are behind)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import dpilqr_amd as dp
from dpilqr_amd import _lib
from dpilqr_amd.device import to_dev
from dpilqr_amd.util import random_setup

def run(model, k, T, B, energy=10.0):
    ns, nc = {0: (4, 2), 3: (4, 2), 4: (6, 3)}[model]
    nd = 3 if ns == 6 else 2
    x0 = np.zeros((B, k * ns)); xf = np.zeros((B, k * ns))
    for i in range(B):
        np.random.seed(i)
        a, b = random_setup(k, ns, is_rotation=False, rel_dist=k, var=k / 2, n_d=nd, random=True, energy=energy)
        x0[i], xf[i] = a.ravel(), b.ravel()
    if ns == 4:
        Q, R = np.diag([1.0, 1, 0, 0]), np.eye(2)
    else:
        Q, R = 50.0 * np.eye(6), np.eye(3)
    pb = dp.ProblemBatch([model] * k, [nd] * k, xf, Q, R, 1000.0 * np.eye(ns), 0.5, 0.1, T)
    U0 = np.zeros((B, T, k * nc))
    if model == 4:
        U0[:, :, 0::3] = 9.80665
    X, _ = pb.rollout(x0, U0)
    Ud = to_dev(U0); mu = to_dev(np.ones(B))
    def timed(f, n=5):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    tiles = pb.make_tiles(X, Ud)
    t_tiles = timed(lambda: pb.make_tiles(X, Ud, tiles))
    t_sweep = timed(lambda: dp.backward_pass_tiles(tiles, B, T, pb.n_x, pb.n_u, mu, blocks=(ns, nc)))
    pb.solve(x0, U0)   # warm
    _lib.profile_enable(True); _lib.profile_read(reset=True)
    t0 = time.perf_counter(); r = pb.solve(x0, U0); torch.cuda.synchronize(); t_solve = (time.perf_counter() - t0) * 1e3
    prof = _lib.profile_read(reset=True); _lib.profile_enable(False)
    nb = float(r["n_bwd"].float().mean())
    print(f"model {model} k={k:2d} n_x={k*ns:2d} T={T:3d} B={B:5d}: tiles {t_tiles:7.2f} ms  sweep {t_sweep:8.2f} ms  "
          f"solve {t_solve:9.1f} ms ({nb:.1f} passes/item, {B / t_solve * 1e3:9.0f} sub/s)  in-solve ms: "
          + ", ".join(f"{k} {v['ms']:.1f}/{v['launches']}" for k, v in prof.items()), flush=True)

cases = [(0, 5, 50, 2048), (0, 6, 50, 1024), (0, 8, 50, 1024), (3, 5, 100, 1024), (3, 8, 100, 512), (3, 12, 100, 256), (3, 15, 100, 256),
         (4, 2, 75, 1024), (4, 4, 75, 1024), (4, 7, 75, 512), (4, 10, 75, 256)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for c in cases:
    run(*c)
