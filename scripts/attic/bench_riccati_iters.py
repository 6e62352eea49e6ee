"""Is the sweep's duration data dependent?  Time it on the tiles / mu of iLQR iteration k of 2048 cfg2 items."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import dpilqr_amd as dp
from dpilqr_amd import _lib
from dpilqr_amd.device import empty, ptr, stream_handle, to_dev
from bench import scenarios, K_AGENTS, T, N_U, N_X
B = 2048
x0, xf = scenarios(0, B)
pb = dp.ProblemBatch([0] * K_AGENTS, [2] * K_AGENTS, xf, np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4), 0.5, 0.1, T)
K = empty((B, T, N_U, N_X)); d = empty((B, T, N_U)); lib = _lib.load()
for it in (0, 1, 2, 3, 5, 8, 12):
    r = pb.solve(x0, np.zeros((B, T, N_U)), n_lqr_iter=max(it, 1), trace=True)
    if it == 0:
        X, _ = pb.rollout(x0, np.zeros((B, T, N_U))); U = torch.zeros((B, T, N_U), dtype=torch.float64, device="cuda"); mu = to_dev(np.ones(B))
    else:
        X, U = r["X"], r["U"]
        tr = r["trace"].cpu().numpy()          # (B, n_iter, 5): mu_before, alpha idx, J_last, J_star, n_fwd
        mu_h = tr[:, it - 1, 0].copy()
        acc = tr[:, it - 1, 1] >= 0
        mu_next = np.where(acc, np.maximum(mu_h * 0.5, 0.0), mu_h)   # rough: what the next pass would use
        mu_next[mu_next <= 1e-6] = 0.0
        mu = to_dev(mu_next)
    tiles = pb.make_tiles(X, U)
    def run(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            _lib.check(lib.dpilqr_backward_pass_tiles_blocks(B, T, N_X, N_U, 4, 2, ptr(tiles), ptr(mu), ptr(K), ptr(d), None, None, None, stream_handle()))
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    run(3)
    print(f"iteration {it:2d}: mean mu {float(mu.mean()):.4f}  sweep {run(100):.1f} us/launch")
