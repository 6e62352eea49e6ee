"""Sustained-load check of the Riccati sweep: many back-to-back launches at a realistic operating point."""
import os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import dpilqr_amd as dp
from dpilqr_amd import _lib
from dpilqr_amd.device import empty, ptr, stream_handle, to_dev
from bench import scenarios, K_AGENTS, T, N_U, N_X, BWD_READ_BYTES, BWD_WRITE_BYTES
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
it = int(os.environ.get("REALISTIC", "0"))
x0, xf = scenarios(0, B)
pb = dp.ProblemBatch([0] * K_AGENTS, [2] * K_AGENTS, xf, np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4), 0.5, 0.1, T)
if it:
    r = pb.solve(x0, np.zeros((B, T, N_U)), n_lqr_iter=it); X, U = r["X"], r["U"]; mu = to_dev(np.full(B, 0.5 ** it))
else:
    X, _ = pb.rollout(x0, np.zeros((B, T, N_U))); U = torch.zeros((B, T, N_U), dtype=torch.float64, device="cuda"); mu = to_dev(np.ones(B))
tiles = pb.make_tiles(X, U); K = empty((B, T, N_U, N_X)); d = empty((B, T, N_U)); lib = _lib.load()
def run(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        _lib.check(lib.dpilqr_backward_pass_tiles_blocks(B, T, N_X, N_U, 4, 2, ptr(tiles), ptr(mu), ptr(K), ptr(d), None, None, None, stream_handle()))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
run(3)
for chunk in (20, reps, reps):
    us = run(chunk)
    print(f"B={B} reps={chunk} realistic_it={it}: {us:.1f} us/launch  {B * (BWD_READ_BYTES + BWD_WRITE_BYTES) / us / 1e3 / 80:.1f} % of 8 TB/s")
