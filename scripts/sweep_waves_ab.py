#!/usr/bin/env python
"""Launch time of the fused cfg2 sweep per batch size (interacting agents): run once as it is and once with DPILQR_MFMA_WAVES=8 to
compare three against two wavefronts per SIMD (round 3: 955 / 744 / 502 / 347 us at 6144 / 4096 / 3072 / 2048 items against
944 / 668 / 569 / 347: the same per full window; whole rounds of the chosen layout decide the rest).
    python scripts/sweep_waves_ab.py [B ...]        (B <= 1024: the two-wavefront team kernel; DPILQR_NO_TEAM=1 switches it off)"""
import sys, time, numpy as np, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import dpilqr_amd as dp
from dpilqr_amd.device import to_dev
from bench import scenarios, K_AGENTS, T, N_U
SIZES = [int(a) for a in sys.argv[1:]] or [6144, 4096, 3072, 2048]
for B in SIZES:
    x0, xf = scenarios(0, B)
    pb = dp.ProblemBatch([0] * K_AGENTS, [2] * K_AGENTS, xf, np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4), 0.5, 0.1, T)
    r = pb.solve(x0, np.zeros((B, T, N_U)), n_lqr_iter=2)
    X, U = r["X"], r["U"]; mu = to_dev(np.full(B, 0.125))
    for _ in range(3): pb.backward_pass_fused(X, U, mu)
    torch.cuda.synchronize(); ts = []
    for _ in range(9):
        t0 = time.perf_counter(); pb.backward_pass_fused(X, U, mu); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"B={B}: fused sweep {np.median(ts)*1e6:.0f} us", flush=True)
