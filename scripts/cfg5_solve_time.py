#!/usr/bin/env python
"""BASELINE config 5 as it is stated -- ONE 20-agent problem (14 Quadcopter12D + 6 zero-padded humans, n_x = 240, n_u = 80, T = 150) --
solved whole, fp64 and fp32: wall time, iterations, per-iteration time; with the team of workgroups (default) and without
(DPILQR_BIG_TEAM=0, a process of its own).   python scripts/cfg5_solve_time.py [n_lqr_iter=8]"""
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
if "--arm" in sys.argv:
    import numpy as np
    import torch
    sys.path.insert(0, str(ROOT))
    import dpilqr_amd as dp
    sys.path.insert(0, str(ROOT / "tests"))
    from test_gpu_big import _cfg5_batch
    iters = int(sys.argv[sys.argv.index("--arm") + 1])
    models, nd, x0, xf, Q, R, Qf, U0, T = _cfg5_batch(True, (6001,))
    pb = dp.ProblemBatch(models, nd, xf, Q, R, Qf, 0.5, 0.1, T)
    for dtype in (torch.float64, torch.float32):
        pb.solve(x0, U0, n_lqr_iter=2, dtype=dtype); torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = pb.solve(x0, U0, n_lqr_iter=iters, dtype=dtype)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        nb = int(r["n_bwd"][0]); nf = int(r["n_fwd"][0])
        print(f"{'fp64' if dtype == torch.float64 else 'fp32'}: whole solve {1e3 * t:7.1f} ms, {nb} iterations ({nf} candidates costed), "
              f"{1e3 * t / max(nb, 1):6.1f} ms per iteration, status {int(r['status'][0])}, J {float(r['J'][0]):.6g}", flush=True)
    sys.exit(0)
iters = sys.argv[1] if len(sys.argv) > 1 else "8"
for name, env in (("team of workgroups (default)", {}), ("single workgroup (DPILQR_BIG_TEAM=0)", {"DPILQR_DEBUG_ROUTES": "1", "DPILQR_BIG_TEAM": "0"})):
    print(f"== {name}", flush=True)
    subprocess.run([sys.executable, __file__, "--arm", iters], check=True, env={**os.environ, **env}, timeout=600)
