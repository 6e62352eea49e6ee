#!/usr/bin/env python
"""The large-cluster sweep's team of workgroups (csrc/riccati_big.hpp, BigTeam) against the single-workgroup pass: the same gains
bit for bit, and what the team buys, per batch size.  Each arm runs in a process of its own (DPILQR_BIG_TEAM is read at launch).
    python scripts/big_team_check.py [B ...]"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "--arm":
    import time
    import numpy as np
    import torch
    sys.path.insert(0, str(ROOT))
    src = (ROOT / "scripts" / "bench_big.py").read_text().split("for B in")[0]
    exec(src)
    out = {}
    for B in [int(a) for a in sys.argv[3:]]:
        x0 = np.zeros((B, n)); xf = np.zeros((B, n))
        for s in range(B):
            np.random.seed(100 + s)
            a, b = random_setup(k, 12, is_rotation=False, rel_dist=k, var=k / 2, n_d=3, random=True, energy=100.0)
            x0[s], xf[s] = a.ravel(), b.ravel()
        U0 = np.zeros((B, T, m)); U0[:, :, [4 * i + 3 for i in range(14)]] = 9.80665 * 63.0 / 2000.0
        pb = dp.ProblemBatch(models, n_dims, xf, Q, R, Qf, 0.5, 0.1, T)
        for dtype in (torch.float64, torch.float32):
            X, J = pb.rollout(x0, U0, dtype=dtype)
            Ud = torch.as_tensor(U0, dtype=dtype, device="cuda")
            mu = torch.ones(B, dtype=torch.float64, device="cuda")
            K, d = pb.backward_pass(X, Ud, mu, dtype=dtype)
            t = timeit(lambda: pb.backward_pass(X, Ud, mu, dtype=dtype))
            name = "fp64" if dtype == torch.float64 else "fp32"
            out[f"K_{B}_{name}"] = K.cpu().numpy(); out[f"d_{B}_{name}"] = d.cpu().numpy(); out[f"t_{B}_{name}"] = t
    np.savez(sys.argv[2], **out)
    sys.exit(0)

import numpy as np
Bs = sys.argv[1:] or ["1", "4", "32"]
res = {}
for arm, env in (("team", {}), ("alone", {"DPILQR_DEBUG_ROUTES": "1", "DPILQR_BIG_TEAM": "0"})):
    f = f"/tmp/big_team_{arm}.npz"
    subprocess.run([sys.executable, __file__, "--arm", f, *Bs], check=True, env={**os.environ, **env}, timeout=600)
    res[arm] = np.load(f)
for B in Bs:
    for name in ("fp64", "fp32"):
        same = all(np.array_equal(res["team"][f"{q}_{B}_{name}"], res["alone"][f"{q}_{B}_{name}"], equal_nan=True) for q in "Kd")
        print(f"B={int(B):4d} {name}: backward pass alone {1e3 * res['alone'][f't_{B}_{name}']:7.2f} ms, team {1e3 * res['team'][f't_{B}_{name}']:7.2f} ms, "
              f"gains {'bit-identical' if same else 'DIFFERENT'}", flush=True)
        if not same:      # where: per item the LAST horizon step (the sweep runs backwards: the first one computed) that differs
            for q in "Kd":
                a, b = res["team"][f"{q}_{B}_{name}"], res["alone"][f"{q}_{B}_{name}"]
                ne = ~((a == b) | (np.isnan(a) & np.isnan(b)))
                for item in np.nonzero(ne.reshape(a.shape[0], -1).any(1))[0][:6]:
                    steps = np.nonzero(ne[item].reshape(a.shape[1], -1).any(1))[0]
                    print(f"      {q}: item {item}: {int(ne[item].sum())} entries differ, steps {steps.min()}..{steps.max()}, "
                          f"at step {steps.max()}: {int(ne[item, steps.max()].sum())} entries, max |diff| {np.nanmax(np.abs(a[item, steps.max()] - b[item, steps.max()])):.3e}"
                          f" of max {np.nanmax(np.abs(b[item, steps.max()])):.3e}", flush=True)
