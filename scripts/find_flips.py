"""Which items does the GPU decide differently from the oracle's own solve?  (Run on the GPU box.)

For three families -- cfg2 (5 DoubleInt4D, T = 50), 8 Unicycle4D (T = 100), 10 Quadcopter6D from hover (T = 75), the
scripts/analysis.py scenario distribution seeded per item -- solve on the GPU with the decision trace and on the host
with the oracle, and write the seeds whose traces differ (and a few that do not, as controls) together with the GPU's
trace to gpurun_out/flips/<family>.npz.  tests/golden/make_golden.py (g9_chaos) then runs the REAL reference on exactly
those seeds and on 32 perturbed copies of each: the fixture that says whether the reference determines its own result.

    python scripts/find_flips.py [out_dir]
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import dpilqr_amd as dp  # noqa: E402
from dpilqr_amd.util import random_setup  # noqa: E402
from oracle import oracle as orc  # noqa: E402

FAMILIES = {   # name: (model enum, k, n_s, n_c, n_d, T, seeds, hover)
    "cfg2": (0, 5, 4, 2, 2, 50, range(0, 4096), False),
    "uni8": (3, 8, 4, 2, 2, 100, range(300, 300 + 192), False),
    "quad10": (4, 10, 6, 3, 3, 75, range(300, 300 + 96), True),
}


def inputs(model, k, ns, nc, nd, T, seeds, hover):
    x0 = np.zeros((len(seeds), k * ns)); xf = np.zeros((len(seeds), k * ns))
    for j, s in enumerate(seeds):
        np.random.seed(s)
        a, b = random_setup(k, ns, is_rotation=False, rel_dist=k, var=k / 2, n_d=nd, random=True, energy=10.0)
        x0[j], xf[j] = a.ravel(), b.ravel()
    Q, R = (50.0 * np.eye(6), np.eye(3)) if ns == 6 else (np.diag([1.0, 1, 0, 0]), np.eye(2))
    U0 = np.zeros((len(seeds), T, k * nc))
    if hover:
        U0[:, :, 0::3] = 9.80665
    return x0, xf, U0, Q, R, 1000.0 * np.eye(ns)


def main():
    out = Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "flips"
    out.mkdir(parents=True, exist_ok=True)
    for name, (model, k, ns, nc, nd, T, seeds, hover) in FAMILIES.items():
        seeds = list(seeds)
        x0, xf, U0, Q, R, Qf = inputs(model, k, ns, nc, nd, T, seeds, hover)
        pb = dp.ProblemBatch([model] * k, [nd] * k, xf, Q, R, Qf, 0.5, 0.1, T)
        g = {key: v.cpu().numpy() for key, v in pb.solve(x0, U0, trace=True).items()}
        proto = orc.Problem([model] * k, [nd] * k, xf[0], Q, R, Qf, 0.5, 0.1, T)
        o = orc.solve_batch(proto, x0, xf, U0, trace=True)
        acc_g = np.nan_to_num(g["trace"][:, :, 1], nan=-9.0); acc_o = np.nan_to_num(o["trace"][:, :, 1], nan=-9.0)
        differ = (g["n_bwd"] != o["n_bwd"]) | (acc_g != acc_o).any(axis=1) | (g["status"] != o["status"])
        idx = np.nonzero(differ)[0]
        ctl = np.nonzero(~differ)[0][:8]
        keep = np.concatenate([idx, ctl])
        np.savez_compressed(out / f"{name}.npz", seeds=np.asarray(seeds)[keep], flipped=differ[keep], gpu_n_bwd=g["n_bwd"][keep],
                            gpu_status=g["status"][keep], gpu_accept=acc_g[keep].astype(np.int8), gpu_J=g["J"][keep],
                            oracle_n_bwd=o["n_bwd"][keep], oracle_accept=acc_o[keep].astype(np.int8), oracle_J=o["J"][keep],
                            first_flip=np.array([int(np.argmax(acc_g[i] != acc_o[i])) for i in keep]))
        print(f"{name}: {len(seeds)} items, {len(idx)} decided differently from the oracle's own solve ({100 * differ.mean():.2f} %); "
              f"kept {len(keep)}; first differing iteration min/median {np.min([np.argmax(acc_g[i] != acc_o[i]) for i in idx]) if len(idx) else -1} / "
              f"{np.median([np.argmax(acc_g[i] != acc_o[i]) for i in idx]) if len(idx) else -1}", flush=True)


if __name__ == "__main__":
    main()
