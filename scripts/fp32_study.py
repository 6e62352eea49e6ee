#!/usr/bin/env python
"""BASELINE config 5: fp32 vs fp64 tolerance study.

For a set of configurations -- cfg2 (5 x DoubleInt4D, T = 50), a Quadcopter6D cluster (cfg4's model, T = 75), a small
heterogeneous team, and config 5 itself (14 Quadcopter12D + 6 zero-padded humans, n_x = 240, n_u = 80, T = 150) -- it
runs the SAME passes and the SAME whole solves in fp64 and in fp32 on the GPU (the *_f32 entry points of
include/dpilqr_hip.h) and, where it finishes in reasonable time, in the CPU oracle (fp64), and prints

  per pass   relative error of the rollout, of the gains (K, d) of one backward pass and of the ten forward passes, fp32
             against fp64 at the same operating point (the fp64 iterate after two iLQR iterations);
  per solve  the decision-flip rate (items whose iteration / forward-pass counts or final status differ), the
             relative difference of the final cost and trajectory on the items whose decisions agree, and the same
             two figures for fp64-GPU against the fp64 oracle (the noise floor of the comparison).

    python scripts/fp32_study.py [--out profiles/r02_fp32_study.json] [--quick]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import dpilqr_amd as dp  # noqa: E402
from dpilqr_amd.util import random_setup  # noqa: E402

G = 9.80665


def rel(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(a.shape[0], -1); b = np.asarray(b, dtype=np.float64).reshape(b.shape[0], -1)
    return np.abs(a - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), 1e-300)


def scenario(models, n_dims, T, B, seed0, energy, radius=0.5, dt=0.1):
    k = len(models)
    ns, nc = dp.batch.MODEL_DIMS[models[0]]
    nd = 3 if ns >= 6 else 2
    x0 = np.zeros((B, k * ns)); xf = np.zeros((B, k * ns))
    for s in range(B):
        np.random.seed(seed0 + s)
        a, b = random_setup(k, ns, is_rotation=False, rel_dist=k, var=k / 2, n_d=nd, random=True, energy=energy)
        x0[s], xf[s] = a.ravel(), b.ravel()
    Q, R = [], []
    for mdl in models:
        if mdl == 8:
            Q.append(np.diag([1.0, 1, 1, 0, 0, 0] + [0.0] * 6)); R.append(np.diag([1.0, 1, 1e-9, 1e-9]))
        elif ns == 4:
            Q.append(np.diag([1.0, 1, 0, 0])); R.append(np.eye(2))
        elif ns == 6:
            Q.append(50.0 * np.eye(6)); R.append(np.eye(3))
        else:
            Q.append(np.eye(ns)); R.append(np.eye(nc))
    Q, R = np.stack(Q), np.stack(R)
    Qf = np.stack([1000.0 * np.eye(ns)] * k)
    U0 = np.zeros((B, T, k * nc))
    for i, mdl in enumerate(models):
        if mdl == 4:
            U0[:, :, nc * i] = G
        if mdl == 7:
            U0[:, :, nc * i + 3] = G * 63.0 / 2000.0
    pb = dp.ProblemBatch(models, n_dims, xf, Q, R, Qf, radius, dt, T)
    return pb, x0, xf, U0, (Q, R, Qf)


def study(name, models, n_dims, T, B, seed0, energy, n_lqr_iter, oracle_items):
    pb, x0, xf, U0, (Q, R, Qf) = scenario(models, n_dims, T, B, seed0, energy)
    f64, f32 = torch.float64, torch.float32
    out = dict(config=name, k=len(models), n_x=pb.n_x, n_u=pb.n_u, T=T, items=B, n_lqr_iter=n_lqr_iter)
    # ---- per pass, at the fp64 iterate after two iterations
    r2 = pb.solve(x0, U0, n_lqr_iter=2)
    X, U = r2["X"], r2["U"]
    mu = torch.full((B,), 0.125, dtype=f64, device="cuda")
    Xr64, Jr64 = pb.rollout(x0, U.cpu().numpy())
    Xr32, Jr32 = pb.rollout(x0, U.cpu().numpy(), dtype=f32)
    K64, d64 = pb.backward_pass(X, U, mu)
    K32, d32 = pb.backward_pass(X, U, mu, dtype=f32)
    al = np.array(dp._lib.alphas())
    Xf64, Uf64, Jf64 = pb.forward_pass(X, U, K64, d64, al)
    Xf32, Uf32, Jf32 = pb.forward_pass(X, U, K64, d64, al, dtype=f32)          # same (fp64) gains, fp32 arithmetic
    fin = torch.isfinite(Jf64) & (Jf64.abs() < 1e12)
    efwd = (Xf32.double() - Xf64).abs().amax(dim=(2, 3)) / Xf64.abs().amax(dim=(2, 3)).clamp_min(1e-300)
    out["pass"] = dict(
        rollout_X=float(np.median(rel(Xr32.cpu().numpy(), Xr64.cpu().numpy()))),
        rollout_J=float(np.median(np.abs(Jr32.cpu().numpy() - Jr64.cpu().numpy()) / np.abs(Jr64.cpu().numpy()))),
        K_median=float(np.median(rel(K32.cpu().numpy(), K64.cpu().numpy()))), K_max=float(rel(K32.cpu().numpy(), K64.cpu().numpy()).max()),
        d_median=float(np.median(rel(d32.cpu().numpy(), d64.cpu().numpy()))), d_max=float(rel(d32.cpu().numpy(), d64.cpu().numpy()).max()),
        forward_X_median=float(efwd[fin].nanmedian()) if fin.any() else None,
        forward_X_max=float(efwd[fin][~torch.isnan(efwd[fin])].max()) if fin.any() else None,
        forward_fp32_nan_where_fp64_finite=float(torch.isnan(efwd[fin]).double().mean()) if fin.any() else None,
        forward_J_median=float(((Jf32 - Jf64).abs() / Jf64.abs())[fin].nanmedian()) if fin.any() else None,
        forward_candidates_finite_fp64=float(fin.double().mean()), forward_candidates_finite_fp32=float(torch.isfinite(Jf32).double().mean()))
    # ---- whole solves
    t0 = time.perf_counter(); s64 = pb.solve(x0, U0, n_lqr_iter=n_lqr_iter); torch.cuda.synchronize(); t64 = time.perf_counter() - t0
    t0 = time.perf_counter(); s32 = pb.solve(x0, U0, n_lqr_iter=n_lqr_iter, dtype=f32); torch.cuda.synchronize(); t32 = time.perf_counter() - t0

    def compare(a, b):
        same = ((a["n_bwd"] == b["n_bwd"]) & (a["n_fwd"] == b["n_fwd"]) & (a["status"] == b["status"])).cpu().numpy().astype(bool)
        Ja, Jb = a["J"].cpu().numpy(), b["J"].cpu().numpy()
        eX = rel(a["X"].double().cpu().numpy(), b["X"].double().cpu().numpy())
        with np.errstate(invalid="ignore"):
            eJ = np.abs(Ja - Jb) / np.maximum(np.abs(Jb), 1e-300)
        eJ = np.where(np.isnan(Ja) & np.isnan(Jb), 0.0, eJ)     # both report the NaN cost of a rejected last candidate (quirk Q2)
        return dict(decision_flip_rate=float(1.0 - same.mean()),
                    J_rel_median_same=float(np.nanmedian(eJ[same])) if same.any() else None,
                    J_rel_max_same=float(np.nanmax(eJ[same])) if same.any() else None,
                    X_rel_median_same=float(np.median(eX[same])) if same.any() else None,
                    X_rel_max_same=float(eX[same].max()) if same.any() else None,
                    J_rel_median_all=float(np.nanmedian(eJ)), frac_within_1e5=float((eX < 1e-5).mean()))
    out["solve_fp32_vs_fp64"] = compare(s32, s64)
    out["solve_seconds"] = dict(fp64=t64, fp32=t32)
    out["mean_iterations"] = dict(fp64=float(s64["n_bwd"].double().mean()), fp32=float(s32["n_bwd"].double().mean()))
    if oracle_items:
        from oracle import oracle as orc
        no = min(oracle_items, B)
        o = dict(X=[], J=[], n_bwd=[], n_fwd=[], status=[])
        for i in range(no):
            p = orc.Problem(models, n_dims, xf[i], Q, R, Qf, 0.5, 0.1, T)
            r = p.solve(x0[i], U0[i], n_lqr_iter=n_lqr_iter)
            for key in o:
                o[key].append(r[key])
        oracle = {key: torch.as_tensor(np.array(v)) for key, v in o.items()}
        head = lambda s: {key: s[key][:no].cpu() for key in oracle}
        out["solve_fp64_vs_oracle"] = dict(items=no, **compare(head(s64), oracle))
        out["solve_fp32_vs_oracle"] = dict(items=no, **compare(head(s32), oracle))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    q = args.quick
    cfgs = [
        ("cfg2: 5 x DoubleInt4D, T=50", [0] * 5, [2] * 5, 50, 128 if q else 1024, 0, 10.0, 50, 64 if q else 256),
        ("cfg4 cluster: 5 x Quadcopter6D, T=75", [4] * 5, [3] * 5, 75, 64 if q else 256, 2000, 10.0, 50, 16 if q else 64),
        ("hetero: 2 x Quadcopter12D + padded human, T=50", [7, 7, 8], [3, 3, 2], 50, 32 if q else 128, 4000, 3.0, 30, 8 if q else 32),
        ("cfg5: 14 x Quadcopter12D + 6 padded humans, T=150", [7] * 14 + [8] * 6, [3] * 14 + [2] * 6, 150, 4 if q else 16, 6000,
         100.0, 6 if q else 12, 1 if q else 2),
    ]
    results = []
    for c in cfgs:
        t0 = time.perf_counter()
        r = study(*c)
        r["study_seconds"] = time.perf_counter() - t0
        results.append(r)
        p, s = r["pass"], r["solve_fp32_vs_fp64"]
        print(f"\n== {r['config']}  (n_x={r['n_x']}, n_u={r['n_u']}, {r['items']} items)")
        print(f"   pass   fp32 vs fp64: rollout X {p['rollout_X']:.1e}  K {p['K_median']:.1e} (max {p['K_max']:.1e})  d {p['d_median']:.1e}"
              f"  forward X {p['forward_X_median']:.1e} (max {p['forward_X_max']:.1e})  J {p['forward_J_median']:.1e}")
        print(f"   solve  fp32 vs fp64: decision flips {100 * s['decision_flip_rate']:.1f} %  J {s['J_rel_median_same']} / X {s['X_rel_median_same']} "
              f"(median, same decisions); within 1e-5: {100 * s['frac_within_1e5']:.1f} %")
        if "solve_fp64_vs_oracle" in r:
            a, b = r["solve_fp64_vs_oracle"], r["solve_fp32_vs_oracle"]
            print(f"   vs oracle ({a['items']} items): fp64 flips {100 * a['decision_flip_rate']:.1f} %, within 1e-5 {100 * a['frac_within_1e5']:.1f} % | "
                  f"fp32 flips {100 * b['decision_flip_rate']:.1f} %, within 1e-5 {100 * b['frac_within_1e5']:.1f} %")
        print(f"   solve time fp64 {r['solve_seconds']['fp64']:.2f} s, fp32 {r['solve_seconds']['fp32']:.2f} s", flush=True)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(json.dumps(results, indent=1))


if __name__ == "__main__":
    main()
