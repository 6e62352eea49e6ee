#!/usr/bin/env python
"""bench.py -- headline benchmark of the batched iLQR hot path on MI355X.

Workload (BASELINE.json configs[1], "cfg2"): a batch of 1024 independent 5-agent DoubleIntDynamics4D
sub-problems per GPU, horizon T=50, fp64, the Monte-Carlo scenario distribution of the reference's
scripts/analysis.py:45-69,140-143 (seed s: np.random.seed(s); random_setup(5,4,rel_dist=5,var=2.5,
energy=10); Q=diag(1,1,0,0), R=I, Qf=1000 I, radius 0.5, dt 0.1, U0=0, tol 1e-3, n_lqr_iter 50).
One "step" = one complete ilqrSolver.solve of every sub-problem of one such batch (device resident:
x0/xf/U0 of all K steps are in HBM before the clock starts).  The K batches (K x 1024 different seeds per
GPU) are handed to the solver together, the way a Monte-Carlo driver would, and it keeps a WINDOW of 6144
sub-problems in flight: sub-problems need 1..25 iLQR iterations, so finished ones are retired on the device
and replaced by not-yet-started ones, and every launch of the hot kernels works on ~6144 sub-problems (two rounds of three sweep wavefronts per SIMD).  All
K x 1024 solves complete inside the timed region.  For N>1 the region also contains the path's one
collective, an RCCL all-gather of the converged (X, U, J, status, n_bwd, n_fwd) of all ranks, issued chunk by chunk
on a side stream while the rest of the job still solves (dpilqr_amd/sharding.py ResultBuffers; buffers pre-allocated
and the collective warmed on them outside the clock).

The timed K-step job is REPEATED on disjoint seeds inside one invocation -- --reps times if given, otherwise five times and on
until the repetitions add up to --min-seconds (2 s) of timed work, at most --max-reps (48) -- and `value` / `ms_per_step` are the
MEDIAN repetition (max over ranks each); the spread is reported beside them (`repetitions`).

    python bench.py [--gpus N --steps K --warmup W]          (N > 1 without a launcher: starts its own N ranks)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
A rank count that is not --gpus, or fewer visible GPUs than ranks (RCCL), ends the run with a non-zero exit code.

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

K_AGENTS, N_S, N_C, T = 5, 4, 2, 50
N_X, N_U = K_AGENTS * N_S, K_AGENTS * N_C
# algorithmic bytes of ONE Riccati backward pass of ONE cfg2 sub-problem, tiles through HBM:
#   read  8*(T*S + n_x + n_x^2),  S = 2 n_x^2 + 2 n_x n_u + n_u^2 + n_x + n_u = 1330 doubles
#   write 8*T*(n_u n_x + n_u)
S_TILE = 2 * N_X * N_X + 2 * N_X * N_U + N_U * N_U + N_X + N_U
BWD_READ_BYTES = 8 * (T * S_TILE + N_X + N_X * N_X)       # 535 360
BWD_WRITE_BYTES = 8 * T * (N_U * N_X + N_U)               # 84 000
HBM_PEAK_GBS = 8000.0                                     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# The fused sweep (the default for this workload: linearize / quadraticize evaluated inside the sweep, no tile records)
# reads only the trajectory and writes the gains: SURVEY.md 8(d) "fused variant"
FUSED_READ_BYTES = 8 * ((T + 1) * N_X + T * N_U)          # 12 160
FUSED_BYTES = FUSED_READ_BYTES + BWD_WRITE_BYTES          # 96 160
# ... so it is bound by the fp64 pipe, not by HBM.  SURVEY.md 8(d): dense flops of one backward pass,
#   T (4 n_x^3 + 8 n_x^2 n_u + 6 n_x n_u^2 + 2/3 n_u^3) = 3.83 Mflop at cfg2
BWD_FLOPS = T * (4 * N_X ** 3 + 8 * N_X ** 2 * N_U + 6 * N_X * N_U ** 2 + 2.0 * N_U ** 3 / 3.0)
# fp64 peak: 256 CUs x 4 SIMDs x 16 FMA/clk x 2 x 2.4 GHz = 78.6 TFLOP/s -- the vector ALU and the fp64 MFMA share one pipe
# of that width (scripts/ubench/mfma_f64.hip measures 76.6 TFLOP/s sustained on this part, profiles/r02_mfma_f64_peak.txt)
FP64_PEAK_TFLOPS = 78.6


def useful_flops_per_step(n, m, ns, k):
    """Flops of one Riccati step ON DATA when [A|B] is block diagonal with k blocks of n_s rows (MultiDynamicalModel.linearize,
    dynamics.py:173-186) -- the products the sweep actually has to form; the dense count above also prices the
    multiplications by structural zeros, which the kernel (correctly) skips, and so overstates pipe utilisation (round-2
    advice).  Per step, with the reference's association (control.py:131-146):
      S1  A^T P, A^T p, B^T (P + mu I), B^T p      n_s terms per output:  2 n_s (n^2 + n + m n + m)
      S2  (A^T P) A, (B^T P') B, (B^T P') A        n_s terms per output:  2 n_s (n^2 + m^2 + m n)   + n^2 + m^2 + m n adds
      S3  LU of Q_uu and n + 1 solves:             2/3 m^3 + 2 m^2 (n + 1)
      S4  K^T Q_uu:                                2 n m^2
      S5  (K^T Q_uu)[K|d], K^T [Q_ux|Q_u] (its transpose supplies Q_ux^T K, Q_ux^T d):  2 * 2 n m (n + 1)   + 3 n (n + 1) adds
      S6  P <- (P + P^T)/2:                        2 n^2"""
    s1 = 2 * ns * (n * n + n + m * n + m)
    s2 = 2 * ns * (n * n + m * m + m * n) + n * n + m * m + m * n
    s3 = 2.0 * m ** 3 / 3.0 + 2 * m * m * (n + 1)
    s4 = 2 * n * m * m
    s5 = 2 * 2 * n * m * (n + 1) + 3 * n * (n + 1)
    s6 = 2 * n * n
    return s1 + s2 + s3 + s4 + s5 + s6


BWD_USEFUL_FLOPS = T * useful_flops_per_step(N_X, N_U, N_S, K_AGENTS)     # 1.93 Mflop at cfg2 (dense count: 3.83)


def kernel_source_sha16():
    """Identifies the sweep's source: profiles/riccati_traffic.json (the PMC pass behind roofline.traffic) names the source it
    was measured on, and a stale file is reported as traffic = null instead of silently describing another kernel."""
    import hashlib
    h = hashlib.sha256()
    for f in ("riccati_mfma.hpp", "riccati_mfma_lane.inc", "tu_riccati.hip"):
        h.update((ROOT / "dpilqr_amd" / "csrc" / f).read_bytes())
    return h.hexdigest()[:16]


def scenarios(seed0, B):
    from dpilqr_amd.util import random_setup
    x0 = np.zeros((B, N_X)); xf = np.zeros((B, N_X))
    state = np.random.get_state()
    for i in range(B):
        np.random.seed(seed0 + i)
        a, b = random_setup(K_AGENTS, N_S, is_rotation=False, rel_dist=K_AGENTS, var=K_AGENTS / 2, n_d=2, random=True,
                            energy=10.0)
        x0[i], xf[i] = a.ravel(), b.ravel()
    np.random.set_state(state)
    return x0, xf


def cpu_baseline(x0, xf, sample):
    """The CPU restatement (oracle/, OpenMP over the batch) on the host cores this process may actually use: the affinity
    mask capped by the cgroup's CPU quota (the GPU boxes show 256 hardware threads behind a quota of 16 CPUs; 256 OpenMP
    threads then run throttled at 2.3 k sub-problems/s against 5.2 k with 16 -- profiles/r03_cpu_probe.txt)."""
    from oracle import oracle as orc
    cores = orc.usable_cores()
    Q, R, Qf = np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4)
    proto = orc.Problem([0] * K_AGENTS, [2] * K_AGENTS, xf[0], Q, R, Qf, 0.5, 0.1, T)
    orc.solve_batch(proto, x0[:4 * cores], xf[:4 * cores], np.zeros((4 * cores, T, N_U)), n_threads=cores)  # warm
    t0 = time.perf_counter()
    o = orc.solve_batch(proto, x0[:sample], xf[:sample], np.zeros((sample, T, N_U)), n_threads=cores)
    dt = time.perf_counter() - t0
    return dict(value=sample / dt, unit="subproblems/s", cores=cores, kind="port",
                sample=f"first {sample} sub-problems of the rank-0 job, oracle/ilqr_oracle.c with OpenMP on {cores} threads "
                       f"(affinity {len(os.sched_getaffinity(0))} hardware threads, cgroup cpu.max "
                       f"{_read('/sys/fs/cgroup/cpu.max')}), {dt:.2f} s wall = {dt * cores:.0f} core-seconds"), o


def cpu_baseline_numpy(x0, xf, o, n=24):
    """SURVEY 8(d)'s other CPU baseline: the NumPy restatement with the reference's own per-step Python structure
    (oracle/numpy_port.py, pinned on the reference's golden solves), single process -- the reference's speed class
    (the reference itself does 6-7.5 sub-problems/s per core in the build container)."""
    from oracle import numpy_port
    Q, R, Qf = np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4)
    t0 = time.perf_counter()
    same = 0
    for i in range(n):
        r = numpy_port.cfg_solver([0] * K_AGENTS, [2] * K_AGENTS, xf[i], Q, R, Qf, 0.5, 0.1, T).solve(x0[i])
        same += int(len(r["trace"]) == o["n_bwd"][i])
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="subproblems/s", cores=1, kind="port",
                sample=f"first {n} sub-problems of the rank-0 job, oracle/numpy_port.py (per-step Python + NumPy + np.linalg.solve, "
                       f"per-agent dynamics through ctypes), one process, {dt:.2f} s; iteration counts equal to the C port's on "
                       f"{same}/{n}")


# ---------------------------------------------------------------------------------------------------------------------------
# BASELINE.json's other configurations in the same line (`configs`), run AFTER and OUTSIDE the headline's timed region: cfg3 and
# cfg4 as many-scenario DP-iLQR calls at their stated sizes (reference path: dpilqr/distributed.py:25-103, one call per scenario
# there), cfg5 as the ONE problem it is (fp64 and fp32), each with the CPU restatement timed beside it on a stated sample.
G_ACC = 9.80665
CFG_SCEN = {   # name: (model enum, class name, agents, n_s, n_c, n_d, T, scenarios, CPU-sample scenarios)
    "cfg3": (3, "UnicycleDynamics4D", 15, 4, 2, 2, 100, 4096, 64),
    "cfg4": (4, "QuadcopterDynamics6D", 10, 6, 3, 3, 75, 8192, 256),
}


def dense_flops_per_pass(n, m, T):
    """T (4 n^3 + 8 n^2 m + 6 n m^2 + 2/3 m^3)"""
    return T * (4.0 * n ** 3 + 8.0 * n * n * m + 6.0 * n * m * m + 2.0 * m ** 3 / 3.0)


def scenario_config(name, scenarios=None, cpu=True):
    """One many-scenario solve_distributed call of cfg3 / cfg4 (X = x0: the reference's first call): wall time with the inputs
    resident in HBM and the stitched results left there, distinct sub-problems per second; the mid-size workgroup sweep
    (k_riccati_wg, fused form -- what the solve loop launches for the configuration's full-size cluster) timed alone for its own
    roofline; the C oracle on the distinct sub-problems of the first few scenarios as the CPU baseline."""
    import torch
    import dpilqr_amd as dp
    from dpilqr_amd.dispatch import solve_scenarios_distributed
    from dpilqr_amd.util import random_setup_batch
    mdl, cls, k, ns, nc, nd, T, S, s_cpu = CFG_SCEN[name]
    S = int(scenarios or S)
    Q, R = (np.diag([1.0, 1, 0, 0]), np.eye(2)) if ns == 4 else (50.0 * np.eye(6), np.eye(3))     # scripts/analysis.py:62-69
    Qf, dt, radius = 1000.0 * np.eye(ns), 0.1, 0.5
    x0, xf = random_setup_batch((0, S), k, ns, var=k / 2, n_d=nd, energy=10.0)      # np.random.seed(s); random_setup(...), s = 0..S-1
    ids = [100 + i for i in range(k)]
    Model = getattr(dp, cls)
    xf0 = xf[0].cpu().numpy()
    prob = dp.ilqrProblem(dp.MultiDynamicalModel([Model(dt, i) for i in ids]),
                          dp.GameCost([dp.ReferenceCost(xf0[i * ns:(i + 1) * ns], Q.copy(), R.copy(), Qf.copy(), id_) for i, id_ in enumerate(ids)],
                                      dp.ProximityCost([ns] * k, radius, [nd] * k)))
    U0 = torch.zeros((S, T, k * nc), dtype=torch.float64, device="cuda")
    if mdl == 4:
        U0[:, :, 0::3] = G_ACC                                                      # hover warm start (examples.py:122)
    sw = min(S, 256)
    solve_scenarios_distributed(prob, x0[:sw, None, :], U0[:sw], radius, xf=xf[:sw], device_out=True)      # warm: kernels, allocator
    torch.cuda.synchronize()
    walls = []
    for _ in range(2):
        t0 = time.perf_counter()
        Xd, Ud, J, info = solve_scenarios_distributed(prob, x0[:, None, :], U0, radius, xf=xf, device_out=True)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
    wall = min(walls)
    out = {"workload": f"{name}: {S} random-goal scenarios x {k} {cls}, T={T}, proximity-graph split (radius {radius}) -> one batched "
                       f"device solve per cluster size, stitched; scenarios s = 0..{S - 1} of scripts/analysis.py's distribution, inputs and "
                       "results resident in HBM",
           "scenarios": S, "wall_s": wall, "wall_s_runs": walls, "distinct_subproblems": int(info["n_unique"]),
           "subproblems_total": int(info["n_subproblems"]), "subproblems_per_s": info["n_unique"] / wall, "scenarios_per_s": S / wall,
           "cluster_sizes": {str(a): int(b) for a, b in sorted(info["sizes"].items())}, "stages_s": info["seconds"],
           "finite_frac": float(torch.isfinite(J).double().mean().item())}
    # the workgroup sweep alone, at the configuration's full cluster size: 2048 items at the iterate two iLQR iterations reach
    Bs = 2048
    n, m = k * ns, k * nc
    pbs = dp.ProblemBatch([mdl] * k, [nd] * k, xf[:Bs], Q, R, Qf, radius, dt, T)
    r2 = pbs.solve(x0[:Bs], U0[:Bs], n_lqr_iter=2)
    mu = torch.full((Bs,), 0.125, dtype=torch.float64, device="cuda")
    for _ in range(2):
        pbs.backward_pass_fused(r2["X"], r2["U"], mu)
    reps_ = 6
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps_):
        pbs.backward_pass_fused(r2["X"], r2["U"], mu)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps_
    useful = T * useful_flops_per_step(n, m, ns, k); dense = dense_flops_per_pass(n, m, T)
    out["sweep_roofline"] = {"bound": "fp64", "kernel": f"k_riccati_wg<{n},{m},{ns},{nc},true>", "items": Bs, "launch_ms": ms,
                             "launches_back_to_back": reps_, "achieved": Bs * useful / (ms * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": Bs * useful / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                             "flops_per_subproblem_pass": useful,
                             "dense_count": {"flops_per_subproblem_pass": dense, "achieved": Bs * dense / (ms * 1e-3) / 1e12,
                                             "frac": Bs * dense / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
                             "hbm_bytes_per_subproblem_pass": 8 * ((T + 1) * n + T * m + T * (m * n + m)),
                             "note": f"the fused mid-size sweep on {Bs} clusters of all {k} agents (the size that dominates a receding-horizon "
                                     "second call), timed alone between one pair of HIP events; flops on data as in `roofline`"}
    del pbs, r2
    if cpu:
        from oracle import oracle as orc
        cores = orc.usable_cores()
        _, _, _, inf = solve_scenarios_distributed(prob, x0[:s_cpu, None, :], U0[:s_cpu], radius, xf=xf[:s_cpu], device_out=True, audit=True)
        n_sub, t_cpu = 0, 0.0
        for kc, a in sorted(inf["audit"].items()):
            proto = orc.Problem([mdl] * kc, [nd] * kc, a["xf"][0], Q, R, Qf, radius, dt, T)
            t0 = time.perf_counter()
            orc.solve_batch(proto, a["x0"], a["xf"], a["U0"], n_threads=cores)
            t_cpu += time.perf_counter() - t0
            n_sub += len(a["J"])
        out["cpu_baseline"] = {"value": n_sub / t_cpu, "unit": "subproblems/s", "cores": cores, "kind": "port",
                               "sample": f"the {n_sub} distinct sub-problems of scenarios 0..{s_cpu - 1} (cluster sizes "
                                         f"{sorted(int(a_) for a_ in inf['audit'])}), oracle/ilqr_oracle.c, OpenMP over each size's batch on {cores} "
                                         f"threads, {t_cpu:.2f} s wall"}
    return out


def cfg5_config(cpu=True, n_lqr_iter=8):
    """BASELINE configs[4] as stated: ONE heterogeneous 20-agent problem (14 QuadcopterDynamics12D + 6 zero-padded HumanDynamics6D,
    n_x = 240, n_u = 80, T = 150), solved whole -- the large-cluster kernels (k_riccati_big as a team of workgroups,
    k_forward<KDIRECT>) -- in fp64 and in fp32 (the tolerance study's two arms), with the C oracle on one core beside them.
    Scenarios: tests/test_gpu_big.py::_cfg5_batch's (energy 100, hover warm start).  Seed 6010 is the MEASURED AND CHECKED one: the
    reference algorithm converges on it in four iterations and its trajectory is determined (3e-13 relative perturbations of x0 move
    the oracle's result by 2e-7), so GPU and oracle can be compared item for item.  Seed 6001 (profiles/r05_cfg5_solve.txt's) runs
    all `n_lqr_iter` iterations and is timed beside it for the per-iteration figure; its trajectory is chaotic (the same
    perturbation moves the oracle's own result by 150 %), so nothing is compared there."""
    import torch
    import dpilqr_amd as dp
    from dpilqr_amd.util import random_setup
    k, T = 20, 150
    models = [7] * 14 + [8] * 6
    nd = [3] * 14 + [2] * 6
    Q = np.stack([np.eye(12)] * 14 + [np.diag([1.0, 1, 1, 0, 0, 0] + [0.0] * 6)] * 6)
    R = np.stack([np.eye(4)] * 14 + [np.diag([1.0, 1, 1e-9, 1e-9])] * 6)
    Qf = np.stack([1000.0 * np.eye(12)] * k)
    U0 = np.zeros((1, T, 80))
    for i in range(14):
        U0[:, :, 4 * i + 3] = G_ACC * 63.0 / 2000.0

    def scenario(seed):
        np.random.seed(seed)
        a, b = random_setup(k, 12, is_rotation=False, rel_dist=k, var=k / 2, n_d=3, random=True, energy=100.0)
        return a.ravel()[None], b.ravel()[None]

    def timed(pb, x0, dtype):
        pb.solve(x0, U0, n_lqr_iter=2, dtype=dtype); torch.cuda.synchronize()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            r = pb.solve(x0, U0, n_lqr_iter=n_lqr_iter, dtype=dtype)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        nb = int(r["n_bwd"][0])
        return r, {"solve_s": min(ts), "solve_s_runs": ts, "iterations": nb, "candidates_costed": int(r["n_fwd"][0]),
                   "ms_per_iteration": 1e3 * min(ts) / max(nb, 1), "status": int(r["status"][0]), "J": float(r["J"][0]),
                   "solves_per_s": 1.0 / min(ts)}

    out = {"workload": f"cfg5: one 20-agent problem (14 QuadcopterDynamics12D + 6 zero-padded HumanDynamics6D), n_x=240, n_u=80, T={T}, "
                       f"whole solve, n_lqr_iter={n_lqr_iter}; seed 6010 (converges, checked against the oracle) and seed 6001 (all "
                       f"{n_lqr_iter} iterations, chaotic: timed only)"}
    x0, xf = scenario(6010)
    pb = dp.ProblemBatch(models, nd, xf, Q, R, Qf, 0.5, 0.1, T)
    res = {}
    for name, dtype in (("f64", torch.float64), ("f32", torch.float32)):
        res[name], out[name] = timed(pb, x0, dtype)
    X64 = res["f64"]["X"].cpu().numpy(); X32 = res["f32"]["X"].cpu().numpy().astype(np.float64)
    out["f32_vs_f64"] = {"rel_err_X": float(np.max(np.abs(X32 - X64)) / np.max(np.abs(X64))),
                         "same_iterations": out["f32"]["iterations"] == out["f64"]["iterations"]}
    x1, xf1 = scenario(6001)
    pb1 = dp.ProblemBatch(models, nd, xf1, Q, R, Qf, 0.5, 0.1, T)
    out["seed_6001"] = {name: timed(pb1, x1, dtype)[1] for name, dtype in (("f64", torch.float64), ("f32", torch.float32))}
    # the backward pass alone (k_riccati_big<double,12,4>, a team of workgroups for the one item): useful / dense flops over its time
    Xd, Ud = res["f64"]["X"], res["f64"]["U"]
    mu = torch.ones(1, dtype=torch.float64, device="cuda")
    pb.backward_pass(Xd, Ud, mu)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(3):
        pb.backward_pass(Xd, Ud, mu)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    useful = T * useful_flops_per_step(240, 80, 12, 20); dense = dense_flops_per_pass(240, 80, T)
    out["sweep_roofline"] = {"bound": "fp64", "kernel": "k_riccati_big<double,12,4>", "items": 1, "launch_ms": ms,
                             "achieved": useful / (ms * 1e-3) / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": useful / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, "flops_per_subproblem_pass": useful,
                             "dense_count": {"flops_per_subproblem_pass": dense, "achieved": dense / (ms * 1e-3) / 1e12,
                                             "frac": dense / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
                             "note": "ONE item: the chain of T = 150 dependent steps on a team of workgroups (nine of 256 CUs); the launch "
                                     "includes the workspace memset and the team reset"}
    if cpu:
        from oracle import oracle as orc
        proto = orc.Problem(models, nd, xf[0], Q, R, Qf, 0.5, 0.1, T)
        t0 = time.perf_counter()
        o = orc.solve_batch(proto, x0, xf, U0, n_lqr_iter=n_lqr_iter, n_threads=1)
        t_cpu = time.perf_counter() - t0
        errX = float(np.max(np.abs(X64[0] - o["X"][0])) / np.max(np.abs(o["X"][0])))
        out["cpu_baseline"] = {"value": 1.0 / t_cpu, "unit": "solves/s", "cores": 1, "kind": "port",
                               "sample": f"seed 6010's problem, whole ({int(o['n_bwd'][0])} iterations), oracle/ilqr_oracle.c on one thread "
                                         f"(one item: nothing to spread over cores), {t_cpu:.2f} s"}
        out["parity_vs_oracle"] = {"seed": 6010, "same_decision_trace": bool(int(o["n_bwd"][0]) == out["f64"]["iterations"] and
                                                                             int(o["n_fwd"][0]) == out["f64"]["candidates_costed"] and
                                                                             int(o["status"][0]) == out["f64"]["status"]),
                                   "rel_err_X": errX, "tolerance": 1e-5}
    return out


def _read(path):
    try:
        return Path(path).read_text().strip()
    except OSError:
        return "n/a"


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N child ranks of this very command line (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, what torch.distributed.run would set), let rank 0's JSON line through on the
    inherited stdout and return non-zero as soon as any rank fails (the others are then stopped: a rank alone would wait in
    its first collective for ever).  Replaces multiprocessing.Pool of the reference's distributed.py:80-97 at job level."""
    import signal
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in env:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(s.getsockname()[1])
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    procs = []

    def die_with_parent():       # runs in the child between fork and exec: nothing here has touched a GPU
        try:
            import ctypes
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGTERM)      # PR_SET_PDEATHSIG: a SIGKILLed launcher still takes its ranks down
        except OSError:
            pass

    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]],
                                      env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), start_new_session=True,
                                      preexec_fn=die_with_parent))

    def stop_all(sig=signal.SIGTERM):
        for p_ in procs:                 # exact process groups we started, nothing by pattern
            if p_.poll() is None:
                try:
                    os.killpg(p_.pid, sig)
                except ProcessLookupError:
                    pass

    def on_signal(signum, frame):        # the driver's timeout, a closed terminal: the ranks must not outlive the launcher
        stop_all()
        deadline = time.time() + 10.0
        while time.time() < deadline and any(p_.poll() is None for p_ in procs):
            time.sleep(0.05)
        stop_all(signal.SIGKILL)
        raise SystemExit(128 + signum)

    for sig in (signal.SIGTERM, signal.SIGHUP):
        signal.signal(sig, on_signal)
    rc = 0
    live = set(range(n))
    t_stop = None                        # when the survivors of a failed rank were told to stop
    try:
        while live:
            if t_stop is not None and time.time() - t_stop > 15.0:
                stop_all(signal.SIGKILL)         # a rank that ignores SIGTERM (stuck in a collective) is not waited for for ever
                t_stop = time.time()
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} of {n} exited with code {code}; stopping the other ranks", file=sys.stderr, flush=True)
                    stop_all()
                    t_stop = time.time()
            time.sleep(0.05)
    except KeyboardInterrupt:
        stop_all()
        rc = 130
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reps", type=int, default=None, help="repetitions of the timed K-step job (disjoint seeds); the median is reported.  "
                    "Given: exactly that many.  Not given: five, and more while --min-seconds is not reached")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="keep repeating the timed job (beyond --reps, on further disjoint seeds, at "
                    "most --max-reps times) until the repetitions add up to this much timed work: a 20-step job is 25 ms, and five of "
                    "them are neither a stable median nor visible to a once-per-second utilisation sampler")
    ap.add_argument("--max-reps", type=int, default=48)
    ap.add_argument("--batch", type=int, default=1024, help="sub-problems per GPU per step (cfg2: 1024)")
    ap.add_argument("--window", type=int, default=6144, help="sub-problems in flight per GPU (a multiple of 3072 = three sweep wavefronts per SIMD)")
    ap.add_argument("--gather-chunk", type=int, default=2048, help="N > 1: items per chunk of the overlapped all-gather")
    ap.add_argument("--gather-path", action="store_true", help="N = 1: run the job through the N > 1 path (results written into "
                    "ResultBuffers, chunks 'gathered' from the progress callback on a side stream; one rank: a copy) -- diagnostic")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-all", action="store_true", help="bracket every kernel class with events in the timed run "
                    "(per-kernel breakdown; costs ~4 %% of throughput in dispatch gaps) instead of the Riccati sweep only")
    ap.add_argument("--cpu-sample", type=int, default=16384)
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (cfg3 / cfg4 / cfg5 after the headline; N = 1 only)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit(f"--gpus {args.gpus}: need at least one GPU")
    if args.gather_path and args.gpus != 1:
        raise SystemExit("--gather-path is the N = 1 diagnostic of the N > 1 path; with --gpus > 1 that path runs anyway")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks, one per GPU, BEFORE anything here
        # has imported torch or touched a GPU (a process that has initialised HIP must never be replaced or forked)
        raise SystemExit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the rank count is not the one asked for")
    # DPILQR_BENCH_BACKEND=gloo (diagnostic): the N > 1 code path with gloo instead of RCCL, every rank on whatever GPU its
    # LOCAL_RANK maps to modulo the visible ones -- how the multi-rank path is exercised on a one-GPU box (tests/test_gpu_api.py)
    backend = os.environ.get("DPILQR_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()            # counting devices does not initialise HIP
    if backend == "nccl" and world > 1 and n_dev < world:
        raise SystemExit(f"bench.py: --gpus {world} needs {world} visible GPUs, one per rank over RCCL; this node shows {n_dev} "
                         "(DPILQR_BENCH_BACKEND=gloo runs the multi-rank path with several ranks per GPU: a diagnostic, not a measurement)")
    local_rank = local_rank % max(n_dev, 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import dpilqr_amd
    from dpilqr_amd import _lib
    from dpilqr_amd.sharding import ResultBuffers
    _lib.require_gpu()

    B, reps = args.batch, max(1, args.reps if args.reps is not None else 5)
    max_reps = max(reps, args.max_reps) if (args.reps is None and args.min_seconds > 0) else reps
    Q, R, Qf = np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4)

    def make_job(n_steps, seed0, host=False):
        """n_steps batches of B scenarios each, resident in HBM; seeds are disjoint across ranks, repetitions and steps.  The
        scenarios are generated on the device (dpilqr_random_setup: bit for bit np.random.seed(s); random_setup(...), checked
        against the reference's own outputs in tests/test_gpu_api.py); host copies (rank 0, first job) feed the CPU baseline."""
        from dpilqr_amd.util import random_setup_batch
        x0, xf = random_setup_batch((seed0, n_steps * B), K_AGENTS, N_S, var=K_AGENTS / 2, n_d=2, energy=10.0)
        pb = dpilqr_amd.ProblemBatch([0] * K_AGENTS, [2] * K_AGENTS, xf, Q, R, Qf, 0.5, 0.1, T)
        U0 = torch.zeros((n_steps * B, T, N_U), dtype=torch.float64, device="cuda")
        return dict(pb=pb, x0=x0, U0=U0, x0_h=x0.cpu().numpy() if host else None, xf_h=xf.cpu().numpy() if host else None)

    seeds_per_rank = (max_reps * args.steps + args.warmup) * B
    seed_base = rank * seeds_per_rank                                    # weak scaling: every rank its own seeds
    # repetition r's job is generated just before it runs, outside the clock (only repetition 0's stays: the parity and roofline
    # legs look at it)
    job_of = lambda r: make_job(args.steps, seed_base + r * args.steps * B, host=(r == 0 and rank == 0))
    jobs = [job_of(0)]
    warm = make_job(args.warmup, seed_base + max_reps * args.steps * B) if args.warmup > 0 else None
    # N > 1: the results' home and the gathered results of all ranks, allocated once for the job's shape
    rb = (ResultBuffers(args.steps * B, T, N_X, N_U, chunk=args.gather_chunk, device=torch.device("cuda", local_rank), timeline=True)
          if (world > 1 or args.gather_path) else None)
    gather_marks = {}                   # the last repetition's: host time and a stream event at the end of the solve, finish()'s own time

    def run(j, gather):
        if gather is not None:
            gather.begin()
            r = j["pb"].solve(j["x0"], j["U0"], n_lqr_iter=50, tol=1e-3, window=args.window, out=gather.out, progress=gather.progress)
            ev = torch.cuda.Event(enable_timing=True); ev.record()
            gather_marks.update(t_solve_end=time.perf_counter(), ev_solve_end=ev)
            gather.finish()             # the one collective of the path: what the solve's progress reports did not already send
            gather_marks["finish_issue_s"] = time.perf_counter() - gather_marks["t_solve_end"]
            return r
        return j["pb"].solve(j["x0"], j["U0"], n_lqr_iter=50, tol=1e-3, window=args.window)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if warm is not None:
        run(warm, None)                 # W untimed steps: every kernel variant loaded, workspace pool filled
    if rb is not None:
        rb.warm()                       # the collective on the timed job's own buffers and message sizes, outside the clock
    _lib.profile_enable(True, classes=None if args.profile_all else ["riccati"]); _lib.profile_read(reset=True)
    for w in (12, 8, 4):
        _lib.profile_read_sweep(w, reset=True)
    times = []
    own_times = []                      # this rank's own clock around the same region (per-rank values of the JSON line)
    r0 = None
    rep = 0
    while True:
        job = jobs[0] if rep == 0 else job_of(rep)
        fence()
        t0 = time.perf_counter()
        r = run(job, rb)
        fence()
        own_times.append(time.perf_counter() - t0)
        if rb is not None:              # (after the fence: every event has completed)
            tl = rb.timeline_relative_to(gather_marks["t_solve_end"], gather_marks["ev_solve_end"])
            gather_marks["timeline"] = {
                "chunks": len(tl), "chunk_items": args.gather_chunk,
                "issued_before_solve_end": sum(1 for _, dt_, _ in tl if dt_ <= 0.0),
                "issue_s_rel_solve_end": [round(dt_, 6) for _, dt_, _ in tl],
                "done_ms_rel_solve_end": [None if ms_ is None else round(ms_, 3) for _, _, ms_ in tl],
                "finish_issue_s": gather_marks["finish_issue_s"], "region_s": own_times[-1],
                "exposed_tail_ms": max([0.0] + [ms_ for _, _, ms_ in tl if ms_ is not None]),
                "note": "this rank's last repetition: per chunk of the one all-gather, when it was issued (host clock) and when it completed "
                        "(side-stream event) relative to the end of the solve -- negative: overlapped with the solve, positive: exposed"}
        dt = torch.tensor([own_times[-1]], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        times.append(float(dt.item()))
        if rep == 0:
            r0 = {k: v.clone() for k, v in r.items()} if rb is not None else r      # the parity leg looks at repetition 0
            if rb is not None:      # untimed sanity of the collective: this rank's block of the gathered results is what it solved,
                g = rb.results()    # and every rank's block is a finished job (status 1..3 everywhere)
                assert torch.equal(g["X"][rank], r["X"]) and torch.equal(g["J"][rank], r["J"]), "all-gather returned another block"
                assert bool(((g["status"] >= 1) & (g["status"] <= 3)).all()), "a rank's gathered results are not finished solves"
        del r
        if rep > 0:
            del job
        rep += 1
        # `times` holds the MAX over ranks: every rank takes the same decision
        if rep >= max_reps or (rep >= reps and sum(times) >= args.min_seconds):
            break
    reps = rep
    prof = _lib.profile_read(reset=True)
    sweep_variants = {w: _lib.profile_read_sweep(w, reset=True) for w in (12, 8, 4)}
    _lib.profile_enable(False)

    order = sorted(range(reps), key=lambda i: times[i])
    elapsed = times[order[(reps - 1) // 2]] if reps % 2 else 0.5 * (times[order[reps // 2 - 1]] + times[order[reps // 2]])
    total_units = B * world * args.steps
    value = total_units / elapsed

    # Who ran this: every rank's device as the runtime names it, and its own median throughput.  A multi-GPU line validates
    # itself -- the rank count is the process group's, the devices must be pairwise distinct (two ranks on one GPU would
    # still print a plausible number), and the N = 1 line says which code path produced it.
    props = torch.cuda.get_device_properties(local_rank)
    own_sorted = sorted(own_times)
    own_med = own_sorted[(reps - 1) // 2] if reps % 2 else 0.5 * (own_sorted[reps // 2 - 1] + own_sorted[reps // 2])
    me = {"rank": rank, "local_rank": local_rank, "device_index": torch.cuda.current_device(), "device_name": props.name,
          "pci_bus": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", -1) & 0xFF,
                                         getattr(props, "pci_device_id", -1) & 0xFF),
          "uuid": str(getattr(props, "uuid", "")), "pid": os.getpid(),
          "value": args.steps * B / own_med, "ms_per_step": own_med / args.steps * 1e3}
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    else:
        ranks = [me]
    ranks_seen = dist.get_world_size() if world > 1 else 1
    if rank == 0:
        if ranks_seen != args.gpus or len(ranks) != args.gpus or sorted(r_["rank"] for r_ in ranks) != list(range(args.gpus)):
            raise SystemExit(f"bench.py: the process group has {ranks_seen} ranks, --gpus asked for {args.gpus}")
        if backend == "nccl" and world > 1:
            ident = [(r_["uuid"] or r_["pci_bus"], r_["device_index"]) for r_ in ranks]
            if len(set(ident)) != world:
                raise SystemExit(f"bench.py: {world} ranks but their devices are not pairwise distinct: {ident}")

    if rank == 0:
        # roofline = the dominant kernel: the sweep variant (wavefronts per workgroup) that ran the full-window launches,
        # exactly the launches a rocprofv3 kernel trace lists under that instantiation's name; the job's other sweep
        # launches (the draining tail's smaller variants) are reported beside it under "all_sweep_launches"
        fused = not _lib.route_flag("DPILQR_NO_FUSED")
        all_ric = prof["riccati"]
        waves = next((w for w in (12, 8, 4) if sweep_variants[w]["launches"]), None)
        ric = sweep_variants[waves] if waves else all_ric
        kernel_name = (f"k_riccati_mfma<{N_X},{N_U},{waves},4,2,{'true' if fused else 'false'}>" if waves
                       else "riccati sweep (all variants)")
        pass_bytes = FUSED_BYTES if fused else BWD_READ_BYTES + BWD_WRITE_BYTES
        sec = ric["ms"] * 1e-3
        gbs = ric["items"] * pass_bytes / sec / 1e9 if sec > 0 else 0.0
        tflops = ric["items"] * BWD_USEFUL_FLOPS / sec / 1e12 if sec > 0 else 0.0
        tflops_dense = ric["items"] * BWD_FLOPS / sec / 1e12 if sec > 0 else 0.0
        all_sec = all_ric["ms"] * 1e-3
        all_gbs = all_ric["items"] * pass_bytes / all_sec / 1e9 if all_sec > 0 else 0.0
        all_tflops = all_ric["items"] * BWD_USEFUL_FLOPS / all_sec / 1e12 if all_sec > 0 else 0.0
        traffic, traffic_note = None, None
        tf = ROOT / "profiles" / "riccati_traffic.json"   # PMC pass (rocprofv3 --pmc), see profiles/README.md
        if tf.exists() and ric["launches"]:   # measured HBM bytes per sub-problem pass x the items of an average launch
            tj = json.loads(tf.read_text())
            key = "hbm_bytes_per_subproblem_pass_fused" if fused else "hbm_bytes_per_subproblem_pass"
            per_pass = tj.get(key)
            if tj.get("kernel_source_sha16") == kernel_source_sha16():
                traffic = per_pass * ric["items"] / ric["launches"] if per_pass else None
                traffic_note = f"PMC pass on this source ({tj.get('kernel_source_sha16')}), {tj.get('fused', {}).get('source', tj.get('source'))}"
            else:
                traffic_note = (f"profiles/riccati_traffic.json was measured on another version of the sweep "
                                f"({tj.get('kernel_source_sha16')} != {kernel_source_sha16()}): re-run scripts/profile_round.sh")
        launches = max(ric["launches"], 1)
        if fused:
            roofline = {"bound": "fp64", "kernel": kernel_name, "achieved": tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": tflops / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                        "flops_per_subproblem_pass": BWD_USEFUL_FLOPS, "flops_per_launch": ric["items"] * BWD_USEFUL_FLOPS / launches,
                        "dense_count": {"flops_per_subproblem_pass": BWD_FLOPS, "achieved": tflops_dense, "frac": tflops_dense / FP64_PEAK_TFLOPS,
                                        "note": "SURVEY 8(d)'s dense flop count, which also prices the multiplications by the structural zeros of "
                                                "block-diagonal [A|B] that the kernel skips: an algorithmic figure, not pipe utilisation"},
                        "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                "bytes_per_subproblem_pass": pass_bytes},
                        "note": "fused sweep: flops on data (block-diagonal [A|B], the reference's association; bench.py "
                                "useful_flops_per_step) over the kernel's launch time against the fp64 pipe (vector ALU and fp64 MFMA "
                                "share it); its HBM traffic is the fused byte count"}
        else:
            roofline = {"bound": "hbm", "kernel": kernel_name, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note, "bytes_per_subproblem_pass": pass_bytes,
                        "algorithmic_bytes_per_launch": ric["items"] * pass_bytes / launches}
        roofline.update({"launches": ric["launches"], "subproblem_passes": ric["items"], "avg_launch_ms": ric["ms"] / launches,
                         "all_sweep_launches": {"launches": all_ric["launches"], "subproblem_passes": all_ric["items"],
                                                "avg_launch_ms": all_ric["ms"] / max(all_ric["launches"], 1),
                                                "achieved": all_tflops if fused else all_gbs,
                                                "frac": all_tflops / FP64_PEAK_TFLOPS if fused else all_gbs / HBM_PEAK_GBS}})
        # the tiles-through-HBM form of the same sweep (the plugin boundary, and the kernel the north star's "40 % of the HBM
        # roofline" refers to): one full window of this job's final iterates, records made once, the sweep timed alone
        j0 = jobs[0]
        nw = min(args.window, r0["X"].shape[0])
        pbw = dpilqr_amd.ProblemBatch([0] * K_AGENTS, [2] * K_AGENTS, j0["pb"]._xf[:nw], Q, R, Qf, 0.5, 0.1, T)
        tiles_w = pbw.make_tiles(r0["X"][:nw], r0["U"][:nw])
        mu_w = torch.full((nw,), 0.125, dtype=torch.float64, device="cuda")
        sweep_reps = 8
        for _ in range(3):
            dpilqr_amd.backward_pass_tiles(tiles_w, nw, T, N_X, N_U, mu_w, blocks=(N_S, N_C))
        # steady state: the launches back to back between ONE pair of events, the way the solve loop issues them (a launch that
        # follows a device synchronisation starts on an idle, down-clocked chip: 1.03 ms against 0.90 ms per 6144 items)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(sweep_reps):
            dpilqr_amd.backward_pass_tiles(tiles_w, nw, T, N_X, N_U, mu_w, blocks=(N_S, N_C))
        e1.record(); torch.cuda.synchronize()
        ms_t = e0.elapsed_time(e1) / sweep_reps
        ms_list = []                      # ... and each launch on its own after a synchronisation (rounds 1-3 reported the median of these)
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            dpilqr_amd.backward_pass_tiles(tiles_w, nw, T, N_X, N_U, mu_w, blocks=(N_S, N_C))
            e1.record(); torch.cuda.synchronize()
            ms_list.append(e0.elapsed_time(e1))
        gbs_t = nw * (BWD_READ_BYTES + BWD_WRITE_BYTES) / (ms_t * 1e-3) / 1e9
        gbs_r = nw * BWD_READ_BYTES / (ms_t * 1e-3) / 1e9
        tiled = {"bound": "hbm", "kernel": f"k_riccati_mfma<{N_X},{N_U},{12 if nw > 2048 else (8 if nw > 1024 else 4)},4,2,false>",
                 "achieved": gbs_t, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs_t / HBM_PEAK_GBS,
                 "read_only": {"achieved": gbs_r, "frac": gbs_r / HBM_PEAK_GBS, "bytes_per_subproblem_pass": BWD_READ_BYTES,
                               "note": "the north star's '>= 40 % HBM-read roofline' counts the tile reads alone"},
                 "items": nw, "launch_ms": ms_t, "launches_back_to_back": sweep_reps,
                 "method_version": 2,      # 1 (rounds 1-3): launch_ms / achieved / frac = median of ISOLATED launches (idle chip);
                                           # 2 (round 4 on): mean of the back-to-back launches; version 1's figure is isolated_launch_ms
                 "isolated": {"launch_ms": float(np.median(ms_list)),
                              "frac": nw * (BWD_READ_BYTES + BWD_WRITE_BYTES) / (float(np.median(ms_list)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "read_only_frac": nw * BWD_READ_BYTES / (float(np.median(ms_list)) * 1e-3) / 1e9 / HBM_PEAK_GBS},
                 "isolated_launch_ms": {"median": float(np.median(ms_list)), "min_max": [min(ms_list), max(ms_list)],
                                        "note": "one launch at a time, each after a device synchronisation (idle chip)"},
                 "bytes_per_subproblem_pass": BWD_READ_BYTES + BWD_WRITE_BYTES,
                 "note": f"record-fed sweep (dpilqr_backward_pass_tiles_blocks) on one window of this job's final iterates, timed "
                         f"alone after the timed region: {sweep_reps} launches back to back between one pair of HIP events"}
        del tiles_w
        # BASELINE configs[1] read literally: ONE batch of 1024 sub-problems solved on its own (latency of a single batch)
        pb1 = dpilqr_amd.ProblemBatch([0] * K_AGENTS, [2] * K_AGENTS, j0["pb"]._xf[:B], Q, R, Qf, 0.5, 0.1, T)
        x1, U1 = j0["x0"][:B], j0["U0"][:B]
        pb1.solve(x1, U1, window=B)
        t1 = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r1 = pb1.solve(x1, U1, window=B)
            torch.cuda.synchronize(); t1.append(time.perf_counter() - t0)
        single = {"items": B, "ms": float(np.median(t1)) * 1e3, "ms_min_max": [min(t1) * 1e3, max(t1) * 1e3],
                  "subproblems_per_s": B / float(np.median(t1)), "global_iterations": int(r1["n_bwd"].max().item()),
                  "note": "one 1024-item batch solved alone (median of 5): bounded by the batch's longest item, "
                          "max(n_bwd) dependent iterations of one sweep + one line search each"}
        nb = r0["n_bwd"].cpu().numpy(); nf = r0["n_fwd"].cpu().numpy(); st = r0["status"].cpu().numpy()
        out = {
            "metric": "ilqr_subproblems_per_sec", "value": value, "unit": "subproblems/s", "n_gpus": world,
            "ranks_seen": ranks_seen, "backend": (backend if world > 1 else None),
            "ranks": ranks,          # per rank: device index / PCI bus / uuid as the runtime reports them, its own value
            "n1_path": ("plain" if rb is None else "gather-path (diagnostic)") if world == 1 else None,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "repetitions": {"n": reps, "statistic": "median", "ms_per_step": [t / args.steps * 1e3 for t in times],
                            "value_min_max": [total_units / max(times), total_units / min(times)]},
            "config": {"workload": f"cfg2: batches of {B} independent 5-agent DoubleIntDynamics4D iLQR sub-problems per GPU "
                                   "per step, T=50, scripts/analysis.py scenario distribution, tol=1e-3, n_lqr_iter=50; "
                                   f"{args.steps} steps = {args.steps * B} distinct seeds per GPU, window of {args.window} in flight; "
                                   f"the job is repeated {reps} times on disjoint seeds and the median is reported",
                       "batch_per_gpu": B, "window": args.window, "n_x": N_X, "n_u": N_U, "horizon": T,
                       "mean_backward_passes": float(nb.mean()), "mean_forward_passes": float(nf.mean()),
                       "converged_frac": float((st == 1).mean()), "linesearch_failed_frac": float((st == 2).mean()),
                       "parallelism": (f"batch-sharded x{world}, one all-gather in chunks of {args.gather_chunk} items overlapped with the solve"
                                       if world > 1 else "single GPU")},
            "gather_timeline": gather_marks.get("timeline"),
            "roofline": roofline, "roofline_tiles_through_hbm": tiled, "single_batch_1024": single,
            "kernel_ms_per_step": {k: v["ms"] / (args.steps * reps) for k, v in prof.items() if v["launches"]},
        }
        if world == 1 and not args.no_cpu_baseline:
            x0_h, xf_h = j0["x0_h"], j0["xf_h"]
            ns = min(args.cpu_sample, x0_h.shape[0])
            out["cpu_baseline"], o = cpu_baseline(x0_h, xf_h, ns)
            out["cpu_baseline_numpy"] = cpu_baseline_numpy(x0_h, xf_h, o)
            # The oracle doubles as a live parity check of this very run (untimed): every item of a sub-sample is held to the
            # ensemble envelope of oracle/parity.py -- an untimed GPU solve of the same items with the decision trace
            # switched on (bit-identical to the timed one: scheduling does not change an item's arithmetic), the oracle
            # replayed along the GPU's decisions from x0 and from eight perturbed copies of x0 -- nothing is exempt
            from oracle import oracle as orc, parity
            nc = min(ns, 2048)
            pbs = dpilqr_amd.ProblemBatch([0] * K_AGENTS, [2] * K_AGENTS, xf_h[:nc], Q, R, Qf, 0.5, 0.1, T)
            g = {k_: v.cpu().numpy() for k_, v in pbs.solve(x0_h[:nc], np.zeros((nc, T, N_U)), trace=True, window=args.window).items()}
            assert np.array_equal(g["X"], r0["X"][:nc].cpu().numpy()), "traced re-solve differs from the timed solve"
            proto = orc.Problem([0] * K_AGENTS, [2] * K_AGENTS, xf_h[0], Q, R, Qf, 0.5, 0.1, T)
            on = {k_: v[:nc] for k_, v in o.items()}
            rep = parity.envelope(g, proto, x0_h[:nc], xf_h[:nc], np.zeros((nc, T, N_U)), natural=on)
            Xg = r0["X"][:ns].cpu().numpy()
            err = np.abs(Xg - o["X"]).reshape(ns, -1).max(axis=1) / np.maximum(np.abs(o["X"]).reshape(ns, -1).max(axis=1), 1e-300)
            same_trace = (nb[:ns] == o["n_bwd"]) & (nf[:ns] == o["n_fwd"]) & (st[:ns] == o["status"])
            out["parity_vs_oracle"] = {
                "items": int(ns), "identical_decision_trace_frac": float(same_trace.mean()),
                "states_within_1e-5_frac": float((err < 1e-5).mean()),
                "all_items_envelope": dict(rep["summary"], violating_items=[int(i) for i in np.where(~rep["ok"])[0][:8]],
                                           reasons=[w for w in rep["why"] if w][:4]),
                "note": "all_items_envelope: every item of the first 2048, through every iteration of its solve, against the oracle "
                        "replayed along the GPU's own decisions from x0 and from 32 perturbed copies of x0 (+-1e-14..5e-13): accepted costs, "
                        "final X, U, J within 10 x the ensemble's spread, every decision that is not the oracle's own verdict on the "
                        "same iterate undetermined in the ensemble too (oracle/parity.py; calibrated in tests/test_parity_envelope.py)"}
        if world == 1 and not args.no_configs:
            # BASELINE's other configurations, untimed by the headline (the literal configs[1] batch is `single_batch_1024` above)
            torch.cuda.empty_cache()
            cpu_legs = not args.no_cpu_baseline
            out["configs"] = {"cfg2_single_batch": dict(single, workload="configs[1] read literally: one batch of 1024 sub-problems alone"),
                              "cfg3": scenario_config("cfg3", cpu=cpu_legs), "cfg4": scenario_config("cfg4", cpu=cpu_legs),
                              "cfg5": cfg5_config(cpu=cpu_legs)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
