#!/usr/bin/env python
"""The reference's Monte-Carlo study (scripts/analysis.py) on the batched device path: centralized vs distributed
receding-horizon control over models x team sizes x trials, logged as the reference's CSV (dpilqr_amd/analysis.py).

    python scripts/monte_carlo_analysis.py                    analysis 2 (limit_solve_time, the reference's default main())
    python scripts/monte_carlo_analysis.py --unlimited        analysis 1
    ... --trials 256 --agents 3 4 5 --models DoubleIntDynamics4D --out logs/run.csv --seed 0
"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--unlimited", action="store_true", help="analysis 1: no t_kill, t_diverge = 4 N dt")
    ap.add_argument("--trials", type=int, default=2)
    ap.add_argument("--agents", type=int, nargs="+", default=[3, 4, 5, 6, 7])
    ap.add_argument("--models", nargs="+", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    from dpilqr_amd import analysis
    models = analysis.MODELS if a.models is None else tuple(m for m in analysis.MODELS if m.__name__ in a.models)
    t0 = time.perf_counter()
    res = analysis.monte_carlo_analysis(not a.unlimited, n_trials=a.trials, n_agents_iter=a.agents, models=models, seed0=a.seed,
                                        log_file=a.out)
    n = sum(len(v) for v in res.values())
    print(f"{n} trials x 2 branches in {time.perf_counter() - t0:.2f} s")


if __name__ == "__main__":
    main()
