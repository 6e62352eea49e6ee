"""Why is the CPU baseline slow on the GPU box?  Prints what the process may use (cores, affinity, cgroup quota)
and times oracle/liboracle.so on cfg2 items at several thread counts.  Diagnostic only (round-2 verdict, weak item 10)."""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def read(p):
    try:
        return Path(p).read_text().strip()
    except OSError as e:
        return f"<{e.__class__.__name__}>"


def main():
    from bench import scenarios, K_AGENTS, N_U, T
    from oracle import oracle as orc
    info = {"os.cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)),
            "cgroup cpu.max": read("/sys/fs/cgroup/cpu.max"), "cgroup cpu.stat": read("/sys/fs/cgroup/cpu.stat"),
            "cfs_quota_us": read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), "cfs_period_us": read("/sys/fs/cgroup/cpu/cpu.cfs_period_us"),
            "loadavg": read("/proc/loadavg"), "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS"),
            "model": [l for l in read("/proc/cpuinfo").splitlines() if l.startswith("model name")][:1]}
    print(json.dumps(info, indent=1), flush=True)
    n = 4096
    x0, xf = scenarios(0, n)
    Q, R, Qf = np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4)
    proto = orc.Problem([0] * K_AGENTS, [2] * K_AGENTS, xf[0], Q, R, Qf, 0.5, 0.1, T)
    rows = []
    for th in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        if th > 2 * (os.cpu_count() or 1):
            break
        m = min(n, max(64, 48 * th))
        orc.solve_batch(proto, x0[:th], xf[:th], np.zeros((th, T, N_U)), n_threads=th)
        t0 = time.perf_counter(); c0 = time.process_time()
        orc.solve_batch(proto, x0[:m], xf[:m], np.zeros((m, T, N_U)), n_threads=th)
        dt = time.perf_counter() - t0; cpu = time.process_time() - c0
        rows.append(dict(threads=th, items=m, wall_s=round(dt, 3), cpu_s=round(cpu, 3), per_s=round(m / dt, 1),
                         per_s_per_thread=round(m / dt / th, 2), cpu_over_wall=round(cpu / dt, 1)))
        print(json.dumps(rows[-1]), flush=True)
    print("cgroup cpu.stat after:", read("/sys/fs/cgroup/cpu.stat"))


if __name__ == "__main__":
    main()
