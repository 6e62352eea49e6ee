"""One batched cfg2 solve (B sub-problems) for rocprofv3: python3 scripts/prof_solve.py [B] [reps]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import dpilqr_amd
from bench import scenarios, K_AGENTS, T, N_U

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
x0, xf = scenarios(0, B)
pb = dpilqr_amd.ProblemBatch([0] * K_AGENTS, [2] * K_AGENTS, xf, np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4), 0.5, 0.1, T)
for _ in range(reps):
    r = pb.solve(x0, np.zeros((B, T, N_U)))
torch.cuda.synchronize()
print("n_bwd mean", r["n_bwd"].double().mean().item())
