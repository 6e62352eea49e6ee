"""K1 -> K2 -> K2 sequence for rocprofv3 --pmc (cycle counts of the first vs second sweep)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import dpilqr_amd as dp
from dpilqr_amd import _lib
from dpilqr_amd.device import empty, ptr, stream_handle, to_dev
from bench import scenarios, K_AGENTS, T, N_U, N_X
B = 1024
x0, xf = scenarios(0, B)
pb = dp.ProblemBatch([0] * K_AGENTS, [2] * K_AGENTS, xf, np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4), 0.5, 0.1, T)
X, J = pb.rollout(x0, np.zeros((B, T, N_U))); U = torch.zeros((B, T, N_U), dtype=torch.float64, device="cuda")
mu = to_dev(np.ones(B)); K = empty((B, T, N_U, N_X)); d = empty((B, T, N_U)); tl = pb.tiles_buffer()
lib = _lib.load()
for rep in range(6):
    pb.make_tiles(X, U, tl)
    for k2 in range(2):
        _lib.check(lib.dpilqr_backward_pass_tiles(B, T, N_X, N_U, ptr(tl), ptr(mu), ptr(K), ptr(d), None, None, None, stream_handle()))
torch.cuda.synchronize()
