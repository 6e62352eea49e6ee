"""Diagnostic: cycles the line-search kernel spends waiting for its per-step prefetch vs in total (needs a
-DDPILQR_PHASE_STAMPS build: python scripts/phase_stamps.py --build).
    python scripts/phase_stamps_fwd.py [B]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
so = ROOT / "build" / "libdpilqr_stamps.so"     # python scripts/phase_stamps.py --build
assert so.exists(), "build the stamps library first: python scripts/phase_stamps.py --build"
from dpilqr_amd import _lib
_lib.LIB_PATH = Path(so)
import dpilqr_amd as dp
from dpilqr_amd.device import ptr
from bench import scenarios, K_AGENTS, T, N_U, N_X
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
x0, xf = scenarios(0, B)
pb = dp.ProblemBatch([0] * K_AGENTS, [2] * K_AGENTS, xf, np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4), 0.5, 0.1, T)
lib = _lib.load()
buf = torch.zeros((B * 12,), dtype=torch.int64, device="cuda")
_lib.check(lib.dpilqr_debug_stamps(ptr(buf)))
r = pb.solve(x0, np.zeros((B, T, N_U)), n_lqr_iter=1, window=B)     # one iteration: K1, K2, K3 once over B items
torch.cuda.synchronize()
s = buf.cpu().numpy()[:2 * B].reshape(B, 2)
print(f"line search, {B} items in one launch: wait for prefetch {s[:,0].mean()/T:.0f} cycles/step of {s[:,1].mean()/T:.0f} cycles/step in the horizon loop")
