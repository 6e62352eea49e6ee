"""Diagnostic: per-phase shader cycles of the tiled sweep (needs a -DDPILQR_PHASE_STAMPS build in /tmp)."""
import ctypes as C, sys, os
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import subprocess, torch
so = "/tmp/libdpilqr_stamps.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off",
                "-DDPILQR_PHASE_STAMPS", f"-I{ROOT/'include'}", f"-I{ROOT/'dpilqr_amd'/'csrc'}", "-o", so,
                str(ROOT/"dpilqr_amd"/"csrc"/"dpilqr_hip.hip")], check=True)
from dpilqr_amd import _lib
_lib.LIB_PATH = Path(so)
import dpilqr_amd as dp
from dpilqr_amd.device import empty, ptr, stream_handle, to_dev
from bench import scenarios, K_AGENTS, T, N_U, N_X
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x0, xf = scenarios(0, B)
pb = dp.ProblemBatch([0] * K_AGENTS, [2] * K_AGENTS, xf, np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4), 0.5, 0.1, T)
X, J = pb.rollout(x0, np.zeros((B, T, N_U))); U = torch.zeros((B, T, N_U), dtype=torch.float64, device="cuda")
mu = to_dev(np.ones(B)); K = empty((B, T, N_U, N_X)); d = empty((B, T, N_U)); tl = pb.make_tiles(X, U)
lib = _lib.load()
buf = torch.zeros((B * 12,), dtype=torch.int64, device="cuda")
_lib.check(lib.dpilqr_debug_stamps(ptr(buf)))
for rep in range(3):
    _lib.check(lib.dpilqr_backward_pass_tiles_blocks(B, T, N_X, N_U, 4, 2, ptr(tl), ptr(mu), ptr(K), ptr(d), None, None, None, stream_handle()))
torch.cuda.synchronize()
s = buf.cpu().numpy()
ph = s[4 * B:].reshape(B, 8)[:, :7] / T
names = ["S0 park AB", "S1 [A|B]^T[P|p]", "S2 [T1;T2][A|B]", "S3 LU solve+store", "S4 T3", "S5+S6 sums, symmetrise", "-"]
tot = ph.sum(1).mean()
for n, v in zip(names, ph.mean(0)):
    print(f"{n:22s} {v:8.0f} cycles/step  {100 * v / tot:5.1f} %")
print(f"total {tot:.0f} cycles/step (stamps serialise LDS at phase ends, so the sum is an upper bound)")
