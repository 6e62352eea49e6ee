"""Where do the sweep's wavefronts run and when do they start? K1 -> K2 -> K2 with in-kernel stamps."""
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
bufs = [torch.zeros((B, 4), dtype=torch.int64, device="cuda") for _ in range(2)]
def k2(buf):
    _lib.check(lib.dpilqr_debug_stamps(ptr(buf)))
    _lib.check(lib.dpilqr_backward_pass_tiles(B, T, N_X, N_U, ptr(tl), ptr(mu), ptr(K), ptr(d), None, None, None, stream_handle()))
for rep in range(4):
    pb.make_tiles(X, U, tl); k2(bufs[0]); torch.cuda.synchronize(); k2(bufs[1]); torch.cuda.synchronize()
for name, buf in zip(["first K2 after K1", "second K2"], bufs):
    s = buf.cpu().numpy()
    t0 = s[:, 0].min()
    start = (s[:, 0] - t0) / 100.0; end = (s[:, 1] - t0) / 100.0   # us
    hw = s[:, 2]; xcc = s[:, 3] & 0xF
    simd = (hw >> 4) & 0x3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
    cuid = xcc * 1000 + se * 100 + sh * 16 + cu
    per_cu = np.bincount(np.unique(cuid, return_inverse=True)[1])
    per_simd = np.bincount(np.unique(cuid * 4 + simd, return_inverse=True)[1])
    print(f"{name}: wave start us: p50 {np.percentile(start,50):.1f} p90 {np.percentile(start,90):.1f} max {start.max():.1f}; "
          f"wave life us: mean {(end-start).mean():.1f} max {(end-start).max():.1f}; kernel span {end.max():.1f}")
    print(f"   CUs used {len(per_cu)}, WGs per CU hist {np.bincount(per_cu)}, waves per SIMD hist {np.bincount(per_simd)}, per XCC {np.bincount(xcc)}")
