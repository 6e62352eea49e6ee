import sys, numpy as np, torch
sys.path.insert(0, '.')
import dpilqr_amd as dp
from dpilqr_amd.device import to_dev, empty, ptr, stream_handle
from dpilqr_amd import _lib
from bench import scenarios, K_AGENTS, T, N_U, N_X
B = 6144
x0, xf = scenarios(0, B)
Q, R, Qf = np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4)
pb = dp.ProblemBatch([0] * K_AGENTS, [2] * K_AGENTS, xf, Q, R, Qf, 0.5, 0.1, T)
import os
if os.path.exists("/tmp/s0_xu.npz"):          # a variant build timed on the default build's iterates
    z = np.load("/tmp/s0_xu.npz"); X, U = to_dev(z["X"]), to_dev(z["U"])
else:
    r = pb.solve(x0, np.zeros((B, T, N_U)), n_lqr_iter=4)
    X, U = r["X"], r["U"]
    np.savez("/tmp/s0_xu.npz", X=X.cpu().numpy(), U=U.cpu().numpy())
mu = to_dev(np.full(B, 0.125))
def timeit(p):
    K = empty((B, T, N_U, N_X)); d = empty((B, T, N_U))
    lib = _lib.load()
    for _ in range(3):
        _lib.check(lib.dpilqr_backward_pass_fused(p._d, ptr(X), ptr(U), ptr(mu), ptr(K), ptr(d), None, stream_handle()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        _lib.check(lib.dpilqr_backward_pass_fused(p._d, ptr(X), ptr(U), ptr(mu), ptr(K), ptr(d), None, stream_handle()))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
print("radius 0.5 :", timeit(pb), "us per 6144-item fused sweep")
pb0 = dp.ProblemBatch([0] * K_AGENTS, [2] * K_AGENTS, xf, Q, R, Qf, 1e-9, 0.1, T)
print("radius 1e-9:", timeit(pb0), "us (no pair ever near: the plugin stage is [l_x | l_u] only)")
# fraction of steps with a near pair
Xh = X.cpu().numpy().reshape(B, T + 1, 5, 4)[:, :, :, :2]
dd = np.linalg.norm(Xh[:, :, :, None] - Xh[:, :, None, :], axis=-1) + np.eye(5) * 9
print("fraction of (item, step) with a pair within the radius:", float((dd.min(axis=(2, 3)) <= 0.5).mean()))
