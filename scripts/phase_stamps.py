"""Diagnostic: per-phase shader cycles of the wavefront sweep (record-fed or fused), from in-kernel s_memtime stamps.
Needs a -DDPILQR_PHASE_STAMPS build of the library: `python scripts/phase_stamps.py --build` makes dpilqr_amd/variants/libdpilqr_stamps.so
(no GPU needed); then, on the GPU box, `python scripts/phase_stamps.py [B] [--fused]`.  `--wg B k [quad6|uni4]`: the workgroup sweep;
`--s3split` (at --build and at run time): its S3 split into blocked elimination / fall-back / barrier + store."""
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
so = ROOT / "dpilqr_amd" / "variants" / "libdpilqr_stamps.so"     # variants/ travels to the GPU box, build/ does not
if "--build" in sys.argv:
    so.parent.mkdir(exist_ok=True)
    (ROOT / "build").mkdir(exist_ok=True)
    csrc = ROOT / "dpilqr_amd" / "csrc"
    objs = []
    procs = []
    for src in sorted(csrc.glob("*.hip")):
        obj = ROOT / "build" / f"stamps_{src.stem}.o"
        objs.append(str(obj))
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                                       "-DDPILQR_PHASE_STAMPS", *(["-DDPILQR_S3_SPLIT"] if "--s3split" in sys.argv else []),
                                       *__import__("__graft_entry__").UNIT_FLAGS.get(src.stem, []),
                                       f"-I{ROOT / 'include'}", f"-I{csrc}", "-c", "-o", str(obj), str(src)]))
    assert all(p.wait() == 0 for p in procs)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(so), *objs], check=True)
    sys.exit(0)
import torch  # noqa: E402
from dpilqr_amd import _lib  # noqa: E402
_lib.LIB_PATH = so
import dpilqr_amd as dp  # noqa: E402
from dpilqr_amd.device import empty, ptr, stream_handle, to_dev  # noqa: E402
from bench import K_AGENTS, N_U, N_X, T, scenarios  # noqa: E402
args = [a for a in sys.argv[1:] if not a.startswith("--")]
fused = "--fused" in sys.argv
if "--wg" in sys.argv:      # the mid-size sweep (k_riccati_wg): python scripts/phase_stamps.py --wg B k [quad6|uni4]
    from dpilqr_amd.util import random_setup
    B, k = int(args[0]), int(args[1])
    mdl, ns, nc, nd, Tw = (3, 4, 2, 2, 100) if (len(args) > 2 and args[2] == "uni4") else (4, 6, 3, 3, 75)
    n, m = k * ns, k * nc
    x0 = np.zeros((B, n)); xf = np.zeros((B, n))
    for s_ in range(B):
        np.random.seed(500 + s_)
        a, b = random_setup(k, ns, is_rotation=False, rel_dist=k, var=k / 2, n_d=nd, random=True, energy=10.0)
        x0[s_], xf[s_] = a.ravel(), b.ravel()
    U0 = np.zeros((B, Tw, m))
    if mdl == 4:
        U0[:, :, 0::3] = 9.80665
    pb = dp.ProblemBatch([mdl] * k, [nd] * k, xf, (50.0 * np.eye(6)) if mdl == 4 else np.diag([1.0, 1, 0, 0]), np.eye(nc), 1000.0 * np.eye(ns), 0.5, 0.1, Tw)
    r = pb.solve(x0, U0, n_lqr_iter=2)
    tl = pb.make_tiles(r["X"], r["U"])
    mu = to_dev(np.full(B, 0.125)); K = empty((B, Tw, m, n)); d = empty((B, Tw, m))
    lib = _lib.load()
    buf = torch.zeros((B * 12,), dtype=torch.int64, device="cuda")
    _lib.check(lib.dpilqr_debug_stamps(ptr(buf)))
    for rep in range(2):
        if fused:
            _lib.check(lib.dpilqr_backward_pass_fused(pb._d, ptr(r["X"]), ptr(r["U"]), ptr(mu), ptr(K), ptr(d), None, stream_handle()))
        else:
            _lib.check(lib.dpilqr_backward_pass_tiles_blocks(B, Tw, n, m, ns, nc, ptr(tl), ptr(mu), ptr(K), ptr(d), None, None, None, stream_handle()))
    torch.cuda.synchronize()
    ph = buf.cpu().numpy()[4 * B:].reshape(B, 8) / Tw
    names = ["S0 stage AB, request l-values", "S1 [A|B]^T[P|p]", "S2 [T1;T2][A|B]", "S3 LU solve+store", "S4 T3", "S5+S6", "-"]
    if "--s3split" in sys.argv:   # a --build --s3split library: slots 6, 7 are times (wavefront 0's), slot 3 is what is left of S3 (barrier, store of K)
        print(f"S3 split: blocked elimination {ph[:, 6].mean():.0f} ticks/step, fall-back {ph[:, 7].mean():.0f}, barrier + store of [K|d] {ph[:, 3].mean():.0f}")
        ph[:, 3] += ph[:, 6] + ph[:, 7]
    else:
        print(f"S3: steps solved by blocks (m = 13..16 and m >= 22; else: by the search-free register elimination) {ph[:, 6].mean() / 10:.1f} %; steps in which the "
              f"fall-back's partial pivoting moved a row: {ph[:, 7].mean() / 10:.1f} %")
    ph = ph[:, :6]
    tot = ph.sum(1).mean()
    for nm_, v in zip(names, ph.mean(0)):
        print(f"{nm_:30s} {v:8.0f} ticks/step  {100 * v / tot:5.1f} %")
    print(f"k_riccati_wg{' (fused)' if fused else ''} n_x={n}: total {tot:.0f} ticks/step (s_memtime shader-clock ticks, ~2.4 GHz)")
    sys.exit(0)
B = int(args[0]) if args else 1024
x0, xf = scenarios(0, B)
pb = dp.ProblemBatch([0] * K_AGENTS, [2] * K_AGENTS, xf, np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4), 0.5, 0.1, T)
r = pb.solve(x0, np.zeros((B, T, N_U)), n_lqr_iter=2)       # an operating point where agents interact
X, U = r["X"], r["U"]
import os
if os.environ.get("RADIUS"):   # the same iterates, another proximity radius for the timed pass (1e-9: no pair is ever near)
    pb = dp.ProblemBatch([0] * K_AGENTS, [2] * K_AGENTS, xf, np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4), float(os.environ["RADIUS"]), 0.1, T)
mu = to_dev(np.full(B, 0.125)); K = empty((B, T, N_U, N_X)); d = empty((B, T, N_U)); tl = pb.make_tiles(X, U)
lib = _lib.load()
buf = torch.zeros((B * 12,), dtype=torch.int64, device="cuda")
_lib.check(lib.dpilqr_debug_stamps(ptr(buf)))
for rep in range(3):
    if fused:
        _lib.check(lib.dpilqr_backward_pass_fused(pb._d, ptr(X), ptr(U), ptr(mu), ptr(K), ptr(d), None, stream_handle()))
    else:
        _lib.check(lib.dpilqr_backward_pass_tiles_blocks(B, T, N_X, N_U, 4, 2, ptr(tl), ptr(mu), ptr(K), ptr(d), None, None, None, stream_handle()))
torch.cuda.synchronize()
s = buf.cpu().numpy()
ph = s[4 * B:].reshape(B, 8)[:, :7] / T
names = ["S0 park AB / pair derivatives", "S1 [A|B]^T[P|p]", "S2 [T1;T2][A|B]", "S3 LU solve+store", "S4 T3", "S5+S6 sums, symmetrise", "-"]
tot = ph.sum(1).mean()
for n, v in zip(names, ph.mean(0)):
    print(f"{n:30s} {v:8.0f} cycles/step  {100 * v / tot:5.1f} %")
print(f"total {tot:.0f} cycles/step (stamps serialise LDS at phase ends, so the sum is an upper bound)")
dur = (s[:4 * B].reshape(B, 4)[:, 1] - s[:4 * B].reshape(B, 4)[:, 0]) / 100.0
print(f"{'fused' if fused else 'record-fed'} sweep, {B} items: wavefront duration mean {dur.mean():.0f} us, max {dur.max():.0f} us")
