"""Micro-benchmark of the Riccati sweep alone on cfg2-shaped tiles: time vs batch size."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import dpilqr_amd as dp
from dpilqr_amd import _lib
from dpilqr_amd.device import empty, ptr, stream_handle, to_dev
from bench import scenarios, K_AGENTS, T, N_U, N_X, BWD_READ_BYTES, BWD_WRITE_BYTES

sizes = [int(a) for a in sys.argv[1:]] or [64, 256, 512, 768, 1024, 1536, 2048, 4096]
Bmax = max(sizes)
x0, xf = scenarios(0, Bmax)
pb = dp.ProblemBatch([0] * K_AGENTS, [2] * K_AGENTS, xf, np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4), 0.5, 0.1, T)
import os
if os.environ.get("REALISTIC"):
    r = pb.solve(x0, np.zeros((Bmax, T, N_U)), n_lqr_iter=int(os.environ["REALISTIC"]))   # operating point after a few iterations
    X, U = r["X"], r["U"]
    mu = to_dev(np.full(Bmax, 0.125))
else:
    X, J = pb.rollout(x0, np.zeros((Bmax, T, N_U)))
    U = torch.zeros((Bmax, T, N_U), dtype=torch.float64, device="cuda")
    mu = to_dev(np.ones(Bmax))
tiles = pb.make_tiles(X, U)
K = empty((Bmax, T, N_U, N_X)); d = empty((Bmax, T, N_U))
lib = _lib.load()
BLK = (0, 0) if os.environ.get('DENSE') else (4, 2)
for B in sizes:
    for rep in range(3):
        _lib.check(lib.dpilqr_backward_pass_tiles_blocks(B, T, N_X, N_U, BLK[0], BLK[1], ptr(tiles), ptr(mu), ptr(K), ptr(d), None, None, None, stream_handle()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for rep in range(n):
        _lib.check(lib.dpilqr_backward_pass_tiles_blocks(B, T, N_X, N_U, BLK[0], BLK[1], ptr(tiles), ptr(mu), ptr(K), ptr(d), None, None, None, stream_handle()))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    gbs = B * (BWD_READ_BYTES + BWD_WRITE_BYTES) / (us * 1e-6) / 1e9
    print(f"B={B:5d}  {us:8.1f} us/launch  {gbs:8.1f} GB/s algorithmic  ({gbs/80:.1f} % of 8 TB/s)")

# K2 timed inside the K1 -> K2 sequence of the solver (tiles freshly written by the producer)
B = int(os.environ.get("B2", "1024"))
Xb, Ub = X[:B].contiguous(), U[:B].contiguous()
pb2 = dp.ProblemBatch([0] * K_AGENTS, [2] * K_AGENTS, xf[:B], np.diag([1.0, 1, 0, 0]), np.eye(2), 1000.0 * np.eye(4), 0.5, 0.1, T)
tl = pb2.tiles_buffer()
tot = 0.0; n = 20
for rep in range(n + 3):
    pb2.make_tiles(Xb, Ub, tl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.dpilqr_backward_pass_tiles_blocks(B, T, N_X, N_U, BLK[0], BLK[1], ptr(tl), ptr(mu), ptr(K), ptr(d), None, None, None, stream_handle()))
    e1.record(); torch.cuda.synchronize()
    if rep >= 3:
        tot += e0.elapsed_time(e1)
print(f"K2 right after K1 (B=1024): {tot / n * 1e3:.1f} us/launch")
tot = 0.0
for rep in range(n + 3):
    pb2.make_tiles(Xb, Ub, tl)
    torch.cuda.synchronize()
    import time; time.sleep(0.002)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.dpilqr_backward_pass_tiles_blocks(B, T, N_X, N_U, BLK[0], BLK[1], ptr(tl), ptr(mu), ptr(K), ptr(d), None, None, None, stream_handle()))
    e1.record(); torch.cuda.synchronize()
    if rep >= 3:
        tot += e0.elapsed_time(e1)
print(f"K2 2 ms after K1 (B=1024): {tot / n * 1e3:.1f} us/launch")

# K1 -> K2 back to back, no host sync inside the loop, K2 bracketed by in-stream events
evs = []
for rep in range(n + 3):
    pb2.make_tiles(Xb, Ub, tl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.dpilqr_backward_pass_tiles_blocks(B, T, N_X, N_U, BLK[0], BLK[1], ptr(tl), ptr(mu), ptr(K), ptr(d), None, None, None, stream_handle()))
    e1.record()
    evs.append((e0, e1))
torch.cuda.synchronize()
ts = [a.elapsed_time(b) * 1e3 for a, b in evs[3:]]
print(f"K1->K2 stream-ordered, no host sync: K2 {np.mean(ts):.1f} us/launch (min {min(ts):.1f})")
# K2 twice after one K1: is the second K2 (same tiles, no producer in between) faster?
evs = []
for rep in range(n + 3):
    pb2.make_tiles(Xb, Ub, tl)
    pair = []
    for k2 in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(lib.dpilqr_backward_pass_tiles_blocks(B, T, N_X, N_U, BLK[0], BLK[1], ptr(tl), ptr(mu), ptr(K), ptr(d), None, None, None, stream_handle()))
        e1.record()
        pair.append((e0, e1))
    evs.append(pair)
torch.cuda.synchronize()
first = np.mean([p[0][0].elapsed_time(p[0][1]) for p in evs[3:]]) * 1e3
second = np.mean([p[1][0].elapsed_time(p[1][1]) for p in evs[3:]]) * 1e3
print(f"K1 -> K2 -> K2: first K2 {first:.1f} us, second K2 {second:.1f} us")

# K1 -> (stream 1 GiB through the caches) -> K2 : does evicting K1's freshly written lines help?
junk = torch.empty(1 << 27, dtype=torch.float64, device="cuda"); junk2 = torch.empty_like(junk)
evs = []
for rep in range(n + 3):
    pb2.make_tiles(Xb, Ub, tl)
    junk2.copy_(junk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.dpilqr_backward_pass_tiles_blocks(B, T, N_X, N_U, BLK[0], BLK[1], ptr(tl), ptr(mu), ptr(K), ptr(d), None, None, None, stream_handle()))
    e1.record()
    evs.append((e0, e1))
torch.cuda.synchronize()
ts = [a.elapsed_time(b) * 1e3 for a, b in evs[3:]]
print(f"K1 -> 2 GiB copy -> K2: K2 {np.mean(ts):.1f} us/launch")
# no K1 at all, but the 2 GiB copy before every K2 (tiles cold in HBM, not freshly written)
evs = []
for rep in range(n + 3):
    junk2.copy_(junk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.dpilqr_backward_pass_tiles_blocks(B, T, N_X, N_U, BLK[0], BLK[1], ptr(tl), ptr(mu), ptr(K), ptr(d), None, None, None, stream_handle()))
    e1.record()
    evs.append((e0, e1))
torch.cuda.synchronize()
ts = [a.elapsed_time(b) * 1e3 for a, b in evs[3:]]
print(f"2 GiB copy -> K2 (tiles cold, written long ago): K2 {np.mean(ts):.1f} us/launch")

# K1 writes a DIFFERENT buffer, K2 reads the old one: global-state effect (clocks) or data-placement effect?
tl2 = pb2.tiles_buffer()
evs = []
for rep in range(n + 3):
    pb2.make_tiles(Xb, Ub, tl2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.dpilqr_backward_pass_tiles_blocks(B, T, N_X, N_U, BLK[0], BLK[1], ptr(tl), ptr(mu), ptr(K), ptr(d), None, None, None, stream_handle()))
    e1.record()
    evs.append((e0, e1))
torch.cuda.synchronize()
ts = [a.elapsed_time(b) * 1e3 for a, b in evs[3:]]
print(f"K1(other buffer) -> K2(old buffer): K2 {np.mean(ts):.1f} us/launch")
