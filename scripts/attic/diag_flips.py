"""Diagnostic: where do GPU and oracle decision traces part, and how close was the call?"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import dpilqr_amd as dp
from dpilqr_amd.util import random_setup
from oracle import oracle as orc
from tests.golden_util import cfg2_params

c = cfg2_params(); B = int(sys.argv[1]) if len(sys.argv) > 1 else 256; base = 1000
x0 = np.zeros((B, 20)); xf = np.zeros((B, 20))
for s in range(B):
    np.random.seed(base + s)
    a, b = random_setup(5, 4, is_rotation=False, rel_dist=5, var=2.5, n_d=2, random=True, energy=10.0)
    x0[s], xf[s] = a.ravel(), b.ravel()
pb = dp.ProblemBatch(c["model"], c["n_dims"], xf, c["Q"], c["R"], c["Qf"], c["radius"], c["dt"], c["T"])
r = pb.solve(x0, np.zeros((B, 50, 10)), trace=True)
tr = r["trace"].cpu().numpy(); nb = r["n_bwd"].cpu().numpy()
nbad = 0
for i in range(B):
    p = orc.Problem(c["model"], c["n_dims"], xf[i], c["Q"], c["R"], c["Qf"], c["radius"], c["dt"], c["T"])
    o = p.solve(x0[i], np.zeros((50, 10)))
    ot = o["trace"]; gt = tr[i][:nb[i]]
    n = min(len(ot), len(gt))
    same = len(ot) == len(gt) and np.array_equal(ot[:, 1], gt[:, 1])
    if not same:
        nbad += 1
        first = next((j for j in range(n) if ot[j, 1] != gt[j, 1]), n)
        print(f"item {i}: oracle n_bwd {len(ot)} gpu {len(gt)} first divergence at iter {first}")
        for j in range(max(0, first - 1), min(n, first + 1)):
            print("   it", j, "oracle acc/J/J*", ot[j, 1], repr(ot[j, 2]), repr(ot[j, 3]), "| gpu", gt[j, 1], repr(gt[j, 2]), repr(gt[j, 3]))
        if first == n and first > 0:
            j = first - 1
            print("   last common it", j, "oracle", ot[j, 1:4], "gpu", gt[j, 1:4], "rel dJ oracle", abs(ot[j-1,3]-ot[j,2])/abs(ot[j-1,3]) if j>0 else None)
print("differing:", nbad, "of", B)
