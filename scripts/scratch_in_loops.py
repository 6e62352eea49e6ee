#!/usr/bin/env python
"""Spilled registers that are reloaded INSIDE loops: per kernel of an assembly listing (hipcc -S --cuda-device-only of a translation
unit), the scratch loads / stores that sit in a basic block the compiler marks as part of a loop.  A spill outside the loops
costs nothing; one inside a horizon loop is a memory round trip per step (k_linesearch_wave<4,10>: 24 of them, -30 % once gone).
    python scripts/scratch_in_loops.py listing.s [min_count]"""
import re
import subprocess
import sys

L = open(sys.argv[1]).read().split("\n")
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 1
kern, in_loop, depth_of = None, False, {}
stats = {}
for l in L:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        kern = m.group(1); in_loop = False
        stats[kern] = dict(loads=0, stores=0, loads_in_loop=0, stores_in_loop=0)
        continue
    if kern is None:
        continue
    if re.match(r"^\.LBB\d+_\d+:", l):
        in_loop = ("Loop" in l)          # "in Loop: Header=..." / "Loop Header" / "Parent Loop"
        continue
    if re.match(r"^; %bb\.\d+:", l):
        in_loop = ("Loop" in l)
        continue
    if "scratch_load" in l:
        stats[kern]["loads"] += 1; stats[kern]["loads_in_loop"] += in_loop
    elif "scratch_store" in l:
        stats[kern]["stores"] += 1; stats[kern]["stores_in_loop"] += in_loop
names = [k for k, v in stats.items() if v["loads_in_loop"] + v["stores_in_loop"] >= thr]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for k, d in sorted(zip(names, dem), key=lambda kd: -stats[kd[0]]["loads_in_loop"]):
    v = stats[k]
    print(f"{re.sub(r'^void dpilqr::', '', d.split('(')[0]):<60} scratch loads in loops {v['loads_in_loop']:4d} (of {v['loads']:4d}), stores in loops {v['stores_in_loop']:4d} (of {v['stores']:4d})")
