#!/usr/bin/env python
"""Instruction census per phase of a kernel's assembly listing (hipcc -S -DDPILQR_PHASE_MARKS): lines, exec-mask regions,
branches, LDS operations, full LDS waits, MFMAs, fp64 VALU, selects, cross-lane moves between the '; ==== end of phase i' marks
of the horizon loop.
    python scripts/isa_census.py listing.s [loop-header-label]"""
import re
import sys

L = open(sys.argv[1]).read().split("\n")
marks = [i for i, l in enumerate(L) if "==== end of phase" in l]
hdr_label = sys.argv[2] if len(sys.argv) > 2 else None
if hdr_label:
    hdr = next(i for i, l in enumerate(L) if l.startswith(hdr_label + ":"))
else:       # the loop header: the last label before the first mark that follows a mark of the highest phase
    hdr = max(i for i, l in enumerate(L[:marks[1]]) if re.match(r"\.LBB\d+_\d+:.*Loop Header", l))
end = [i for i, l in enumerate(L) if "s_endpgm" in l][-1]
bounds = [hdr] + [i for i in marks if i > hdr] + [end]
PAT = [("saveexec", "saveexec"), ("branch", "s_cbranch"), ("lds", "ds_read|ds_write"), ("lgkm0", r"lgkmcnt\(0\)"),
       ("vmcnt", "vmcnt"), ("mfma", "v_mfma"), ("fp64", r"v_(fma|mul|add|fmac)_f64"), ("cndmask", "v_cndmask"),
       ("readlane", "v_readlane|v_readfirstlane"), ("dpp", "dpp"), ("scratch", "scratch_")]
print("phase".ljust(8) + "lines".rjust(7) + "".join(n.rjust(9) for n, _ in PAT))
for k in range(len(bounds) - 1):
    seg = [l for l in L[bounds[k]:bounds[k + 1]] if l.startswith("\t") and not l.startswith("\t;") and not l.startswith("\t.")]
    name = f"S{k}" if k < len(bounds) - 2 else f"S{k}+"
    print(name.ljust(8) + str(len(seg)).rjust(7) + "".join(str(sum(1 for l in seg if re.search(p, l))).rjust(9) for _, p in PAT))
print("(the last row also holds the out-of-line blocks the compiler placed behind the loop)")
