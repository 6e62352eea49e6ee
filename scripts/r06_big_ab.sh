#!/bin/bash
# Round 6: config 5's backward pass with the blocked LU (default: panels eight columns wide where a panel's rows fit one per lane)
# against the all-four-wide form (variant panel4) and the column form (variant lucol): times, and the gains compared bit for bit
# in fp64 and fp32.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$R/gpurun_out/r06_big; mkdir -p $out
cd $R
for arm in new panel4 lucol; do
  if [ $arm != new ]; then export DPILQR_DEBUG_ROUTES=1 DPILQR_LIB=$R/dpilqr_amd/variants/libdpilqr_hip_$arm.so; else unset DPILQR_LIB DPILQR_DEBUG_ROUTES; fi
  python3 scripts/bench_big.py 1 32 256 2>&1 | grep backward | cut -c1-40 > $out/bench_big_$arm.txt
  python3 scripts/big_pass_dump.py $out/pass_$arm > $out/dump_$arm.txt 2>&1
  python3 scripts/big_pass_dump.py $out/pass32_$arm f32 > $out/dump32_$arm.txt 2>&1
done
unset DPILQR_LIB DPILQR_DEBUG_ROUTES
python3 - <<PY
import numpy as np
for pre in ("pass", "pass32"):
    for other in ("panel4", "lucol"):
        for f in ("K", "d"):
            a = np.load("$out/%s_new_%s.npy" % (pre, f)); b = np.load("$out/%s_%s_%s.npy" % (pre, other, f))
            print(pre, "default vs", other, f, "identical" if np.array_equal(a, b) else "max rel diff %.3e" % (np.max(np.abs(a - b)) / np.max(np.abs(b))))
PY
rm -f $out/pass*.npy
for arm in new panel4 lucol; do echo "== $arm"; cat $out/bench_big_$arm.txt; done
