#!/bin/bash
# Round 6: config 5's backward pass with the blocked LU (default) against the column form (variant lucol) and the gains compared.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$R/gpurun_out/r06_big; mkdir -p $out
cd $R
for arm in new lucol; do
  if [ $arm != new ]; then export DPILQR_DEBUG_ROUTES=1 DPILQR_LIB=$R/dpilqr_amd/variants/libdpilqr_hip_$arm.so; else unset DPILQR_LIB DPILQR_DEBUG_ROUTES; fi
  python3 scripts/bench_big.py 1 32 256 > $out/bench_big_$arm.txt 2>&1
  python3 scripts/big_pass_dump.py $out/pass_$arm > $out/dump_$arm.txt 2>&1
done
python3 - <<PY
import numpy as np
for f in ("K","d"):
    a=np.load("$out/pass_new_"+f+".npy"); b=np.load("$out/pass_lucol_"+f+".npy")
    print(f, "identical" if np.array_equal(a,b) else "max rel diff %.3e" % (np.max(np.abs(a-b))/np.max(np.abs(b))))
PY
rm -f $out/pass_*.npy
tail -n 8 $out/bench_big_*.txt
