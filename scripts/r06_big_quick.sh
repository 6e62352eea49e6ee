#!/bin/bash
# config 5, one item: backward-pass time (default library), phase clocks (variant bigstamps), gains against the column-form LU (variant lucol)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$R/gpurun_out/r06_big; mkdir -p $out
cd $R
python3 scripts/bench_big.py 1 8 32 256 > $out/bench_big_new.txt 2>&1
python3 scripts/big_pass_dump.py $out/pass_new > $out/dump_new.txt 2>&1
python3 scripts/big_pass_dump.py $out/pass32_new f32 > $out/dump32_new.txt 2>&1
export DPILQR_DEBUG_ROUTES=1 DPILQR_LIB=$R/dpilqr_amd/variants/libdpilqr_hip_lucol.so
python3 scripts/big_pass_dump.py $out/pass_lucol > $out/dump_lucol.txt 2>&1
python3 scripts/big_pass_dump.py $out/pass32_lucol f32 > $out/dump32_lucol.txt 2>&1
export DPILQR_LIB=$R/dpilqr_amd/variants/libdpilqr_hip_bigstamps.so
python3 scripts/bench_big.py 1 2>&1 | grep -E "phases|backward" | awk 'NR<=2 || /phases/' | sort | uniq -c | sort -rn | head -6 > $out/phases_new.txt
unset DPILQR_LIB DPILQR_DEBUG_ROUTES
python3 - <<PY
import numpy as np
for pre in ("pass","pass32"):
  for f in ("K","d"):
    a=np.load("$out/%s_new_%s.npy"%(pre,f)); b=np.load("$out/%s_lucol_%s.npy"%(pre,f))
    print(pre, f, "identical" if np.array_equal(a,b) else "max rel diff %.3e" % (np.max(np.abs(a-b))/np.max(np.abs(b))))
PY
rm -f $out/pass*.npy
cat $out/bench_big_new.txt | grep backward; cat $out/phases_new.txt | cut -c1-220
