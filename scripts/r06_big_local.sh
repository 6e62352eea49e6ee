#!/bin/bash
# Round 6: the team's hand-overs through the XCD's own L2 (default where the parts report one XCD) against agent-scope
# hand-overs (DPILQR_BIG_TEAM_AGENT=1): time, phase clocks, and the gains of the two compared bit for bit.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$R/gpurun_out/r06_local; mkdir -p $out
cd $R
export DPILQR_DEBUG_ROUTES=1
for arm in local agent; do
  if [ $arm == agent ]; then export DPILQR_BIG_TEAM_AGENT=1; else unset DPILQR_BIG_TEAM_AGENT; fi
  python3 scripts/bench_big.py 1 8 32 > $out/bench_big_$arm.txt 2>&1
  python3 scripts/big_pass_dump.py $out/pass_$arm > $out/dump_$arm.txt 2>&1
  python3 scripts/big_pass_dump.py $out/pass32_$arm f32 > $out/dump32_$arm.txt 2>&1
  DPILQR_LIB=$R/dpilqr_amd/variants/libdpilqr_hip_bigstamps.so python3 scripts/bench_big.py 1 2>&1 | grep -E "phases" | sort | uniq -c | sort -rn | head -3 > $out/phases_$arm.txt
done
unset DPILQR_BIG_TEAM_AGENT
python3 - <<PY
import numpy as np
for pre in ("pass","pass32"):
  for f in ("K","d"):
    a=np.load("$out/%s_local_%s.npy"%(pre,f)); b=np.load("$out/%s_agent_%s.npy"%(pre,f))
    print(pre, f, "identical" if np.array_equal(a,b) else "max rel diff %.3e" % (np.max(np.abs(a-b))/np.max(np.abs(b))))
PY
rm -f $out/pass*.npy
for arm in local agent; do echo "== $arm"; grep backward $out/bench_big_$arm.txt | cut -c1-60; cut -c1-220 $out/phases_$arm.txt; done
python3 scripts/big_team_check.py 1 8 32 64 > $out/team_check.txt 2>&1; tail -8 $out/team_check.txt
