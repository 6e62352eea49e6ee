#!/bin/bash
# Round 6: the team's hand-overs under repetition -- the team tests of tests/test_gpu_big.py (bit-identity to the single workgroup at
# 1 .. 32 items in both types, late and stalled helpers, uneven load, full XCDs with both kinds of release) N times, each in a fresh
# process, then scripts/big_team_check.py (1 .. 128 items) M times.  A race in a hand-over shows as a DIFFERENT row or a failed test.
n=${1:-12}; m=${2:-4}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
fail=0
for i in $(seq 1 $n); do
  timeout 300 python -m pytest tests/test_gpu_big.py -q -m gpu -x -k "team" -p no:cacheprovider > /tmp/tsoak_$i.log 2>&1 || { fail=$((fail+1)); tail -5 /tmp/tsoak_$i.log; }
done
echo "team tests: $n fresh processes, $fail failed; last: $(tail -1 /tmp/tsoak_$n.log)"
bad=0
for i in $(seq 1 $m); do
  timeout 600 python3 scripts/big_team_check.py 1 8 24 32 64 128 2>&1 | grep -v amdgpu > /tmp/tcheck_$i.log
  bad=$((bad + $(grep -c DIFFERENT /tmp/tcheck_$i.log)))
done
echo "big_team_check: $m runs of 12 rows, $bad rows DIFFERENT"
