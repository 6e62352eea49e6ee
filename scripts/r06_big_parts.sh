#!/bin/bash
# Round 6: config 5, the backward pass with 5 .. 32 parts per team (DPILQR_BIG_TEAM_PARTS; round 5's rule -- as many as give every
# wavefront at most one tile pair -- is nine), one item and eight
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
export DPILQR_DEBUG_ROUTES=1
for p in ${PARTS:-5 7 9 11 13 17 25 32}; do
  echo -n "parts $p: "; DPILQR_BIG_TEAM_PARTS=$p python3 scripts/bench_big.py ${ITEMS:-1} 2>&1 | grep "backward" | cut -c1-40 | tr '\n' ' '; echo
done
