#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
export DPILQR_DEBUG_ROUTES=1
echo "== release fences"; python3 scripts/bench_big.py 1 2>&1 | grep backward
export DPILQR_BIG_TEAM_RELAXED=1
echo "== relaxed arrivals (timing only)"; python3 scripts/bench_big.py 1 2>&1 | grep backward
