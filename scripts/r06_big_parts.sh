#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
export DPILQR_DEBUG_ROUTES=1 DPILQR_LIB=$R/dpilqr_amd/variants/libdpilqr_hip_bigstamps.so
python3 scripts/bench_big.py 1 2>&1 | grep -E "phases" | sort | uniq -c | sort -rn | head -3
unset DPILQR_LIB DPILQR_DEBUG_ROUTES
./scripts/ubench/s5_shapes
