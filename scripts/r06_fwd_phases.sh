#!/bin/bash
# config 5's forward pass (ten candidates, then the bare rollout): thread 0's clocks per part of a step (variant fwdstamps)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; mkdir -p gpurun_out
DPILQR_DEBUG_ROUTES=1 DPILQR_LIB=$R/dpilqr_amd/variants/libdpilqr_hip_fwdstamps.so python3 scripts/bench_big.py 1 2>&1 | grep -E "horizon_pass|forward" | sort | uniq -c | sort -rn | head -12 | cut -c1-260 > gpurun_out/r06_fwd_phases.txt
cat gpurun_out/r06_fwd_phases.txt
