#!/bin/bash
# timing experiment (wrong results in the variants): what S1 / S2 of the mid-size sweep cost in LAUNCH time
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R
for v in base wgskip1 wgskip2 wgskip3; do
  if [ $v == base ]; then unset DPILQR_LIB DPILQR_DEBUG_ROUTES; else export DPILQR_DEBUG_ROUTES=1 DPILQR_LIB=$R/dpilqr_amd/variants/libdpilqr_hip_$v.so; fi
  echo "== $v"
  python3 scripts/bench_wg.py --model uni4 15 2>&1 | grep "k=" | cut -c1-150
  python3 scripts/bench_wg.py --model quad6 10 6 2>&1 | grep "k=" | cut -c1-150
done
