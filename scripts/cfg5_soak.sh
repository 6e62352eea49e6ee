#!/bin/bash
# config 5's passes and whole solves N times, each in a FRESH process (the large-cluster sweep's round-4 failure -- an HSA
# aperture violation at 82 spilled registers -- depended on the machine's state, not on the data): bash scripts/cfg5_soak.sh [N] [lib]
n=${1:-20}; lib=$2
[ -n "$lib" ] && export DPILQR_DEBUG_ROUTES=1 DPILQR_LIB=$lib
fail=0
for i in $(seq 1 $n); do
  python -m pytest tests/test_gpu_big.py -q -m gpu -x -k "cfg5 or hetero or forced_big" -p no:cacheprovider > /tmp/soak_$i.log 2>&1 || { fail=$((fail+1)); tail -5 /tmp/soak_$i.log; }
done
echo "cfg5 soak: $n fresh processes, $fail failed (library: ${lib:-default}); last: $(tail -1 /tmp/soak_$n.log)"
