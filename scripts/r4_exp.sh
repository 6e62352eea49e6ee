#!/bin/bash
V=$PWD/dpilqr_amd/variants
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
python -m pytest tests -m gpu -x -q -k "model or passes or solve_misc or unicycle or uni or three_state or cfg3 or rhc or distributed or warmstart" 2>&1 | tail -3
for lib in "" $V/libdpilqr_hip_trigrec.so; do echo "DPILQR_LIB=$lib"; DPILQR_LIB=$lib python scripts/solve_breakdown.py --model uni4 3 5 9 15 2>&1 | q | cut -c1-330; done
echo "== the whole gpu suite on the recurrence build"
DPILQR_LIB=$V/libdpilqr_hip_trigrec.so python -m pytest tests -m gpu -q -k "not cfg4 and not cfg5 and not bench" 2>&1 | tail -12
