#!/bin/bash
V=$PWD/dpilqr_amd/variants
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
for lib in $V/libdpilqr_hip_presc.so ""; do echo "DPILQR_LIB=$lib  (presc = sin, cos of the three angles by separate calls)"; DPILQR_LIB=$lib python scripts/bench_q12.py 1 2 3 5 2>&1 | q | cut -c1-330; done
python scripts/bench_big.py 256 2>&1 | q | tail -4
python -m pytest tests -m gpu -x -q -k "model or twelve or hetero or quad12 or cfg5 or big or passes" 2>&1 | tail -3
