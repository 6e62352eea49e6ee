#!/bin/bash
V=$PWD/dpilqr_amd/variants
for lib in "" $V/libdpilqr_hip_skipb.so "" $V/libdpilqr_hip_skipb.so; do echo "DPILQR_LIB=$lib"; DPILQR_LIB=$lib python scripts/sweep_waves_ab.py 6144 2048 2>&1 | grep -v "Warning\|x\[mask\]\|amdgpu.ids"; done
