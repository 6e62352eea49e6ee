#!/bin/bash
# A/B of the mid-size sweep: round-2 kernels (build/libdpilqr_hip_r02wg.so, if present) against the current library.
# gpurun -- 'bash scripts/ab_wg.sh'
out=gpurun_out/r03_wg; mkdir -p $out
for lib in build/libdpilqr_hip_r02wg.so dpilqr_amd/libdpilqr_hip.so; do
  [ -f $lib ] || continue
  tag=$(basename $lib .so)
  DPILQR_DEBUG_ROUTES=1 DPILQR_LIB=$PWD/$lib python scripts/bench_wg.py --model uni4 6 9 12 15 > $out/${tag}_uni4.txt 2>&1
  DPILQR_DEBUG_ROUTES=1 DPILQR_LIB=$PWD/$lib python scripts/bench_wg.py --model quad6 4 5 7 10 > $out/${tag}_quad6.txt 2>&1
done
tail -n 20 $out/*.txt
