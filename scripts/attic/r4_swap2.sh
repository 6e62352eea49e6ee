#!/bin/bash
out=gpurun_out/r4s; mkdir -p $out
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
V=$PWD/dpilqr_amd/variants/libdpilqr_hip_noswap.so
rm -f $out/swap2.txt
for rep in 1 2; do
for t in 0 1; do
  if [ $t == 1 ]; then export DPILQR_LIB=$V; else unset DPILQR_LIB; fi
  echo "== noswap=$t" >> $out/swap2.txt
  timeout 300 python scripts/bench_wg.py --model uni4 12 15 2>&1 | q | cut -c1-130 >> $out/swap2.txt
  timeout 300 python scripts/bench_wg.py --model quad6 5 8 10 2>&1 | q | cut -c1-130 >> $out/swap2.txt
done; done
cat $out/swap2.txt
