#!/bin/bash
# round 4: the blocked Gauss-Jordan's threshold rule decided once per panel (speculate without swaps, redo the panel with them)
# against a ballot per pivot (variant perpivot: build_variant.sh perpivot tu_riccati -DDPILQR_GJ_CHECK_PER_PIVOT), one gpurun call
out=gpurun_out/r4s; mkdir -p $out
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
V=$PWD/dpilqr_amd/variants/libdpilqr_hip_perpivot.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -k "blocked or six_state or sweep_multi or fused or cfg3 or cfg4 or golden" > $out/pytest4.log 2>&1
tail -3 $out/pytest4.log
rm -f $out/swap4.txt
for rep in 1 2; do
for t in 0 1; do
  if [ $t == 1 ]; then export DPILQR_LIB=$V; else unset DPILQR_LIB; fi
  echo "== perpivot=$t" >> $out/swap4.txt
  timeout 300 python scripts/bench_wg.py --model uni4 12 15 2>&1 | q | cut -c1-130 >> $out/swap4.txt
  timeout 300 python scripts/bench_wg.py --model quad6 5 8 10 2>&1 | q | cut -c1-130 >> $out/swap4.txt
done; done
cat $out/swap4.txt
