#!/bin/bash
# round 4: row swaps inside the blocked Gauss-Jordan (riccati_wg.hpp) against declining the step (variant noswap), one gpurun call
out=gpurun_out/r4s; mkdir -p $out
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
V=$PWD/dpilqr_amd/variants/libdpilqr_hip_noswap.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -k "blocked or six_state or sweep_multi or fused or cfg3 or cfg4 or golden" > $out/pytest.log 2>&1
tail -5 $out/pytest.log
rm -f $out/swap.txt
for rep in 1 2; do
for t in 0 1; do
  if [ $t == 1 ]; then export DPILQR_LIB=$V; else unset DPILQR_LIB; fi
  echo "== noswap=$t" >> $out/swap.txt
  timeout 300 python scripts/solve_breakdown.py --model uni4 9 12 15 2>&1 | q >> $out/swap.txt
  timeout 300 python scripts/solve_breakdown.py --model quad6 5 8 10 2>&1 | q >> $out/swap.txt
done; done
for t in 0 1 0 1; do
  if [ $t == 1 ]; then export DPILQR_LIB=$V; else unset DPILQR_LIB; fi
  echo "== noswap=$t" >> $out/swap.txt
  timeout 600 python scripts/montecarlo.py cfg4 8192 2>&1 | q | grep "first call\|second call" | cut -c1-120 >> $out/swap.txt
  timeout 600 python scripts/montecarlo.py cfg3 4096 2>&1 | q | grep "first call\|second call" | cut -c1-120 >> $out/swap.txt
done
cat $out/swap.txt | cut -c1-230
