#!/bin/bash
# round 4: the four-state family's S1 / S2 by the structure of its blocks in the fused workgroup sweep, against the general chains
# (variant nostruct: build_variant.sh nostruct tu_riccati -DDPILQR_WG_NO_STRUCT4), one gpurun call
out=gpurun_out/r4t; mkdir -p $out
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
V=$PWD/dpilqr_amd/variants/libdpilqr_hip_nostruct.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -k "fused or blocked or sweep_multi or cfg3 or golden or window" > $out/pytest.log 2>&1
tail -3 $out/pytest.log
rm -f $out/struct.txt
for rep in 1 2; do
for t in 0 1; do
  if [ $t == 1 ]; then export DPILQR_LIB=$V; else unset DPILQR_LIB; fi
  echo "== nostruct=$t" >> $out/struct.txt
  timeout 300 python scripts/bench_wg.py --model uni4 6 9 12 15 2>&1 | q | cut -c1-130 >> $out/struct.txt
done; done
for t in 0 1 0 1; do
  if [ $t == 1 ]; then export DPILQR_LIB=$V; else unset DPILQR_LIB; fi
  echo "== nostruct=$t" >> $out/struct.txt
  timeout 600 python scripts/montecarlo.py cfg3 4096 2>&1 | q | grep "first call\|second call" | cut -c1-120 >> $out/struct.txt
done
cat $out/struct.txt
