#!/bin/bash
# round 4: one sub-normal guard per RK4 step (models with a heading / control trigonometry too) against one per sub-step
# (variant div6old: build_variant.sh div6old tu_forward -DDPILQR_DIV6_PER_SUBSTEP), one gpurun call
out=gpurun_out/r4d; mkdir -p $out
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
V=$PWD/dpilqr_amd/variants/libdpilqr_hip_div6old.so
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_api.py -q -x -k "golden or trace or model or rollout or heading or cfg3 or cfg4 or solve_misc or forward or passes" > $out/pytest.log 2>&1
tail -3 $out/pytest.log
rm -f $out/div6.txt
for rep in 1 2; do
for t in 0 1; do
  if [ $t == 1 ]; then export DPILQR_LIB=$V; else unset DPILQR_LIB; fi
  echo "== per-sub-step guard=$t" >> $out/div6.txt
  for m in "uni4 5" "uni4 15" "quad6 4" "quad6 10"; do timeout 300 python scripts/solve_breakdown.py --model $m 2>&1 | q | tail -1 | cut -c1-250 >> $out/div6.txt; done
done; done
cat $out/div6.txt
