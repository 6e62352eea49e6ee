#!/bin/bash
# round 4: in-sweep production for the four-state clusters the fused forms do not serve (six agents; unhinted smaller ones)
# against their previous routes (DPILQR_NO_INPROD4=1), one gpurun call
out=gpurun_out/r4j; mkdir -p $out
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -k "in_sweep or fused or cfg3 or golden or solve_misc or window" > $out/pytest.log 2>&1
tail -3 $out/pytest.log
rm -f $out/ip4.txt
for rep in 1 2; do
for off in 0 1; do
  if [ $off == 1 ]; then export DPILQR_NO_INPROD4=1; else unset DPILQR_NO_INPROD4; fi
  echo "== DPILQR_NO_INPROD4=$off" >> $out/ip4.txt
  timeout 300 python scripts/bench_wg.py --model uni4 6 2>&1 | q | cut -c1-130 >> $out/ip4.txt
  timeout 300 python scripts/solve_breakdown.py --model uni4 6 2>&1 | q | cut -c1-250 >> $out/ip4.txt
  timeout 600 python scripts/montecarlo.py cfg3 4096 2>&1 | q | grep "first call\|second call" | cut -c1-120 >> $out/ip4.txt
done; done
cat $out/ip4.txt
