#!/bin/bash
# round 4: in-sweep production for one twelve-state agent against producer + padded record-fed sweep (DPILQR_NO_INPROD12=1);
# and the per-model-case Jacobian stores for the other families (register counts down): per-size passes again
out=gpurun_out/r4k; mkdir -p $out
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_big.py -q -x -k "in_sweep or twelve or six_state or three_state or padded or golden or solve_misc or warmstart" > $out/pytest.log 2>&1
tail -3 $out/pytest.log
rm -f $out/ip12.txt
for rep in 1 2; do
for off in 0 1; do
  if [ $off == 1 ]; then export DPILQR_NO_INPROD12=1; else unset DPILQR_NO_INPROD12; fi
  echo "== DPILQR_NO_INPROD12=$off" >> $out/ip12.txt
  timeout 300 python scripts/bench_wg.py --model quad12 --B 512 1 2>&1 | q | cut -c1-130 >> $out/ip12.txt
  timeout 300 python scripts/bench_wg.py --model quad12 --B 2048 1 2>&1 | q | cut -c1-130 >> $out/ip12.txt
  timeout 300 python scripts/bench_q12.py 1 2>&1 | q | tail -2 | cut -c1-200 >> $out/ip12.txt
done; done
timeout 300 python scripts/bench_wg.py --model quad6 1 2 3 4 2>&1 | q | cut -c1-130 >> $out/ip12.txt
timeout 300 python scripts/bench_wg.py --model car3 2 4 6 2>&1 | q | cut -c1-130 >> $out/ip12.txt
cat $out/ip12.txt
