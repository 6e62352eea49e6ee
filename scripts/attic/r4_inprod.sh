#!/bin/bash
# round 4: in-sweep production (riccati_mfma.hpp, PNS) against the record-fed routes (DPILQR_NO_INPROD=1), one gpurun call
out=gpurun_out/r4i; mkdir -p $out
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "in_sweep or six_state or three_state or padded or test_passes or solve_misc or golden" > $out/pytest.log 2>&1
tail -5 $out/pytest.log
rm -f $out/small.txt
for off in 0 1; do
  if [ $off == 1 ]; then export DPILQR_NO_INPROD=1; else unset DPILQR_NO_INPROD; fi
  echo "DPILQR_NO_INPROD=$off" >> $out/small.txt
  timeout 300 python scripts/bench_wg.py --model quad6 1 2 3 4 2>&1 | q >> $out/small.txt
  timeout 300 python scripts/bench_wg.py --model car3 1 2 3 4 5 6 2>&1 | q >> $out/small.txt
  timeout 300 python scripts/solve_breakdown.py --model quad6 1 2 3 4 2>&1 | q >> $out/small.txt
done
cat $out/small.txt
