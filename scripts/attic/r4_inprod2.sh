#!/bin/bash
out=gpurun_out/r4i; mkdir -p $out
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
timeout 1500 python -m pytest tests -q -x -m gpu > $out/pytest_all.log 2>&1
tail -5 $out/pytest_all.log
rm -f $out/mc.txt
for rep in 1 2; do
for off in 0 1; do
  if [ $off == 1 ]; then export DPILQR_NO_INPROD=1; else unset DPILQR_NO_INPROD; fi
  echo "DPILQR_NO_INPROD=$off" >> $out/mc.txt
  timeout 600 python scripts/montecarlo.py cfg4 8192 2>&1 | q | tail -4 >> $out/mc.txt
done; done
cat $out/mc.txt
