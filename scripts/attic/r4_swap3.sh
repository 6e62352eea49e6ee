#!/bin/bash
out=gpurun_out/r4s; mkdir -p $out
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
timeout 1500 python -m pytest tests -q -x -m gpu > $out/pytest_all.log 2>&1
tail -3 $out/pytest_all.log
timeout 900 python scripts/envelope_tail.py 2>&1 | q > $out/envelope_tail.txt
cat $out/envelope_tail.txt
