#!/bin/bash
out=gpurun_out/r4l; mkdir -p $out
rm -f $out/lsteam2.txt
for rep in 1 2 3; do
for t in 1024 512 256 0; do
  if [ $t == 0 ]; then export DPILQR_LS_NO_TEAM=1; unset DPILQR_LS_TEAM_MAX; else unset DPILQR_LS_NO_TEAM; export DPILQR_LS_TEAM_MAX=$t; fi
  echo -n "team max $t: " >> $out/lsteam2.txt
  timeout 600 python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), 'sub/s', 'single_batch ms', round(d['single_batch_1024']['ms'],3))" >> $out/lsteam2.txt
done; done
cat $out/lsteam2.txt
