#!/bin/bash
out=gpurun_out/r4e; mkdir -p $out
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "bit_identical or solve_cfg2 or solve_misc or window_invariance or enqueue or all_items" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
{
for rep in 1 2; do
for t in 0 1; do
  if [ $t == 1 ]; then export DPILQR_NO_TEAM=1; else unset DPILQR_NO_TEAM; fi
  echo "DPILQR_NO_TEAM=$t"
  timeout 300 python scripts/sweep_waves_ab.py 64 256 512 1024 2>&1 | q
done; done
for t in 0 1; do
  if [ $t == 1 ]; then export DPILQR_NO_TEAM=1; else unset DPILQR_NO_TEAM; fi
  echo "DPILQR_NO_TEAM=$t"
  timeout 600 python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), 'sub/s', 'single_batch ms', round(d['single_batch_1024']['ms'],3), 'tiled ms', round(d['roofline_tiles_through_hbm']['launch_ms'],4), 'read_only frac', round(d['roofline_tiles_through_hbm']['read_only']['frac'],4))"
done
} > $out/team.txt 2>&1
cat $out/team.txt
