#!/bin/bash
# round 4: the line search with two wavefronts per item (rollout / costs) for launches of at most 1024 items, against one
# wavefront per item (DPILQR_LS_NO_TEAM=1), one gpurun call
out=gpurun_out/r4l; mkdir -p $out
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_configs.py -q -x -k "team or bit_identical or solve_cfg2 or window_invariance or enqueue or all_items or golden or trace or progress or cfg3 or cfg4 or solve_misc" > $out/pytest.log 2>&1
tail -3 $out/pytest.log
rm -f $out/lsteam.txt
for rep in 1 2; do
for t in 0 1; do
  if [ $t == 1 ]; then export DPILQR_LS_NO_TEAM=1; else unset DPILQR_LS_NO_TEAM; fi
  echo "== DPILQR_LS_NO_TEAM=$t" >> $out/lsteam.txt
  timeout 300 python scripts/bench_ls.py --iters 8 --B 1024 2>&1 | q >> $out/lsteam.txt
  timeout 300 python scripts/bench_ls.py --iters 8 --B 256 2>&1 | q >> $out/lsteam.txt
  timeout 600 python bench.py --steps 20 --no-cpu-baseline --profile-all 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), 'sub/s', 'single_batch ms', round(d['single_batch_1024']['ms'],3), {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})" >> $out/lsteam.txt
  for m in "uni4 5" "uni4 3" "quad6 2" "quad6 4" "quad6 5"; do timeout 300 python scripts/solve_breakdown.py --model $m 2>&1 | q | tail -1 | cut -c1-250 >> $out/lsteam.txt; done
  timeout 600 python scripts/montecarlo.py cfg4 8192 2>&1 | q | grep "first call\|second call" | cut -c1-110 >> $out/lsteam.txt
  timeout 600 python scripts/montecarlo.py cfg3 4096 2>&1 | q | grep "first call\|second call" | cut -c1-110 >> $out/lsteam.txt
done; done
cat $out/lsteam.txt
