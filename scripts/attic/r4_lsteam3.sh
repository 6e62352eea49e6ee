#!/bin/bash
out=gpurun_out/r4l; mkdir -p $out
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
rm -f $out/lsteam3.txt
for rep in 1 2; do
for t in 1024 2048 6144; do
  export DPILQR_LS_TEAM_MAX=$t
  echo "== team max $t" >> $out/lsteam3.txt
  for m in "uni4 5" "uni4 3" "quad6 2" "quad6 4"; do timeout 300 python scripts/solve_breakdown.py --model $m 2>&1 | q | tail -1 | cut -c1-40,130-250 >> $out/lsteam3.txt; done
  timeout 300 python scripts/bench_ls.py --iters 8 --B 2048 2>&1 | q >> $out/lsteam3.txt
  timeout 300 python scripts/bench_ls.py --iters 8 --B 6144 2>&1 | q >> $out/lsteam3.txt
  timeout 600 python scripts/montecarlo.py cfg3 4096 2>&1 | q | grep "first call\|second call" | cut -c1-110 >> $out/lsteam3.txt
done; done
cat $out/lsteam3.txt
