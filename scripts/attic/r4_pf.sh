#!/bin/bash
# A/B inside one gpurun call of line-search builds: default against the variants given as arguments (dpilqr_amd/variants/libdpilqr_hip_<tag>.so)
out=gpurun_out/r4q; mkdir -p $out
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
export DPILQR_LS_NO_PACK=${NO_PACK-1}
[ -z "$DPILQR_LS_NO_PACK" ] && unset DPILQR_LS_NO_PACK
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "two_pass or bit_identical or solve_cfg2 or window_invariance or enqueue or all_items or golden or trace" > $out/pytest.log 2>&1; tail -3 $out/pytest.log
{
for rep in 1 2; do
for t in default "$@"; do
  if [ $t != default ]; then export DPILQR_LIB=$PWD/dpilqr_amd/variants/libdpilqr_hip_$t.so; else unset DPILQR_LIB; fi
  echo "== $t"
  timeout 300 python scripts/bench_ls.py --iters 8 2>&1 | q
  timeout 300 python scripts/bench_ls.py --iters 8 --B 1024 2>&1 | q
done; done
for t in default "$@" default "$@"; do
  if [ $t != default ]; then export DPILQR_LIB=$PWD/dpilqr_amd/variants/libdpilqr_hip_$t.so; else unset DPILQR_LIB; fi
  echo "== $t"
  timeout 600 python bench.py --steps 20 --no-cpu-baseline --profile-all 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), 'sub/s', 'single_batch ms', round(d['single_batch_1024']['ms'],3), {k: round(v,4) for k,v in d['kernel_ms_per_step'].items()})"
done
for m in "uni4 5" "uni4 15" "quad6 4"; do
for t in default "$@"; do
  if [ $t != default ]; then export DPILQR_LIB=$PWD/dpilqr_amd/variants/libdpilqr_hip_$t.so; else unset DPILQR_LIB; fi
  echo "== $t $m"; timeout 300 python scripts/solve_breakdown.py --model $m 2>&1 | q | tail -1 | sed 's/.*riccati/riccati/'
done; done
} > $out/pf.txt 2>&1
cat $out/pf.txt
