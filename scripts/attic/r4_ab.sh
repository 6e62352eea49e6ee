#!/bin/bash
# round 4, one GPU session: A/B measurements (every pair in the same session on the same device)
out=gpurun_out/r4d; mkdir -p $out; V=$PWD/dpilqr_amd/variants
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
( time python -m pytest tests/test_gpu_configs.py tests/test_gpu_big.py -x -q --durations=5 ) > $out/pytest_cfg.log 2>&1
{
echo "== line search, clusters of 7+ agents: per-agent gap in the staged K (main) vs none (nokgap)"
for lib in "" $V/libdpilqr_hip_nokgap.so; do
  echo "DPILQR_LIB=$lib"
  DPILQR_LIB=$lib python scripts/solve_breakdown.py --model uni4 9 12 15 2>&1 | q
  DPILQR_LIB=$lib python scripts/solve_breakdown.py --model quad6 7 8 10 2>&1 | q
done
echo "== line search, cfg2: gap for every size (kgapall) vs main"
for lib in "" $V/libdpilqr_hip_kgapall.so; do echo "DPILQR_LIB=$lib"; DPILQR_LIB=$lib python scripts/bench_ls.py 2>&1 | q; done
echo "== record-fed sweep, 6144 items: non-temporal record loads (recnt) vs main"
for lib in "" $V/libdpilqr_hip_recnt.so "" $V/libdpilqr_hip_recnt.so; do echo "DPILQR_LIB=$lib"; REALISTIC=4 DPILQR_LIB=$lib python scripts/bench_riccati.py 2048 6144 2>&1 | q | head -2; done
echo "== phase stamps, fused sweep, 1024 and 6144 items"
python scripts/phase_stamps.py 1024 --fused 2>&1 | q
python scripts/phase_stamps.py 6144 --fused 2>&1 | q
echo "== FMA contraction library-wide (fma) vs main: bench, 20 steps"
for lib in "" $V/libdpilqr_hip_fma.so; do echo "DPILQR_LIB=$lib"; DPILQR_LIB=$lib python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python scripts/benchline.py; done
} > $out/ab.txt 2>&1
DPILQR_LIB=$V/libdpilqr_hip_fma.so python -m pytest tests -m gpu -q -k "not cfg4 and not cfg5 and not bench" > $out/pytest_fma.log 2>&1
{
echo "== cfg4, 4096 scenarios: padded wavefront sweep for n_x = 6, 18 (main) vs DPILQR_RICCATI_NO_PAD=1"
python scripts/montecarlo.py cfg4 4096 2>&1 | q | cut -c1-330
DPILQR_RICCATI_NO_PAD=1 python scripts/montecarlo.py cfg4 4096 2>&1 | q | cut -c1-330
} >> $out/ab.txt 2>&1
tail -4 $out/pytest_cfg.log; grep -E "passed|failed" $out/pytest_fma.log | tail -2
