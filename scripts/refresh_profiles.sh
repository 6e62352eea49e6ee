#!/bin/bash
# One gpurun call that regenerates the round's profiles/<R>_* files (copy them from gpurun_out/ afterwards):
#   gpurun --timeout 3000 -- 'ROUND=r04 bash scripts/refresh_profiles.sh'
R=${ROUND:-r04}
q() { grep -v "Warning\|x\[mask\]\|amdgpu.ids"; }
ROUND=$R bash scripts/profile_round.sh > gpurun_out/${R}_profile_round.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench_20steps.json 2> gpurun_out/${R}_bench_20.err
python bench.py --steps 100 --warmup 5 > gpurun_out/${R}_bench_100steps.json 2> gpurun_out/${R}_bench_100.err
python scripts/bench_wg.py --model quad6 4 5 6 7 8 9 10 2>&1 | q > gpurun_out/${R}_wg_sweep_quad6.txt
python scripts/bench_wg.py --model uni4 6 7 9 11 12 13 15 2>&1 | q > gpurun_out/${R}_wg_sweep_uni4.txt
( for m in "uni4 3 5 9 12 15" "quad6 1 2 3 5 8 10"; do python scripts/solve_breakdown.py --model $m; done ) 2>&1 | q > gpurun_out/${R}_solve_breakdown.txt
python scripts/bench_q12.py 1 2 3 5 2>&1 | q > gpurun_out/${R}_quad12_small.txt
python scripts/sweep_waves_ab.py 64 256 512 1024 2048 6144 2>&1 | q > gpurun_out/${R}_fused_sweep_sizes.txt
ROUND=$R bash scripts/profile_wg.sh > gpurun_out/${R}_profile_wg.log 2>&1
python scripts/montecarlo.py cfg4 8192 gpurun_out/${R}_cfg4_8192_scenarios.json > gpurun_out/${R}_cfg4.log 2>&1
python scripts/montecarlo.py cfg3 4096 gpurun_out/${R}_cfg3_4096_scenarios.json > gpurun_out/${R}_cfg3.log 2>&1
python scripts/bench_big.py 1 8 32 256 2>&1 | q > gpurun_out/${R}_cfg5_bench_big.txt
python scripts/big_team_check.py 1 8 32 64 128 2>&1 | q > gpurun_out/${R}_big_team.txt
python scripts/cfg5_solve_time.py 8 2>&1 | q > gpurun_out/${R}_cfg5_solve.txt
python scripts/bench_ls_sizes.py uni4:15 uni4:14 uni4:12 uni4:8 quad6:10 quad6:8 quad6:6 2>&1 | q > gpurun_out/${R}_ls_sizes.txt
python scripts/kernel_resources.py --csv gpurun_out/${R}_kernel_resources.csv > /dev/null 2>&1
ROUND=$R bash scripts/profile_configs.sh > gpurun_out/${R}_profile_configs.log 2>&1
scripts/ubench/l1_inv > gpurun_out/${R}_l1_inv_raw.txt 2>&1; scripts/ubench/mfma_chain > gpurun_out/${R}_mfma_chain.txt 2>&1; scripts/ubench/trig_inline_check > gpurun_out/${R}_trig_inline_check.txt 2>&1
bash scripts/r06_fwd_phases.sh > /dev/null 2>&1
tail -3 gpurun_out/${R}_cfg4.log | cut -c1-300; tail -3 gpurun_out/${R}_cfg3.log | cut -c1-300; tail -4 gpurun_out/${R}_cfg5_bench_big.txt | cut -c1-200
cut -c1-250 gpurun_out/${R}_bench_20steps.json; cut -c1-250 gpurun_out/${R}_bench_100steps.json
