bash scripts/profile_round.sh > gpurun_out/r03_profile_round.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_20steps.json 2> gpurun_out/r03_bench_20.err
python bench.py --steps 100 --warmup 5 > gpurun_out/r03_bench_100steps.json 2> gpurun_out/r03_bench_100.err
python scripts/bench_wg.py --model quad6 4 5 6 7 8 9 10 > gpurun_out/r03_wg_sweep_quad6.txt 2>&1
python scripts/bench_wg.py --model uni4 6 7 9 11 12 13 15 > gpurun_out/r03_wg_sweep_uni4.txt 2>&1
( echo "# python scripts/phase_stamps.py --wg 2048 {10 quad6 | 15 uni4} [--fused]   (a -DDPILQR_PHASE_STAMPS build: python scripts/phase_stamps.py --build; ticks = shader clocks; record-fed first, fused second)"
  python scripts/phase_stamps.py --wg 2048 10 quad6; python scripts/phase_stamps.py --wg 2048 10 quad6 --fused
  python scripts/phase_stamps.py --wg 2048 15 uni4; python scripts/phase_stamps.py --wg 2048 15 uni4 --fused
  python scripts/phase_stamps.py --wg 2048 6 quad6; python scripts/phase_stamps.py --wg 2048 6 quad6 --fused ) > gpurun_out/r03_wg_phases.txt 2>&1
python scripts/montecarlo.py cfg4 8192 gpurun_out/r03_cfg4_8192_scenarios.json > gpurun_out/r03_cfg4.log 2>&1
python scripts/montecarlo.py cfg3 4096 gpurun_out/r03_cfg3_4096_scenarios.json > gpurun_out/r03_cfg3.log 2>&1
python scripts/bench_big.py 1 32 256 > gpurun_out/r03_cfg5_bench_big.txt 2>&1
tail -3 gpurun_out/r03_cfg4.log | cut -c1-300; tail -3 gpurun_out/r03_cfg3.log | cut -c1-300; tail -4 gpurun_out/r03_cfg5_bench_big.txt | cut -c1-200
cut -c1-250 gpurun_out/r03_bench_20steps.json; cut -c1-250 gpurun_out/r03_bench_100steps.json
