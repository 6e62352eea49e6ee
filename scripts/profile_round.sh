#!/bin/bash
# The rocprofv3 passes behind profiles/r03_*: run on the GPU box from the repo root (gpurun -- 'bash scripts/profile_round.sh').
# Kernel trace + stats first, then the PMC counters in passes of their own (never together with a trace domain).  The program
# itself follows `--` (no env / bash -c hop: the profiler's library has initialised the GPU before the program starts).
# Summaries: the *_kernel_stats.csv of the first pass is copied as it is; scripts/summarize_pmc.py turns the FETCH_SIZE /
# WRITE_SIZE passes into profiles/r03_bench_hbm_counters.csv and profiles/riccati_traffic.json (which names the sweep's source
# hash: bench.py reports roofline.traffic = null when the file was measured on another version of the kernel).
R=${ROUND:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_kt -- python3 bench.py --steps 20 --warmup 5 --reps 5 --no-cpu-baseline --no-configs > gpurun_out/${R}_kt_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${R}_fetch -- python3 bench.py --steps 16 --warmup 1 --reps 1 --no-cpu-baseline --no-configs > gpurun_out/${R}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${R}_write -- python3 bench.py --steps 16 --warmup 1 --reps 1 --no-cpu-baseline --no-configs > gpurun_out/${R}_write.log 2>&1
rocprofv3 --pmc VALUBusy MfmaUtil --output-format csv -d gpurun_out/${R}_util -- python3 bench.py --steps 12 --warmup 1 --reps 1 --no-cpu-baseline --no-configs > gpurun_out/${R}_util.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/${R}_insts -- python3 bench.py --steps 12 --warmup 1 --reps 1 --no-cpu-baseline --no-configs > gpurun_out/${R}_insts.log 2>&1
F=$(ls gpurun_out/${R}_fetch/*/*counter_collection.csv | head -1); W=$(ls gpurun_out/${R}_write/*/*counter_collection.csv | head -1)
python3 scripts/summarize_pmc.py $F $W ${R} 6144 --fused > gpurun_out/${R}_pmc_summary.txt 2>&1
cp profiles/${R}_bench_hbm_counters.csv profiles/riccati_traffic.json gpurun_out/
cp $(ls gpurun_out/${R}_kt/*/*kernel_stats.csv | head -1) gpurun_out/${R}_bench_kernel_stats.csv
python3 scripts/summarize_counters.py gpurun_out/${R}_util gpurun_out/${R}_insts > gpurun_out/${R}_utilisation_counters.csv 2> gpurun_out/${R}_counters.err
tail -1 gpurun_out/${R}_kt_bench.log | cut -c1-300; head -12 gpurun_out/${R}_bench_kernel_stats.csv | cut -c1-200; tail -5 gpurun_out/${R}_pmc_summary.txt
