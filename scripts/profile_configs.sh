#!/bin/bash
# rocprofv3 kernel statistics of the cfg3 / cfg4 / cfg5 workloads at the sizes of scripts/montecarlo.py's Monte-Carlo runs
# (profiles/<R>_cfg*_kernel_stats.csv): run on the GPU box from the repo root through gpurun.
R=${ROUND:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_kt_cfg4 -- python3 scripts/montecarlo.py cfg4 8192 > gpurun_out/${R}_kt_cfg4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_kt_cfg3 -- python3 scripts/montecarlo.py cfg3 4096 > gpurun_out/${R}_kt_cfg3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_kt_cfg5 -- python3 scripts/bench_big.py 256 > gpurun_out/${R}_kt_cfg5.log 2>&1
for c in cfg4 cfg3 cfg5; do f=$(find gpurun_out/${R}_kt_$c -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-90,160-260 > gpurun_out/${R}_${c}_kernel_stats_top.txt; cp $f gpurun_out/${R}_${c}_kernel_stats.csv; done
tail -2 gpurun_out/${R}_kt_cfg4.log | cut -c1-250; head -8 gpurun_out/${R}_cfg4_kernel_stats.csv | cut -c1-200
