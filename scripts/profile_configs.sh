#!/bin/bash
# rocprofv3 kernel statistics of the cfg3 / cfg4 / cfg5 workloads (profiles/r02_cfg*_kernel_stats.csv): run on the GPU box from
# the repo root through gpurun.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_kt_cfg4 -- python3 scripts/montecarlo.py cfg4 2048 > gpurun_out/r02_kt_cfg4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_kt_cfg3 -- python3 scripts/montecarlo.py cfg3 256 > gpurun_out/r02_kt_cfg3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_kt_cfg5 -- python3 scripts/bench_big.py 256 > gpurun_out/r02_kt_cfg5.log 2>&1
tail -2 gpurun_out/r02_kt_cfg4.log | cut -c1-250; tail -2 gpurun_out/r02_kt_cfg5.log | cut -c1-250
