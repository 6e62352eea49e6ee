#!/bin/bash
# The rocprofv3 passes behind profiles/r02_*: run on the GPU box from the repo root (gpurun -- 'bash scripts/profile_round.sh');
# kernel trace + stats, then the PMC counters in passes of their own (never together with a trace domain); summaries are made
# with scripts/summarize_pmc.py ... --fused and copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_kt -- python3 bench.py --steps 100 --warmup 0 --no-cpu-baseline > gpurun_out/r02_kt_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02_fetch -- python3 bench.py --steps 16 --warmup 1 --no-cpu-baseline > gpurun_out/r02_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02_write -- python3 bench.py --steps 16 --warmup 1 --no-cpu-baseline > gpurun_out/r02_write.log 2>&1
rocprofv3 --pmc VALUBusy MfmaUtil --output-format csv -d gpurun_out/r02_util -- python3 bench.py --steps 12 --warmup 1 --no-cpu-baseline > gpurun_out/r02_util.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/r02_insts -- python3 bench.py --steps 12 --warmup 1 --no-cpu-baseline > gpurun_out/r02_insts.log 2>&1
ls gpurun_out/r02_kt/*/ | head; tail -1 gpurun_out/r02_kt_bench.log | cut -c1-300
