#!/bin/bash
# Counters of the mid-size sweep k_riccati_wg (profiles/r02_wg_counters.csv): run on the GPU box from the repo root through gpurun.
# Each --pmc set in a pass of its own, never together with a trace domain.  The workload: scripts/solve_breakdown.py, one
# windowed solve of 2048 clusters of 15 Unicycle4D / 10 Quadcopter6D agents (the fused sweep, the line search).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in "uni4 15" "quad6 10"; do
  tag=$(echo $m | tr ' ' '_')
  rocprofv3 --pmc VALUBusy MfmaUtil --output-format csv -d gpurun_out/r02_wg_util_$tag -- python3 scripts/solve_breakdown.py --model $m > gpurun_out/r02_wg_util_$tag.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02_wg_fetch_$tag -- python3 scripts/solve_breakdown.py --model $m > gpurun_out/r02_wg_fetch_$tag.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02_wg_write_$tag -- python3 scripts/solve_breakdown.py --model $m > gpurun_out/r02_wg_write_$tag.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE --output-format csv -d gpurun_out/r02_wg_insts_$tag -- python3 scripts/solve_breakdown.py --model $m > gpurun_out/r02_wg_insts_$tag.log 2>&1
done
ls gpurun_out/r02_wg_util_uni4_15/*/ | head -3
