#!/bin/bash
# Counters of the mid-size sweep k_riccati_wg (profiles/r03_wg_counters.csv): run on the GPU box from the repo root through gpurun.
# Each --pmc set in a pass of its own, never together with a trace domain.  The workload: scripts/solve_breakdown.py, one
# windowed solve of 2048 clusters of 15 Unicycle4D / 10 Quadcopter6D agents (the fused sweep, the line search).
R=${ROUND:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in "uni4 15" "quad6 10"; do
  tag=$(echo $m | tr ' ' '_')
  rocprofv3 --pmc VALUBusy MfmaUtil --output-format csv -d gpurun_out/${R}_wg_util_$tag -- python3 scripts/solve_breakdown.py --model $m > gpurun_out/${R}_wg_util_$tag.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${R}_wg_fetch_$tag -- python3 scripts/solve_breakdown.py --model $m > gpurun_out/${R}_wg_fetch_$tag.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${R}_wg_write_$tag -- python3 scripts/solve_breakdown.py --model $m > gpurun_out/${R}_wg_write_$tag.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE --output-format csv -d gpurun_out/${R}_wg_insts_$tag -- python3 scripts/solve_breakdown.py --model $m > gpurun_out/${R}_wg_insts_$tag.log 2>&1
  python3 scripts/summarize_counters.py gpurun_out/${R}_wg_util_$tag gpurun_out/${R}_wg_fetch_$tag gpurun_out/${R}_wg_write_$tag gpurun_out/${R}_wg_insts_$tag | sed "s/^/$tag,/" > gpurun_out/${R}_wg_counters_$tag.csv
done
cat gpurun_out/${R}_wg_counters_uni4_15.csv gpurun_out/${R}_wg_counters_quad6_10.csv | grep -v "^.*,kernel,counter" > gpurun_out/${R}_wg_counters.csv
sed -i '1i workload,kernel,counter,launches_in_pass,full_launches,grid_threads_full,median_over_full_launches' gpurun_out/${R}_wg_counters.csv
grep -E "riccati_wg|linesearch" gpurun_out/${R}_wg_counters.csv | grep -E "VALUBusy|MfmaUtil|FETCH|WRITE"
