#!/bin/bash
# Round 6: the line search of cfg3 / cfg4 before and after its spills were removed (profiles/r06_ls_spills.txt).
#   gpurun -- 'bash scripts/r06_ls_ab.sh'     needs dpilqr_amd/variants/libdpilqr_hip_lsold.so (the previous forward_wave.hpp)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$R/gpurun_out/r06_ls; mkdir -p $out
for arm in old new; do
  if [ $arm == old ]; then export DPILQR_DEBUG_ROUTES=1 DPILQR_LIB=$R/dpilqr_amd/variants/libdpilqr_hip_lsold.so; else unset DPILQR_LIB DPILQR_DEBUG_ROUTES; fi
  for cfg in "cfg3 4096" "cfg4 8192"; do
    tag=${arm}_${cfg%% *}
    [ -f $out/$tag.txt ] || python3 $R/scripts/montecarlo.py $cfg > $out/$tag.txt 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$tag -- python3 $R/scripts/montecarlo.py $cfg > $out/${tag}_prof.txt 2>&1
    f=$(find $out/prof_$tag -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp $f $out/${tag}_kernel_stats.csv
    rm -rf $out/prof_$tag
  done
done
tail -n 4 $out/*_cfg?.txt
