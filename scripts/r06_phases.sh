#!/bin/bash
# Round 6 diagnostics on the GPU box: phase stamps of the mid-size sweep and of config 5's sweep (the -DDPILQR_PHASE_STAMPS variant),
# the line search per cluster size with the previous and the current forward_wave.hpp.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
out=$R/gpurun_out/r06_phases; mkdir -p $out
cd $R
python3 scripts/phase_stamps.py --wg 2048 15 uni4 --fused > $out/wg_uni4_15_fused.txt 2>&1
python3 scripts/phase_stamps.py --wg 2048 10 quad6 --fused > $out/wg_quad6_10_fused.txt 2>&1
python3 scripts/phase_stamps.py --wg 2048 15 uni4 > $out/wg_uni4_15_rec.txt 2>&1
export DPILQR_DEBUG_ROUTES=1 DPILQR_LIB=$R/dpilqr_amd/variants/libdpilqr_stamps.so
python3 scripts/bench_big.py 1 > $out/big_stamps_team.txt 2>&1
DPILQR_BIG_TEAM=0 python3 scripts/bench_big.py 1 > $out/big_stamps_alone.txt 2>&1
export DPILQR_LIB=$R/dpilqr_amd/variants/libdpilqr_hip_lsold.so
python3 scripts/bench_ls_sizes.py > $out/ls_sizes_old.txt 2>&1
unset DPILQR_LIB DPILQR_DEBUG_ROUTES
python3 scripts/bench_ls_sizes.py > $out/ls_sizes_new.txt 2>&1
tail -n 12 $out/*.txt
