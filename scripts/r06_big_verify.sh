#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
cd $R; out=$R/gpurun_out/r06_big; mkdir -p $out
python3 scripts/big_team_check.py 1 8 32 64 128 > $out/team_check.txt 2>&1; tail -12 $out/team_check.txt
python -m pytest tests/test_gpu_big.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | head
bash scripts/cfg5_soak.sh 6 2>&1 | tail -2
python3 scripts/cfg5_solve_time.py 8 > $out/cfg5_solve.txt 2>&1; cat $out/cfg5_solve.txt | grep -v amdgpu
