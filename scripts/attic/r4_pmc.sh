#!/bin/bash
# stall / LDS counters of the fused sweep at 6144 (three wavefronts per SIMD), 2048 (two) and 1024 items (the team kernel)
out=gpurun_out/r4f; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $out/p1 -- python3 scripts/sweep_waves_ab.py 6144 2048 1024 > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/p2 -- python3 scripts/sweep_waves_ab.py 6144 2048 1024 > $out/p2.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $out/p3 -- python3 scripts/sweep_waves_ab.py 6144 2048 1024 > $out/p3.log 2>&1
python3 scripts/summarize_counters.py $out/p1 $out/p2 $out/p3 > $out/counters.csv 2> $out/counters.err
tail -2 $out/p1.log $out/p2.log $out/p3.log; cat $out/counters.csv | grep -i "riccati_fused"
