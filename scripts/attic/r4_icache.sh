#!/bin/bash
# instruction-cache counters of the mid-size sweep (k_riccati_wg) and of the cfg2 kernels: is the unrolled code's size a bound?
out=gpurun_out/r4c; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|SQC_|IFETCH|INST_CACHE" | head -40 > $out/avail.txt
cat $out/avail.txt | cut -c1-160
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $out/ic1 -- python3 scripts/solve_breakdown.py --model quad6 10 > $out/ic1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $out/ic2 -- python3 scripts/solve_breakdown.py --model quad6 10 > $out/ic2.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $out/ic3 -- python3 scripts/bench_ls.py --iters 4 --reps 2 > $out/ic3.log 2>&1
tail -2 $out/ic1.log $out/ic2.log $out/ic3.log | cut -c1-200
python3 - <<PY
import csv,glob,collections
for d in ("ic1","ic2","ic3"):
    fs=glob.glob("$out/"+d+"/**/*counter_collection.csv",recursive=True)
    if not fs: print(d,"no csv"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"].split("(")[0][-60:]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    for k,v in agg.items():
        if any(x in k for x in ("riccati","linesearch")): print(d,k,{a:round(b) for a,b in v.items()})
PY
