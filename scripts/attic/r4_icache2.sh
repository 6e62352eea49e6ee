#!/bin/bash
out=gpurun_out/r4c; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -B4 "SQ_IFETCH_LEVEL, HIGH_RES" | head -12 | cut -c1-200
for m in "quad6 10" "quad6 7" "quad6 5"; do
  tag=$(echo $m | tr ' ' '_')
  rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQC_TC_INST_REQ --output-format csv -d $out/if_$tag -- python3 scripts/solve_breakdown.py --model $m > $out/if_$tag.log 2>&1
done
python3 - <<PY
import csv,glob,collections
for d in ("if_quad6_10","if_quad6_7","if_quad6_5"):
    fs=glob.glob("$out/"+d+"/**/*counter_collection.csv",recursive=True)
    if not fs: print(d,"no csv"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        k=r["Kernel_Name"].split("(")[0][-45:]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in agg.items():
        if "riccati" in k:
            w=v.get("SQ_WAVE_CYCLES",1)
            print(d,k,{a:round(b/w,4) for a,b in v.items() if a!="SQ_WAVE_CYCLES"}, "ifetch latency", round(v.get("SQ_IFETCH_LEVEL",0)/max(v.get("SQ_IFETCH",1),1),1))
PY
