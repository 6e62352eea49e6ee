cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_kt_cfg4 -- python3 scripts/montecarlo.py cfg4 8192 > gpurun_out/r03_kt_cfg4.log 2>&1
tail -3 gpurun_out/r03_kt_cfg4.log | cut -c1-200
f=$(ls gpurun_out/r03_kt_cfg4/*/*kernel_stats.csv | head -1); cp $f gpurun_out/r03_cfg4_kernel_stats.csv
python3 - <<'PY'
import csv, glob
rows = list(csv.DictReader(open('gpurun_out/r03_cfg4_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e9
print('sum of kernel durations %.3f s' % tot)
cls = {}
for r in rows:
    n = r['Name']
    k = 'sweep' if 'riccati' in n else 'linesearch' if 'linesearch' in n else 'tiles' if 'make_tiles' in n else 'rollout' if 'rollout' in n else 'other'
    cls[k] = cls.get(k, 0) + float(r['TotalDurationNs']) / 1e9
print({k: round(v, 3) for k, v in cls.items()})
PY
