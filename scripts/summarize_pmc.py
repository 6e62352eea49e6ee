#!/usr/bin/env python
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection.csv each) of bench.py into
profiles/<tag>_bench_hbm_counters.csv and profiles/riccati_traffic.json (read by bench.py for roofline.traffic).

    python scripts/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <tag> [window]

gfx950: FETCH_SIZE tallies 64 B per 128-B request on wide coalesced reads (MI355X_MICROARCH.md, "HBM"), so read
bytes = 2 x FETCH_SIZE KB; WRITE_SIZE is exact for 16-B-per-lane stores.  "Full" launches are those whose grid
covers the whole window of sub-problems."""
import csv, json, statistics, sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
fused = "--fused" in sys.argv           # the passes were taken with the fused sweep (bench.py's default from round 2 on)
argv = [a for a in sys.argv if a != "--fused"]
fetch_csv, write_csv, tag = argv[1], argv[2], argv[3]
window = int(argv[4]) if len(argv) > 4 else 2048
ALGO = 96160 if fused else 619360  # bytes per sub-problem backward pass, bench.py (fused: trajectory in, gains out)

seen_wave_producer = 'k_make_tiles_wave' in open(fetch_csv).read()


def short(name):
    # the record-fed wavefront sweep (bench.py's separate "tiles through HBM" measurement at the end of a run) apart from the fused one
    if "k_riccati_mfma<" in name and ", true>" not in name and "team" not in name and "general" not in name: return "riccati_records"
    if "k_riccati" in name: return "riccati"
    if "k_make_tiles_wave" in name: return "tiles"
    if "k_make_tiles" in name: return "tiles_static" if seen_wave_producer else "tiles"
    if "k_linesearch_wave" in name or "k_forward" in name: return "forward"
    return None

def load(path, counter):
    by = defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k and r["Counter_Name"] == counter:
            by[k].append((int(r["Grid_Size"]), float(r["Counter_Value"]), r["Kernel_Name"]))
    return by

F, Wr = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
rows = []
out = {}
for k in [c for c in ("riccati", "riccati_records", "tiles", "tiles_static", "forward") if c in F]:
    gmax = max(g for g, _, _ in F[k])
    full_f = [v for g, v, _ in F[k] if g == gmax]
    full_w = [v for g, v, _ in Wr[k] if g == gmax]
    # the first full producer launch of a solve also places the static part: use the median
    f, w = statistics.median(full_f), statistics.median(full_w)
    rows.append([k, len(F[k]), len(full_f), f, w, int(2 * f * 1024), int(w * 1024)])
    out[k] = (f, w, [n for g, _, n in F[k] if g == gmax][0])
with open(ROOT / "profiles" / f"{tag}_bench_hbm_counters.csv", "w", newline="") as fh:
    wr = csv.writer(fh)
    wr.writerow(["kernel", "launches_in_pass", "full_window_launches", "FETCH_SIZE_KB_median_full", "WRITE_SIZE_KB_median_full",
                 "corrected_read_bytes", "write_bytes"])
    wr.writerows(rows)
f, w, name = out["riccati"]
total = 2 * f * 1024 + w * 1024
js = {"kernel": name.split("(")[0].replace("void dpilqr::", ""), "window_items": window,
      "FETCH_SIZE_KB_full_window_launch": f, "WRITE_SIZE_KB_full_window_launch": w,
      "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM); "
                    "WRITE_SIZE exact for 16-B-per-lane stores",
      "hbm_bytes_per_full_window_launch": total, "hbm_bytes_per_subproblem_pass": total / window,
      "algorithmic_bytes_per_subproblem_pass": ALGO,
      "source": f"profiles/{tag}_bench_hbm_counters.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, "
                "bench.py --steps 16 --warmup 1 --reps 1 --no-cpu-baseline)"}
tj = ROOT / "profiles" / "riccati_traffic.json"
old = json.loads(tj.read_text()) if tj.exists() else {}
if fused:      # the fused sweep's entry beside the record-fed sweep's (bench.py reads that one when DPILQR_NO_FUSED is set)
    old["hbm_bytes_per_subproblem_pass_fused"] = js["hbm_bytes_per_subproblem_pass"]
    old["fused"] = js
    js = old
    if "riccati_records" in out:     # the same passes also saw bench.py's record-fed launches (full window): refresh that entry too
        fr, wrr, nm = out["riccati_records"]
        tot = 2 * fr * 1024 + wrr * 1024
        js.update({"kernel": nm.split("(")[0].replace("void dpilqr::", ""), "window_items": window,
                   "FETCH_SIZE_KB_full_window_launch": fr, "WRITE_SIZE_KB_full_window_launch": wrr,
                   "hbm_bytes_per_full_window_launch": tot, "hbm_bytes_per_subproblem_pass": tot / window,
                   "algorithmic_bytes_per_subproblem_pass": 619360,
                   "source": f"profiles/{tag}_bench_hbm_counters.csv, row riccati_records: the launches of bench.py's roofline_tiles_through_hbm "
                             "leg inside the same two passes"})
else:
    for key in ("hbm_bytes_per_subproblem_pass_fused", "fused"):
        if key in old:
            js[key] = old[key]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402  (the hash of the sweep's source these passes were taken on)
js["kernel_source_sha16"] = bench.kernel_source_sha16()
tj.write_text(json.dumps(js, indent=1))
for r in rows: print(r)
print(json.dumps(js, indent=1))
