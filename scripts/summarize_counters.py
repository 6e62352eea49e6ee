#!/usr/bin/env python
"""rocprofv3 --pmc passes (counter_collection.csv under each given directory) -> one CSV on stdout: per kernel class and
counter the number of launches in the pass, how many of them had the largest grid ("full" launches: the whole window /
batch) and the median counter value over those.

    python scripts/summarize_counters.py <pass dir> [<pass dir> ...] > profiles/<tag>_utilisation_counters.csv

Kernel classes: the sweeps by variant (riccati_fused_w12 = k_riccati_mfma<..., 12, ..., true>, riccati_wg_fused, ...), the
line search, the generic forward pass, the tile producer, the rollout."""
import csv
import re
import statistics
import sys
from collections import defaultdict
from pathlib import Path


def short(name):
    m = re.search(r"k_riccati_mfma<\d+, \d+, (\d+), \d+, \d+(, true)?>", name)
    if m:
        return f"riccati_{'fused' if m.group(2) else 'records'}_w{m.group(1)}"
    if "k_riccati_mfma_team" in name: return "riccati_fused_team"
    if "k_riccati_mfma_pad" in name: return "riccati_records_padded"
    if "k_riccati_mfma_general" in name: return "riccati_fused_general"
    m = re.search(r"k_riccati_wg<(\d+), (\d+), \d+, \d+(, true)?>", name)
    if m:
        return f"riccati_wg_{'fused' if m.group(3) else 'records'}_n{m.group(1)}"
    if "k_riccati_big" in name: return "riccati_big"
    if "k_linesearch_wave" in name: return "linesearch"
    if "k_forward" in name: return "forward_generic"
    if "k_rollout_wave" in name: return "rollout"
    if "k_make_tiles" in name: return "tiles"
    return None


by = defaultdict(list)
for d in sys.argv[1:]:
    for f in Path(d).rglob("*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                by[(k, r["Counter_Name"])].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "launches_in_pass", "full_launches", "grid_threads_full", "median_over_full_launches"])
for (k, c), rows in sorted(by.items()):
    gmax = max(g for g, _ in rows)
    full = [v for g, v in rows if g == gmax]
    w.writerow([k, c, len(rows), len(full), gmax, statistics.median(full)])
