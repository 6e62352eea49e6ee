"""The PMC traffic file behind bench.py's roofline.traffic names the sweep source it was measured on; bench.py reports
traffic = null for a stale file.  This check makes the staleness visible on CPU, before the bench runs: after a change to the
sweep's source, re-run scripts/profile_round.sh on the GPU box and commit its summaries."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def test_traffic_file_was_measured_on_the_current_sweep_source():
    import bench
    d = json.loads((ROOT / "profiles" / "riccati_traffic.json").read_text())
    assert d["kernel_source_sha16"] == bench.kernel_source_sha16(), \
        "profiles/riccati_traffic.json is stale: gpurun -- 'bash scripts/profile_round.sh', then copy the summaries into profiles/"
    f = d["fused"]
    # counters within 5 % of the algorithmic bytes of the fused sweep (no wasted re-reads), never below them
    assert 1.0 <= f["hbm_bytes_per_subproblem_pass"] / f["algorithmic_bytes_per_subproblem_pass"] < 1.05
