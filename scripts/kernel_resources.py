#!/usr/bin/env python
"""Register / scratch / LDS budget of every kernel in the BUILT library (dpilqr_amd/libdpilqr_hip.so).

Reads the gfx950 code objects out of the .so's .hip_fatbin section (one clang offload bundle per translation unit) and their
AMDGPU metadata notes (llvm-readelf --notes): per kernel .vgpr_count, .agpr_count, .sgpr_count, .vgpr_spill_count,
.sgpr_spill_count, .private_segment_fixed_size (scratch bytes per lane), .group_segment_fixed_size (static LDS),
.max_flat_workgroup_size.  No recompilation: what is listed is what ships.

    python scripts/kernel_resources.py [--csv profiles/r03_kernel_resources.csv] [--filter k_riccati_wg]

tests/test_kernel_resources.py uses resources() to fail the build when a hot-path kernel spills."""
import argparse
import csv
import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = ["vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size"]


def code_objects(so_path):
    """The gfx950 ELF images embedded in the shared library."""
    with tempfile.TemporaryDirectory() as td:
        fat = Path(td) / "fat.bin"
        subprocess.run([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", str(so_path), str(fat)], check=True)
        blob = fat.read_bytes()
    out = []
    pos = blob.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(MAGIC, pos + len(MAGIC))
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    d = r.stdout.split("\n") if r.returncode == 0 else names
    return dict(zip(names, d))


def resources(so_path=ROOT / "dpilqr_amd" / "libdpilqr_hip.so"):
    """[{name, demangled, vgpr_count, ..., private_segment_fixed_size, ...}] for every kernel of the library."""
    rows = []
    for img in code_objects(so_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img); f.flush()
            notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", f.name], capture_output=True, text=True, check=True).stdout
        # one YAML-ish block per kernel: "- .agpr_count: ..\n  .args: ...\n  .name: _Z...\n ..."
        for block in re.split(r"\n\s*- (?=\.agpr_count|\.args)", notes):
            m = re.search(r"^\s*\.name:\s*(\S+)", block, re.M)
            if not m or ".vgpr_count" not in block:
                continue
            row = {"name": m.group(1)}
            for fld in FIELDS:
                v = re.search(r"^\s*\." + fld + r":\s*(\d+)", block, re.M)
                row[fld] = int(v.group(1)) if v else 0
            rows.append(row)
    dm = demangle([r["name"] for r in rows])
    for r in rows:
        d = dm.get(r["name"], r["name"])
        r["demangled"] = re.sub(r"^void dpilqr::", "", d.split("(")[0])
    rows.sort(key=lambda r: r["demangled"])
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--csv", default=None)
    ap.add_argument("--filter", default="")
    a = ap.parse_args()
    rows = [r for r in resources() if a.filter in r["demangled"]]
    if a.csv:
        with open(a.csv, "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel"] + FIELDS)
            for r in rows:
                w.writerow([r["demangled"]] + [r[f] for f in FIELDS])
    spilling = [r for r in rows if r["vgpr_spill_count"] or r["sgpr_spill_count"] or r["private_segment_fixed_size"]]
    for r in rows:
        print(f"{r['demangled']:<70} vgpr {r['vgpr_count']:>3} agpr {r['agpr_count']:>3} sgpr {r['sgpr_count']:>3} "
              f"spill v{r['vgpr_spill_count']:>3} s{r['sgpr_spill_count']:>4} scratch {r['private_segment_fixed_size']:>4} B  "
              f"lds {r['group_segment_fixed_size']:>6}  wg {r['max_flat_workgroup_size']}")
    print(f"{len(rows)} kernels, {len(spilling)} with spills or scratch", file=sys.stderr)


if __name__ == "__main__":
    main()
