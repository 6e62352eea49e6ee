#!/usr/bin/env python
"""Compile the HIP library to gfx950 assembly and print per-kernel instruction statistics
(static counts; the tiled Riccati sweep is issue-bound at one wave per SIMD, so instruction count ~ time)."""
import re
import subprocess
import sys
from collections import Counter
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
pat = sys.argv[1] if len(sys.argv) > 1 else "k_riccati_tiledILi20ELi10"
out = Path("/tmp/dpilqr.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                f"-I{ROOT/'include'}", f"-I{ROOT/'dpilqr_amd'/'csrc'}", "-S", "--cuda-device-only", "-o", str(out),
                str(ROOT / "dpilqr_amd" / "csrc" / "dpilqr_hip.hip")], check=True, stderr=subprocess.DEVNULL)
s = out.read_text()
for m in re.finditer(r"^(_Z\w+):[^\n]*\n", s, re.M):
    name = m.group(1)
    if pat not in name:
        continue
    end = s.index(".Lfunc_end", m.end())
    body = s[m.end():end]
    ins = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
    c = Counter(l.split()[0] for l in ins)
    meta = re.search(re.escape(name) + r".*?\.vgpr_count:\s*(\d+)", s[end:], re.S)
    agpr = re.search(r"\.agpr_count:\s*(\d+)", s[end:end + 20000])
    print(f"== {name}: {len(ins)} instructions")
    groups = {"fma/mul/add f64": sum(v for k, v in c.items() if re.match(r"v_(fma|fmac|mul|add|max|min)_f64", k)),
              "ds_read": sum(v for k, v in c.items() if k.startswith("ds_read")),
              "ds_write": sum(v for k, v in c.items() if k.startswith("ds_write")),
              "accvgpr": sum(v for k, v in c.items() if "accvgpr" in k),
              "v_mov": sum(v for k, v in c.items() if k.startswith("v_mov")),
              "readlane": c.get("v_readlane_b32", 0), "writelane": c.get("v_writelane_b32", 0),
              "cndmask": sum(v for k, v in c.items() if k.startswith("v_cndmask")),
              "waitcnt": c.get("s_waitcnt", 0), "s_nop": c.get("s_nop", 0),
              "global": sum(v for k, v in c.items() if k.startswith("global_")),
              "scratch": sum(v for k, v in c.items() if k.startswith("scratch_")),
              "branches": sum(v for k, v in c.items() if k.startswith("s_cbranch") or k == "s_branch")}
    print("   " + "  ".join(f"{k}={v}" for k, v in groups.items()))
    rest = sum(c.values()) - sum(groups.values())
    print(f"   other={rest}; top: " + ", ".join(f"{k}:{v}" for k, v in c.most_common(12)))
