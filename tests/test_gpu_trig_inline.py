"""dpilqr_amd/csrc/trig_inline.hpp -- sin / cos / tan written out as one basic block for the twelve-state quadcopter's stage evaluations --
against the device library's sincos() and tan(), bit for bit (scripts/ubench/trig_inline_check.hip: 2^26 arguments per range)."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
def test_inline_trig_is_the_device_librarys_bit_for_bit(tmp_path):
    exe = tmp_path / "trig_inline_check"
    src = ROOT / "scripts" / "ubench" / "trig_inline_check.hip"
    # the kernels' own flags: above all -ffp-contract=off (a product and a sum written apart stay apart)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", f"-I{ROOT / 'dpilqr_amd' / 'csrc'}",
                    "-o", str(exe), str(src)], check=True, capture_output=True, timeout=600)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if "arguments" in l]
    assert len(lines) == 3 and all(": 0 differ" in l for l in lines), out.stdout
