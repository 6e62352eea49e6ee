"""build()'s staleness stamp is a content hash with repository-relative names: a copy of the tree at another path, with
other modification times, compiles ZERO translation units and never invokes hipcc (round-4 review: absolute -I paths inside
the stamp made the GPU box rebuild all nine units with its own compiler, so the smoke test validated a binary nobody ships)."""
import os
import shutil
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _copy_tree(dst):
    for name in ("__graft_entry__.py", "include", "dpilqr_amd", "oracle"):
        src = ROOT / name
        if src.is_dir():
            shutil.copytree(src, dst / name, ignore=shutil.ignore_patterns("__pycache__", "variants", "_ref"),
                            copy_function=shutil.copy)          # shutil.copy: NEW modification times, in directory order
        else:
            shutil.copy(src, dst / name)


def _build_in(dst, extra_env=None):
    env = dict(os.environ, HIPCC="/bin/false", **(extra_env or {}))      # any compiler invocation fails the build
    code = ("import json, __graft_entry__ as g; g.build(); "
            "print('LAST_BUILD ' + json.dumps(g.last_build) + ' ' + str(g.ROOT))")
    return subprocess.run([sys.executable, "-c", code], cwd=str(dst), env=env, capture_output=True, text=True, timeout=600)


def test_copy_at_another_path_compiles_nothing(tmp_path):
    import __graft_entry__ as g
    g.build()
    dst = tmp_path / "elsewhere" / "tree"
    dst.mkdir(parents=True)
    _copy_tree(dst)
    # make the library OLDER than every source: modification times must not matter
    old = time.time() - 86400
    os.utime(dst / "dpilqr_amd" / "libdpilqr_hip.so", (old, old))
    out = _build_in(dst)
    assert out.returncode == 0, out.stdout + out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("LAST_BUILD ")][-1]
    assert '"compiled": []' in line and '"linked": false' in line, line
    assert str(dst) in line                                     # it really was the copy's own __graft_entry__
    assert "compiled 0 of" in out.stdout


def test_a_changed_source_is_noticed(tmp_path):
    """...and the stamp is not vacuous: touching the CONTENT of one header makes the copy stale (the fake compiler fails)."""
    import __graft_entry__ as g
    g.build()
    dst = tmp_path / "tree"
    dst.mkdir()
    _copy_tree(dst)
    hdr = dst / "dpilqr_amd" / "csrc" / "solve_state.hpp"
    hdr.write_text(hdr.read_text() + "\n// changed\n")
    out = _build_in(dst)
    assert out.returncode != 0 and "hipcc failed" in out.stderr, out.stdout + out.stderr


def test_flags_name_no_absolute_path():
    import __graft_entry__ as g
    assert not any(str(g.ROOT) in f or f.startswith("-I/") for f in g.HIPCC_FLAGS), g.HIPCC_FLAGS
