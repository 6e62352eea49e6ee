"""The C-ABI library loads without a GPU and exports every symbol include/dpilqr_hip.h declares (CPU only;
no compute entry point is called here)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "dpilqr_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dpilqr_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from dpilqr_amd import _lib
    return _lib


def test_header_and_binding_agree(lib):
    names = declared_symbols()
    assert len(names) >= 20
    assert set(names) == set(lib.SIGNATURES), set(names) ^ set(lib.SIGNATURES)


def test_every_symbol_exported(lib):
    L = lib.load()
    for name in declared_symbols():
        assert hasattr(L, name), f"{name} declared in dpilqr_hip.h but not exported"
    assert L.dpilqr_abi_version() == 4


def test_host_only_entry_points(lib):
    """tile layout / alpha table / model dims are pure host functions: usable (and checked) without a GPU."""
    lay, stride = lib.tile_layout(20, 10)
    assert stride == 1330 and lay["A"] == (0, 30) and lay["B"] == (20, 30) and lay["Lxx"][0] == 600
    assert lay["Lu"][0] == lay["Lx"][0] + 20
    a = lib.alphas()
    import numpy as np
    assert np.array_equal(np.array(a, dtype=np.float32), (1.1 ** (-np.arange(10, dtype=np.float32) ** 2)))
    ns, nc = C.c_int32(), C.c_int32()
    assert lib.load().dpilqr_model_dims(7, C.byref(ns), C.byref(nc)) == 0 and (ns.value, nc.value) == (12, 4)
    assert lib.load().dpilqr_model_dims(8, C.byref(ns), C.byref(nc)) == 0 and (ns.value, nc.value) == (12, 4)   # padded human
    assert lib.load().dpilqr_model_dims(9, C.byref(ns), C.byref(nc)) == lib.EINVAL


def test_errors_are_codes_not_crashes(lib):
    L = lib.load()
    assert L.dpilqr_tile_layout(0, 1, None, None, None) == lib.EINVAL
    assert b"tile_layout" in L.dpilqr_last_error()
    assert L.dpilqr_rollout(None, None, None, None, None, None) == lib.EINVAL
    d = lib.BatchDesc(1, 2, 5, 2, 10, 0, 0.1, 1.0, 200.0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0)   # (n_s,n_c)=(5,2): no such family
    assert L.dpilqr_rollout(C.byref(d), 1, 1, 1, 1, None) == lib.EINVAL
    assert L.dpilqr_solve_workspace_bytes(C.byref(d), 0, 1) > 0


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(lib.DpilqrError):
        lib.require_gpu()
    import numpy as np
    import dpilqr_amd as dp
    with pytest.raises(lib.DpilqrError):      # product code has no CPU fallback
        dp.DoubleIntDynamics4D(0.1, 100)(np.zeros(4), np.zeros(2))


def test_route_switches_are_ignored_without_the_debug_gate(lib):
    """A stray DPILQR_* variable in a production process's environment must not change which kernel serves a batch
    (csrc/launch.hpp: route_env; the library's only getenv is behind DPILQR_DEBUG_ROUTES=1).  Observable without a GPU: the
    solve's workspace is sized for the route -- tile records for the record-fed sweep, none for the fused sweep (DPILQR_NO_FUSED
    flips that), the large-cluster sweep's scratch for DPILQR_FORCE_BIG."""
    import os
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys; sys.path.insert(0, %r)\n"
        "from dpilqr_amd import _lib\n"
        "L = _lib.load()\n"
        # cfg2's hinted descriptor: five DoubleIntDynamics4D agents, planar, shared weights (batch.py packs the hints the same way)
        "d = _lib.BatchDesc(64, 5, 4, 2, 50, 1 | (3 << 8) | (1 << 16) | (1 << 17), 0.1, 1.0, 200.0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0)\n"
        "print(L.dpilqr_solve_workspace_bytes(C.byref(d), 64, 1), _lib.LIB_PATH)\n" % str(ROOT))

    def run(**env):
        e = {k: v for k, v in os.environ.items() if not k.startswith("DPILQR_")}
        e.update(env)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        nbytes, path = out.stdout.split()
        return int(nbytes), path

    plain, path = run()
    assert path.endswith("dpilqr_amd/libdpilqr_hip.so")
    # without the gate: every switch ignored, the binding's library override too
    assert run(DPILQR_NO_FUSED="1") == (plain, path)
    assert run(DPILQR_FORCE_BIG="1") == (plain, path)
    assert run(DPILQR_LIB="/nonexistent/libx.so") == (plain, path)
    assert run(DPILQR_DEBUG_ROUTES="0", DPILQR_NO_FUSED="1") == (plain, path)
    # with it: the record-fed route needs its tile records, the large-cluster route its scratch
    assert run(DPILQR_DEBUG_ROUTES="1", DPILQR_NO_FUSED="1")[0] > plain
    assert run(DPILQR_DEBUG_ROUTES="1", DPILQR_FORCE_BIG="1")[0] != plain


def test_no_environment_reads_outside_the_gate():
    """`grep getenv csrc/` shows one gate: dpilqr_hip.hip's route_env."""
    hits = []
    for f in sorted(p for p in (ROOT / "dpilqr_amd" / "csrc").iterdir() if p.is_file()):
        for i, line in enumerate(f.read_text().splitlines(), 1):
            code_part = line.split("//")[0]
            if "getenv" in code_part:
                hits.append((f.name, i))
    assert hits and all(name == "dpilqr_hip.hip" for name, _ in hits) and len(hits) == 2, hits
    for f in sorted(p for p in (ROOT / "dpilqr_amd" / "csrc").iterdir() if p.is_file()):
        assert "__builtin_trap" not in f.read_text(), f.name       # include/dpilqr_hip.h: "never throws or aborts"
