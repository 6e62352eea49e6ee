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
    assert L.dpilqr_abi_version() == 3


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
