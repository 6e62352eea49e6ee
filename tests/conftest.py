import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# the library ignores its A/B route switches (DPILQR_BIG_TEAM, DPILQR_FORCE_BIG, ...) unless this gate is open; the tests that
# compare routes need them (csrc/launch.hpp: route_env).  The gate is read once, at the library's first launch.
os.environ.setdefault("DPILQR_DEBUG_ROUTES", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(GOLDEN / f"{name}.npz"))
        return cache[name]

    return load


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    den = max(np.max(np.abs(b)), 1e-300)
    return float(np.max(np.abs(a - b)) / den)
