"""dpilqr_amd/csrc/trig_inline.hpp -- sin / cos / tan written out as one basic block for the twelve-state quadcopter's stage evaluations --
against the device library's sincos() and tan(), bit for bit (scripts/ubench/trig_inline_check.hip: 2^26 arguments per range)."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def dp():
    import dpilqr_amd
    from dpilqr_amd import _lib
    _lib.require_gpu()   # loud failure if the HIP library or the GPU is missing
    return dpilqr_amd


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


@pytest.mark.gpu
def test_quadcopter12_stage_evaluation_small_and_huge_angles(dp):
    """ModelDef<Quadcopter12D>::f takes the inline path where every lane's Euler angles are below 2^30 and the library calls where
    any lane's is not: both against the oracle's libm evaluation, in batches that are all small, all huge, and mixed lane by lane
    (a wavefront with one huge angle takes the fall-back for all its lanes).  theta is kept away from the poles of tan."""
    import numpy as np
    import torch
    from oracle import oracle as orc
    from dpilqr_amd import _lib
    from dpilqr_amd.device import empty, ptr, stream_handle, to_dev
    rng = np.random.default_rng(4242)
    n, model, ns, nc = 4096, 7, 12, 4
    from dpilqr_amd.bbdynamics import Model
    assert Model.Quadcopter12D.value == model
    lib = _lib.load()
    for case in ("small", "huge", "mixed"):
        x = rng.normal(size=(n, ns)); u = rng.normal(size=(n, nc))
        x[:, 4] = rng.uniform(-1.2, 1.2, n)                       # theta: |cos| >= 0.36
        big = {"small": np.zeros(n, bool), "huge": np.ones(n, bool), "mixed": rng.random(n) < 0.02}[case]
        # psi and phi of the chosen lanes beyond the small-argument reduction's limit (2^30 ~ 1.07e9); theta stays small
        x[big, 3] = rng.uniform(-1.0, 1.0, big.sum()) * 10.0 ** rng.uniform(9.1, 12.0, big.sum())
        x[big, 5] = rng.uniform(-1.0, 1.0, big.sum()) * 10.0 ** rng.uniform(9.1, 12.0, big.sum())
        md = to_dev(np.full(n, model), torch.int32)
        f = empty((n, ns)); xd, ud = to_dev(x), to_dev(u)
        _lib.check(lib.dpilqr_model_f(n, ns, ptr(md), ptr(xd), ptr(ud), ptr(f), stream_handle()))
        got = f.cpu().numpy()
        ref = np.stack([orc.model_f(model, x[i], u[i]) for i in range(n)])
        err = np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)
        assert np.isfinite(got).all() and err.max() < 1e-12, (case, float(err.max()))
