"""bench.py's rank launcher (no GPU needed): `python bench.py --gpus N` must start N ranks on its own or fail loudly --
never print a one-GPU line for an N-GPU request (round-3 review: the driver's command shape is plain `python bench.py --gpus N`)."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench(args, env_extra=None, drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "DPILQR_BENCH_BACKEND")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], env=env, capture_output=True, text=True, timeout=300,
                          cwd=str(ROOT))


def test_rank_count_mismatch_is_an_error():
    out = _bench(["--gpus", "2", "--steps", "1"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "--gpus 2 but WORLD_SIZE=4" in out.stderr
    out = _bench(["--gpus", "1", "--steps", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in out.stderr


def test_gpus_n_without_a_launcher_starts_n_ranks_and_relays_their_failure():
    """Here there is no GPU at all, so every rank the launcher starts must refuse (RCCL needs one GPU per rank); what is
    checked is that N ranks WERE started (each names its own rank count), that the launcher's exit code is non-zero and that
    no JSON line came out."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("a multi-GPU box runs the real thing (tests/test_gpu_api.py)")
    out = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--reps", "1", "--no-cpu-baseline"])
    assert out.returncode != 0, out.stdout + out.stderr
    assert "needs 2 visible GPUs" in out.stderr and "exited with code" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_gpus_zero_is_rejected():
    out = _bench(["--gpus", "0"])
    assert out.returncode != 0 and "at least one GPU" in out.stderr
