"""bench.py as its own launcher (plain `python bench.py --gpus N`): what can be checked without a GPU.  The parent
starts N fresh ranks, relays rank 0's line and ends non-zero, promptly, when a rank fails; a request for more GPUs than
the machine has is refused with a message before anything is started.  (Role of `mpirun -n 8` around the reference's
distributed tests, hoomd/md/test/CMakeLists.txt:22-32,66-75.)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _gpus():
    import torch
    return torch.cuda.device_count()


def test_more_gpus_than_the_machine_has_is_refused_before_any_rank_starts():
    n = _gpus() + 7
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n)], capture_output=True, text=True, timeout=120, env=_env())
    assert r.returncode == 2
    assert f"needs {n} GPUs" in r.stderr and "--backend gloo" in r.stderr
    assert r.stdout.strip() == ""
    assert time.time() - t0 < 60


def test_a_failing_rank_ends_the_whole_launch_with_a_nonzero_status():
    if _gpus() > 0:
        import pytest
        pytest.skip("needs a machine without a GPU: there the ranks fail at their first device call")
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--lattice", "12x12x12", "--equil", "10", "--steps", "4",
                        "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 1, r.stderr[-2000:]
    assert "stopping the other ranks" in r.stderr
    assert not any(line.startswith("{") for line in r.stdout.splitlines())
    assert time.time() - t0 < 120


def test_world_size_that_does_not_match_is_an_error():
    env = dict(_env(), WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in r.stderr
