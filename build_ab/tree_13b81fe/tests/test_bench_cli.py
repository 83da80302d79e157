"""bench.py's launcher logic on a CPU-only box: a bare `--gpus N` (no WORLD_SIZE) must start N ranks itself and
touch no GPU in the parent; here the ranks then stop with the "needs a GPU" message (there is no CPU fallback)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bare_multi_gpu_invocation_spawns_ranks_without_touching_a_gpu():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-box test (the GPU-side twin is tests/test_gpu_multi.py)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--backend", "gloo"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    # the ranks were started (each reports for itself -- but the launcher stops the second one as soon as the first has failed, so
    # only one message is certain) and the parent relayed their failure
    assert out.stderr.count("bench.py needs a GPU") >= 1, out.stderr[-1500:]
    assert "torch.distributed" in out.stderr or "ChildFailedError" in out.stderr or "local_rank" in out.stderr, out.stderr[-1500:]


def test_gpus_must_match_world_size_under_torchrun():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=4" in out.stderr
