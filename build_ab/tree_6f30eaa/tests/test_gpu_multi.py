"""Multi-GPU: the real RCCL collective of the tempering swap step (needs >= 2 GPUs; skips cleanly on a one-GPU box)
and a bare `python bench.py --gpus 2` rehearsed on one GPU with the gloo backend."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _n_gpus():
    import torch
    return torch.cuda.device_count()      # does not initialise the GPU on this image


def test_on_stream_tempering_over_rccl_two_gpus(tmp_path):
    if _n_gpus() < 2:
        pytest.skip("needs two GPUs: the nccl (RCCL) all-gather under torch.cuda.ExternalStream")
    out = str(tmp_path / "res.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_nccl_worker.py"), out]
    assert subprocess.call(cmd, env=env, timeout=900) == 0
    res = [json.load(open(out + f".{r}")) for r in range(2)]
    for r in res:
        assert r["perm"] == res[0]["solo_perm"] and r["swaps"] == res[0]["solo_swaps"] > 0
        assert r["gathered"] == [0.5, 1.5, 2.5, 3.5, 4.5, 0.0]
    assert res[0]["energies"] + res[1]["energies"] == res[0]["solo_energies"]


def test_bare_bench_invocation_spawns_its_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the driver's form): the parent starts the
    ranks itself and rank 0 prints the one JSON line with n_gpus = 2.  gloo backend: both ranks share this box's GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--replicas", "8", "--precondition-s", "0.05", "--backend", "gloo"], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["value"] > 0 and "cpu_baseline" not in rec and -1.6 < rec["energy_per_site"] < 0.0


def test_single_gpu_bench_line_keeps_the_contract():
    """`python bench.py --gpus 1 --steps K --warmup W` (the driver's form): ONE JSON line on stdout with the contract's keys, the
    `roofline` and `cpu_baseline` objects, and numbers that agree with each other (value x ms_per_step, achieved / peak = frac,
    algorithmic bytes per launch = SURVEY 8d's 0.375 B per attempt x the attempts of one colour half-sweep)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"], env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in rec, key
    assert rec["n_gpus"] == 1 and rec["steps"] == 20 and rec["warmup"] == 5 and rec["higher_is_better"] is True
    assert rec["vs_baseline"] is None and rec["dtype"] == "u32" and rec["data"] == "synthetic" and "workload" in rec["config"]
    attempts_per_step = 256 * 4096 * 4096
    assert abs(rec["value"] * rec["ms_per_step"] * 1e-3 / attempts_per_step - 1.0) < 1e-9
    roof = rec["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and 0.2 < roof["frac"] < 1.0
    assert roof["algorithmic_bytes_per_launch"] == 0.375 * attempts_per_step / 2
    assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * roof["achieved"]
    assert roof["traffic"] is None or 0.9 < roof["traffic"] / roof["algorithmic_bytes_per_launch"] < 1.5
    assert 0.2 < roof["copy_ceiling"]["frac"] <= 1.2 and roof["copy_ceiling"]["peak"] > 2000.0
    cpu = rec["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and 1e6 < cpu["value"] < 1e10 and rec["value"] > 100 * cpu["value"]
    assert -1.6 < rec["energy_per_site"] < -1.0


def test_sharded_c3_ladder_matches_the_in_kernel_exchange(tmp_path):
    """the sharded tempering protocol (one strip launch per round + all-gather + exchange kernel; two
    gloo ranks sharing this GPU, 2 x 32 rungs of 1024^2) against the single-rank ladder whose rounds run inside the strip
    launch: same permutation, swap count, energies and configurations."""
    out = str(tmp_path / "res.json")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_gloo_pt_worker.py"), out, "1024", "64", "40"]
    assert subprocess.call(cmd, env=env, timeout=900) == 0
    res = [json.load(open(out + f".{r}")) for r in range(2)]
    for r in res:
        assert r["perm"] == res[0]["solo_perm"] and r["swaps"] == res[0]["solo_swaps"] > 0
    assert res[0]["energies"] + res[1]["energies"] == res[0]["solo_energies"]
    assert [res[0]["checksum"], res[1]["checksum"]] == res[0]["solo_checksums"]


def test_rccl_collective_on_the_engine_stream_single_rank(tmp_path):
    """What one GPU can execute of the RCCL path: a nccl process group beside libisingmc.so and the tempering round's
    all_gather_into_tensor enqueued on the engine's stream (torch.cuda.ExternalStream) between measurement and exchange."""
    out = str(tmp_path / "res.json")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tests", "_nccl_world1_worker.py"), out], env=env, timeout=600) == 0
    res = json.load(open(out))
    assert res["perm_equal"] and res["states_equal"] and res["rounds"] == 10 and res["swaps"] == res["swaps_ref"] > 0
