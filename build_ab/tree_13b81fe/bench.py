#!/usr/bin/env python3
"""Headline benchmark: spin-flip attempts/s on the 4096 x 4096 2-d Ising lattice (BASELINE.json c2).

A "step" = one Metropolis timestep (one full sweep: black then white half-sweep) of every replica
resident on the GPU: the `for _ in 0..timesteps { do_time_step(beta) }` loop of
Lattice.run_monte_carlo (reference lattice.rs:204-207) for 256 experiments at once.  Spins are
already in HBM when the timed region starts; edge-list ingest, lattice recognition, the random start
and the final energy/state read-back are outside it.

  python bench.py [--gpus N] [--steps K] [--warmup W]

--gpus N > 1 without WORLD_SIZE in the environment: this process starts N rank processes itself
(torch.distributed.run, rendezvous on 127.0.0.1) BEFORE anything touches a GPU, relays their output and
exits with their code; launched under torchrun (WORLD_SIZE set) it is one of the ranks.  Replicas are the
shard: every rank runs its own 256 replicas (weak scaling), no data-path collective.  Rank 0 prints ONE
JSON line.

Timed region: after `--precondition-s` seconds of the same sweeps (reported as "precondition_s": the chip
settles its clock a few tens of ms after the load step -- a cold 20-step run reads 20-25 % slow, see
"cold_ms_per_step") and W warm-up steps, exactly K steps between barrier + synchronize pairs.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L = 4096
BETA = 0.4407
REPLICAS_PER_GPU = 256
SEED_GEN = 1
BYTES_PER_ATTEMPT = 0.375  # read own + other colour plane, write own plane, bit-packed
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
N_CU, SIMD_PER_CU = 256, 4


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--replicas", type=int, default=REPLICAS_PER_GPU, help="replicas per GPU")
    ap.add_argument("--precondition-s", type=float, default=0.5,
                    help="seconds of untimed sweeps before the counted warm-up (device clock/power settle)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo lets several ranks share one "
                         "GPU to rehearse the multi-rank path on a one-GPU box)")
    return ap.parse_args()


def spawn_ranks(args):
    """Parent of a bare `python bench.py --gpus N`: start the N ranks, touch no GPU (no torch import here)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def lattice_edges(W, H, J=-1.0):
    import numpy as np
    ids = np.arange(W * H, dtype=np.uint64).reshape(H, W)
    ea = np.stack([ids, ids], axis=-1).reshape(-1)
    eb = np.stack([np.roll(ids, -1, axis=1), np.roll(ids, -1, axis=0)], axis=-1).reshape(-1)
    return np.ascontiguousarray(ea), np.ascontiguousarray(eb), np.full(ea.shape, float(J))


def cpu_baseline(ea, eb, ej, nvars):
    """C restatement of the reference algorithm (oracle engine A: random-site sequential Metropolis,
    adjacency list, f64 dE + exp, one replica per thread), timed on this host on a bounded sample."""
    from oracle import oracle as O
    cores = min(16, os.cpu_count() or 1, len(os.sched_getaffinity(0)))  # the GPU box's CPU share
    sweeps = 2
    seeds = O.make_seeds(SEED_GEN, cores)
    sec, attempts = O.ref_bench(ea, eb, ej, nvars, seeds, BETA, sweeps, cores)
    return {"value": attempts / sec, "unit": "spin-flip attempts/s", "cores": cores, "kind": "port",
            "sample": f"{L}x{L} J=-1 beta={BETA}, {cores} replicas (one per thread) x {sweeps} sweep(s), "
                      f"{sec:.1f} s; C restatement of the reference algorithm, not the Rust crate"}


def _profile_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def traffic_from_profile():
    """HBM bytes per colour half-sweep (all replicas) from the committed rocprofv3 --pmc passes, or None.
    NOT measured in this run: counters need their own rocprofv3 passes (tools/profile.sh)."""
    d = _profile_json("traffic_latest.json")
    return d.get("hbm_bytes_per_launch") if d else None


def copy_ceiling_gbs(torch, mib=2048, reps=10):
    """the on-box streaming-copy ceiling, measured in this run (outside the timed region): `mib` MiB from one device
    buffer to another, bytes read + bytes written per second; the better of the runtime's device-to-device copy and an elementwise
    kernel (dst = src + 1 on 32-bit words), both on torch's current stream so that torch events see them."""
    src = torch.ones((mib << 20) // 4, dtype=torch.int32, device="cuda")
    dst = torch.empty_like(src)
    best = 0.0
    for op in (lambda: dst.copy_(src), lambda: torch.add(src, 1, out=dst)):
        for _ in range(3):
            op()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            op()
        b.record()
        b.synchronize()
        best = max(best, 2.0 * (mib << 20) * reps / (a.elapsed_time(b) * 1e-3) / 1e9)
    return best


def valu_bound(avg_launch_us, clock_ghz, quads_per_launch):
    """Secondary roofline: the sweep kernel is bound by vector-ALU issue cycles, not by HBM.
    achieved = VALU-busy SIMD cycles per wave of quads from the SQ counters of the same kernel -- a COMMITTED profile
    (profiles/sq_latest.json: another session, possibly another box), lane-weighted, so an estimate;
    peak = SIMD cycles a wave of quads owns in THIS run = launch time x live shader clock x SIMDs / wave-quads per launch.
    The two terms come from different sessions: the ratio is reported as an estimate and clamped to 1 (a kernel cannot be
    busier than the cycles it owns; raw_ratio keeps what the division gave, saturated says it was clamped)."""
    d = _profile_json("sq_latest.json")
    if not d or not clock_ghz:
        return None
    owned = avg_launch_us * 1e-6 * clock_ghz * 1e9 * N_CU * SIMD_PER_CU / quads_per_launch
    busy = d["valu_busy_cycles_per_quad"]
    raw = busy / owned
    return {"bound": "valu", "achieved": busy, "peak": owned, "unit": "SIMD cycles per wave of 64 quads (8192 spins)",
            "frac": min(1.0, raw), "raw_ratio": raw, "saturated": raw >= 1.0, "kind": "profile-derived estimate",
            "clock_ghz": clock_ghz, "valu_insts_per_quad": d.get("valu_insts_per_quad"),
            "achieved_source": d.get("source"), "peak_source": "this run: HIP-event launch time x live shader clock"}


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))

    import numpy as np  # noqa: F401
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    device = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    from pyisingmontecarlo_amd import _capi

    ea, eb, ej = lattice_edges(L, L)
    nvars = L * L
    R = args.replicas
    # seeds are keyed by GLOBAL replica id: rank r owns experiments [r*R, (r+1)*R) of world*R
    seeds = _capi.make_seeds(SEED_GEN, world * R)[rank * R:(rank + 1) * R]
    t_ingest = time.perf_counter()
    graph = _capi.Graph(ea, eb, ej, nvars=nvars, device=device)
    assert graph.kind == _capi.KIND_LATTICE2D
    states = _capi.States(graph, seeds)
    t_ingest = time.perf_counter() - t_ingest

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # cold: the same K steps straight after the random start (reported, never `value`)
    barrier()
    cold_ms = states.do_time_steps_timed(args.steps, BETA)
    # pre-conditioning: >= precondition_s seconds of the same sweeps on the same state
    t0 = time.perf_counter()
    pre_steps = 0
    while time.perf_counter() - t0 < args.precondition_s:
        states.do_time_steps(50, BETA)
        pre_steps += 50
    precondition_s = time.perf_counter() - t0
    states.do_time_steps(args.warmup, BETA)
    barrier()
    t0 = time.perf_counter()
    device_ms = states.do_time_steps_timed(args.steps, BETA)  # blocking; HIP events on the engine's stream
    barrier()
    wall = time.perf_counter() - t0

    stats = torch.tensor([wall, device_ms, cold_ms], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    wall, device_ms, cold_ms = float(stats[0]), float(stats[1]), float(stats[2])

    # outside the timed region: energy per site at beta_c must be near -sqrt(2); live shader clock under the kernel
    e_site = float(states.energies().mean()) / nvars
    clock_ghz = states.shader_clock_ghz(60, BETA, probe_ms=10.0) if rank == 0 else None

    if rank == 0:
        attempts = world * R * nvars * args.steps
        launches = 2 * args.steps                       # one colour half-sweep of all replicas = one "launch"
        bytes_per_launch = BYTES_PER_ATTEMPT * R * nvars / 2
        avg_launch_us = device_ms * 1e3 / launches
        achieved = bytes_per_launch / (avg_launch_us * 1e-6) / 1e9
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_from_profile(),
                    "traffic_source": "profiles/traffic_latest.json (committed rocprofv3 --pmc passes, not measured in this run)",
                    "kernel": "lat_sweep_loop_kernel<uniformJ> (2 quads per thread; replicas in 2 stream lanes)",
                    "algorithmic_bytes_per_launch": bytes_per_launch,
                    "avg_launch_us": avg_launch_us,
                    "launch": "one colour half-sweep of all replicas (2 kernel dispatches, one per lane)"}
        copy_gbs = copy_ceiling_gbs(torch)
        roofline["copy_ceiling"] = {"achieved": achieved, "peak": copy_gbs, "unit": "GB/s", "frac": achieved / copy_gbs,
                                    "peak_source": "this run: 2 GiB device-to-device copy (better of the runtime copy and an elementwise kernel), read + written bytes per second"}
        sec = valu_bound(avg_launch_us, clock_ghz, R * nvars / 2 / 128 / 64)  # wave-quads: 64 lanes x 128 spins
        if sec:
            roofline["secondary"] = sec
        out = {
            "metric": "spin-flip attempts/s (whole node), 4096^2 2D Ising",
            "value": attempts / wall,
            "unit": "spin-flip attempts/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"c2: {L}x{L} periodic Ising J=-1 beta={BETA}, {R} replicas/GPU, "
                                   "checkerboard Metropolis, bit-packed spins, Philox4x32-10",
                       "replicas_per_gpu": R, "lattice": [L, L], "beta": BETA, "parallelism": f"replicas x{world}"},
            "precondition_s": precondition_s, "precondition_steps": pre_steps,
            "cold_ms_per_step": cold_ms / args.steps,
            "roofline": roofline,
            "device_attempts_per_s": R * nvars * args.steps / (device_ms * 1e-3),
            "shader_clock_ghz": clock_ghz,
            "energy_per_site": e_site,
            "ingest_s": t_ingest,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ea, eb, ej, nvars)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
