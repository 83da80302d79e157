#!/usr/bin/env python3
"""Headline benchmark: spin-flip attempts/s on the 4096 x 4096 2-d Ising lattice (BASELINE.json c2).

A "step" = one Metropolis timestep (one full sweep: black then white half-sweep) of every replica
resident on the GPU: the `for _ in 0..timesteps { do_time_step(beta) }` loop of
Lattice.run_monte_carlo (reference lattice.rs:204-207) for 256 experiments at once.  Spins are
already in HBM when the timed region starts; edge-list ingest, lattice recognition, the random start
and the final energy/state read-back are outside it.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU)

Replicas are the shard: every rank runs its own 256 replicas (weak scaling), no data-path collective.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L = 4096
BETA = 0.4407
REPLICAS_PER_GPU = 256
SEED_GEN = 1
BYTES_PER_ATTEMPT = 0.375  # read own + other colour plane, write own plane, bit-packed
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def lattice_edges(W, H, J=-1.0):
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


def traffic_from_profile():
    """HBM bytes per sweep-kernel launch from the committed rocprofv3 --pmc passes (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        with open(path) as f:
            return json.load(f).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--replicas", type=int, default=REPLICAS_PER_GPU, help="replicas per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo lets several ranks share one "
                         "GPU to rehearse the multi-rank path on a one-GPU box)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

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

    states.do_time_steps(args.warmup, BETA)
    barrier()
    t0 = time.perf_counter()
    device_ms = states.do_time_steps_timed(args.steps, BETA)  # blocking; HIP events on the engine's stream
    barrier()
    wall = time.perf_counter() - t0

    stats = torch.tensor([wall, device_ms], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    wall, device_ms = float(stats[0]), float(stats[1])

    # sanity outside the timed region: energy per site at beta_c must be near -sqrt(2)
    e_site = float(states.energies().mean()) / nvars

    if rank == 0:
        attempts = world * R * nvars * args.steps
        launches = 2 * args.steps                       # one kernel launch per colour per step
        bytes_per_launch = BYTES_PER_ATTEMPT * R * nvars / 2
        achieved = bytes_per_launch / (device_ms * 1e-3 / launches) / 1e9
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
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_from_profile(),
                         "kernel": "lat_sweep_loop_kernel<uniformJ> (2 quads per thread)",
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "avg_launch_us": device_ms * 1e3 / launches},
            "device_attempts_per_s": R * nvars * args.steps / (device_ms * 1e-3),
            "energy_per_site": e_site,
            "ingest_s": t_ingest,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ea, eb, ej, nvars)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
