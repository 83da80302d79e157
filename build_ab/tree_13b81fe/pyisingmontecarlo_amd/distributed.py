"""One process per GPU: replica sharding over torch.distributed (backend "nccl" = RCCL over xGMI).

Replicas are independent in the reference (one rayon task per experiment, lattice.rs:192-197), so the
sweep itself needs NO collective: rank r owns the contiguous block of experiments
[r*ceil(R/world), ...) and keys every replica's Philox stream by its GLOBAL experiment index, which
makes the results independent of the number of GPUs.  The only exchange on the hot path is the
parallel-tempering swap step (tempering.py): an all-gather of one float64 per replica.
"""
import os

import numpy as np


def shard_bounds(n, world, rank):
    """Contiguous block [lo, hi) of n replicas owned by `rank` (replica r -> GPU r // ceil(R/G))."""
    per = block_size(n, world)
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def block_size(n, world):
    """Replicas per rank: ceil(n / world), rounded up to a multiple of 32 once a rank holds >= 32 of them.
    Results never depend on the cut (isingmc_states_create_range decides everything from the global
    experiment index; a shard that cuts a 32-replica group of the replica-packed path simulates the whole
    group), but aligned blocks avoid that duplicated work, and per-replica betas on the packed path
    (tempering) require them."""
    per = -(-n // world) if world > 0 else n
    return -(-per // 32) * 32 if per >= 32 else per


def _dist():
    import torch.distributed as dist
    return dist


def is_initialized():
    try:
        dist = _dist()
    except ImportError:
        return False
    return dist.is_available() and dist.is_initialized()


def world_rank(group=None):
    if not is_initialized():
        return 1, 0
    dist = _dist()
    return dist.get_world_size(group), dist.get_rank(group)


def init_from_env(backend=None):
    """torchrun entry: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  Binds this
    process to GPU LOCAL_RANK (also exported as ISINGMC_DEVICE for the C++ shim)."""
    import torch
    dist = _dist()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("ISINGMC_DEVICE", str(local_rank))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend)
    return world_rank()


def all_gather_f64(local, per, group=None):
    """All-gather `local` (float64, <= per entries; padded to `per`) -> float64[world * per] on every rank.

    The tempering swap step's one collective: with backend nccl this is an RCCL all-gather of
    per*8 bytes per rank over xGMI (pure latency)."""
    local = np.asarray(local, dtype=np.float64)
    world, _ = world_rank(group)
    buf = np.zeros(per, dtype=np.float64)
    buf[:local.size] = local
    if world == 1:
        return buf
    import torch
    dist = _dist()
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    send = torch.from_numpy(buf).to(dev)
    recv = torch.empty(world * per, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    return recv.cpu().numpy()


def run_monte_carlo(lattice, beta, timesteps, num_experiments, group=None, **kwargs):
    """Lattice.run_monte_carlo (lattice.rs:171-221) with the experiments sharded over the ranks.

    Every rank gets energies float64[num_experiments] (all-gathered) and the bool[local, N] states of
    its own block (4 GiB of states for BASELINE's c2 are not worth shipping over xGMI by default).
    `lattice` is a py_monte_carlo.Lattice; seeds follow its seed_gen, so seed_gen must be set.
    """
    world, rank = world_rank(group)
    lo, hi = shard_bounds(num_experiments, world, rank)
    energies, states = lattice.run_monte_carlo(beta, timesteps, num_experiments, replica_range=(lo, hi), **kwargs)
    per = block_size(num_experiments, world)
    all_e = all_gather_f64(energies, per, group)
    # blocks are contiguous and only the tail ranks can be short: drop the padding
    out = np.concatenate([all_e[r * per:r * per + (shard_bounds(num_experiments, world, r)[1] -
                                                  shard_bounds(num_experiments, world, r)[0])]
                          for r in range(world)])
    return out, states, (lo, hi)
