"""Worker for tests/test_gpu_multi.py: one rank of a world_size-2 group, backend nccl (= RCCL), one GPU per rank.
The on-stream tempering ladder sharded over two GPUs -- its one collective is
torch.distributed.all_gather_into_tensor enqueued on the engine's HIP stream under torch.cuda.ExternalStream --
must reproduce the single-GPU ladder, which rank 0 also runs."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
    from oracle import exact as X
    from pyisingmontecarlo_amd import distributed as D
    from pyisingmontecarlo_amd.tempering import ClassicalTempering

    W, H, G = 256, 64, 12
    edges = X.square_lattice_edges(W, H, -1.0)
    betas = np.linspace(0.40, 0.48, G)

    def ladder(group, device):
        pt = ClassicalTempering(edges, seed=21, group=group, device=device)
        for b in betas:
            pt.add_graph(float(b))
        pt.timesteps(4)
        pt.timesteps(30, replica_swap_freq=3)                      # 10 exchange rounds, all on the stream
        return pt

    pt = ladder(None, rank)                                        # sharded over the default group
    assert pt._on_stream and pt._world == world
    res = {"perm": pt.get_permutation().tolist(), "swaps": pt.get_total_swaps(),
           "energies": pt._states.energies().tolist(), "lo": pt._lo, "hi": pt._hi}
    # the non-lattice path's collective: numpy -> device -> RCCL all-gather -> host
    lo, hi = D.shard_bounds(5, world, rank)
    res["gathered"] = D.all_gather_f64(np.arange(lo, hi, dtype=np.float64) + 0.5, 3).tolist()
    if rank == 0:
        solo_group = dist.new_group([0])
    else:
        solo_group = dist.new_group([0])                           # every rank must call new_group
    if rank == 0:
        solo = ClassicalTempering(edges, seed=21, group=solo_group, device=0)
        for b in betas:
            solo.add_graph(float(b))
        solo.timesteps(4)
        solo.timesteps(30, replica_swap_freq=3)
        res["solo_perm"] = solo.get_permutation().tolist()
        res["solo_swaps"] = solo.get_total_swaps()
        res["solo_energies"] = solo._states.energies().tolist()
    with open(out_path + f".{rank}", "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
