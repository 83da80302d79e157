"""Worker for tests/test_distributed_gloo.py: one rank of a world_size-2 gloo group on CPU.
The tempering / sharding HOST logic is the product's; the sweep engine is the CPU oracle stand-in
(tests/helpers.py) because there is no GPU here."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from helpers import OracleLatEngine
    from oracle import exact as X
    from pyisingmontecarlo_amd import distributed as D
    from pyisingmontecarlo_amd.tempering import ClassicalTempering

    res = {}
    # 1. the swap step's collective with an uneven split (5 slots over 2 ranks: 3 + 2, padded to 3)
    lo, hi = D.shard_bounds(5, world, rank)
    gathered = D.all_gather_f64(np.arange(lo, hi, dtype=np.float64) + 0.5, 3)
    res["gathered"] = gathered.tolist()

    # 2. a 5-rung ladder sharded over the ranks
    W, H = 64, 4
    edges = X.square_lattice_edges(W, H, -1.0)
    pt = ClassicalTempering(edges, seed=77, engine_factory=lambda: OracleLatEngine(W, H))
    for b in np.linspace(0.40, 0.46, 5):
        pt.add_graph(b)
    pt.timesteps(3)
    states, energies, rungs = pt.timesteps_sample(12, replica_swap_freq=2, sampling_freq=4)
    res.update(energies=energies.tolist(), perm=pt.get_permutation().tolist(), swaps=pt.get_total_swaps(),
               states_sum=int(states.sum()), rungs=rungs.tolist(), lo=lo, hi=hi,
               local_states=states.astype(np.uint8).reshape(states.shape[0], -1).sum(axis=1).tolist())
    with open(out_path, "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
