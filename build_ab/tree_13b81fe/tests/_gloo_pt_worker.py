"""Worker for tests/test_gpu_multi.py::test_sharded_c3_ladder_matches_the_in_kernel_exchange: one rank of a world_size-2
gloo group, BOTH ranks on this box's one GPU.  BASELINE c3's per-GPU share (1024^2, 64 rungs, an exchange round every 10
sweeps) sharded 2 x 32 rungs -- one strip launch per round, the energies all-gathered between measurement and exchange --
must reproduce the single-rank ladder whose exchange rounds run INSIDE the strip launch (rank 0 runs that one too)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path, L, G, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    from oracle import exact as X
    from pyisingmontecarlo_amd.tempering import ClassicalTempering

    edges = X.square_lattice_edges(L, L, -1.0)
    betas = np.linspace(0.44000, 0.44030, G)

    def ladder(group):
        pt = ClassicalTempering(edges, seed=5, group=group, device=0)
        for b in betas:
            pt.add_graph(float(b))
        pt.timesteps(2)
        pt.timesteps(T, replica_swap_freq=10)
        return pt

    pt = ladder(None)
    assert pt._on_stream and pt._world == world
    st = pt._states
    res = {"perm": pt.get_permutation().tolist(), "swaps": pt.get_total_swaps(), "energies": st.energies().tolist(),
           "lo": pt._lo, "hi": pt._hi, "checksum": int(st.packed().astype(np.uint64).sum())}
    solo_group = dist.new_group([0])                               # every rank must call new_group
    if rank == 0:
        solo = ladder(solo_group)
        assert solo._world == 1
        ss = solo._states
        res.update(solo_perm=solo.get_permutation().tolist(), solo_swaps=solo.get_total_swaps(),
                   solo_energies=ss.energies().tolist(),
                   solo_checksums=[int(ss.packed()[lo:hi].astype(np.uint64).sum()) for lo, hi in ((0, G // 2), (G // 2, G))])
    with open(out_path + f".{rank}", "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
