"""Worker for tests/test_gpu_multi.py::test_rccl_collective_on_the_engine_stream_single_rank: backend nccl (= RCCL) with ONE rank --
the most of the multi-GPU tempering round that a one-GPU box can execute: process-group creation next to libisingmc.so (one HIP
runtime for both), and the round's collective, torch.distributed.all_gather_into_tensor, enqueued on the ENGINE's HIP stream under
torch.cuda.ExternalStream between the measurement kernels and the exchange kernel, with no host synchronisation."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_path = sys.argv[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    from oracle import exact as X
    from pyisingmontecarlo_amd import _capi

    W, H, G = 256, 64, 12
    ea, eb, ej = X.square_lattice_edges(W, H, -1.0)
    g = _capi.Graph(ea, eb, ej)
    seeds = _capi.make_seeds(3, G)
    betas = np.linspace(0.40, 0.48, G)
    ref = _capi.States(g, seeds)
    ref.pt_attach(betas, 0, G, 1, 99)
    st = _capi.States(g, seeds)
    st.pt_attach(betas, 0, G, 1, 99)
    local, gathered = st.pt_buffers()
    stream = st.pt_stream()
    scratch = torch.empty_like(gathered)
    for _ in range(10):
        ref.pt_time_steps(3); ref.pt_measure(); ref.pt_swap()
        st.pt_time_steps(3)
        st.pt_measure()                       # single rank: the energies land in `gathered` directly
        with torch.cuda.stream(stream):       # the collective of the sharded round, on the engine's stream, behind the measurement
            dist.all_gather_into_tensor(scratch, gathered)
            gathered.copy_(scratch)
        st.pt_swap()
    p_ref, r_ref, s_ref = ref.pt_state()
    p, r, s = st.pt_state()
    res = {"perm_equal": bool(np.array_equal(p, p_ref)), "rounds": int(r), "swaps": int(s), "swaps_ref": int(s_ref),
           "states_equal": bool(np.array_equal(st.packed(), ref.packed()))}
    with open(out_path, "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
