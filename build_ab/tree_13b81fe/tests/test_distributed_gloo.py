"""N > 1 path on CPU: world_size-2 gloo group; the sharded tempering ladder must reproduce the
single-process run exactly (same energies, same permutation, same swap count)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_sharded_ladder_equals_single_process(tmp_path, exact):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import OracleLatEngine
    from pyisingmontecarlo_amd.tempering import ClassicalTempering

    W, H = 64, 4
    pt = ClassicalTempering(exact.square_lattice_edges(W, H, -1.0), seed=77,
                            engine_factory=lambda: OracleLatEngine(W, H))
    for b in np.linspace(0.40, 0.46, 5):
        pt.add_graph(b)
    pt.timesteps(3)
    states, energies = pt.timesteps_sample(12, replica_swap_freq=2, sampling_freq=4)
    assert states.shape == (5, 3, W * H) and energies.shape == (5,)
    assert pt.get_total_swaps() > 0

    port = _free_port()
    outs = [str(tmp_path / f"rank{r}.json") for r in range(2)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_gloo_worker.py"), str(r), "2",
                               str(port), outs[r]], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    res = [json.load(open(o)) for o in outs]

    for r in res:
        assert r["gathered"] == [0.5, 1.5, 2.5, 3.5, 4.5, 0.0]          # rank-major, tail padded
        np.testing.assert_array_equal(r["energies"], energies)           # identical on every rank
        assert r["perm"] == pt.get_permutation().tolist()
        assert r["swaps"] == pt.get_total_swaps()
    assert (res[0]["lo"], res[0]["hi"], res[1]["lo"], res[1]["hi"]) == (0, 3, 3, 5)
    # configurations stay on their rank; indexed by rung they equal the single-process arrays
    total = sum(r["states_sum"] for r in res)
    assert total == int(states.sum())
    rungs = np.array(res[0]["rungs"] + res[1]["rungs"])                  # slot -> rung at each sample
    for s in range(3):
        assert sorted(rungs[:, s]) == list(range(5))
