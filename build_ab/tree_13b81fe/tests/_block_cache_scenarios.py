"""Scenarios of tests/test_gpu_block_cache.py: call sequences that make a states object hand device blocks back to the
library's block cache (isingmc.hip cached_free) while work it enqueued without a synchronisation may still be running, with a
second object created at once so that the freed block is picked up again.  Run in-process (cache on) and as
`python tests/_block_cache_scenarios.py` with ISINGMC_NO_ALLOC_CACHE=1 (hipMalloc / hipFree: the device-wide synchronisation
the cache removed); the digests must agree.  Each scenario returns {name: ndarray}."""
import hashlib
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _lattice(W, H, j=-1.0, rng=None):
    from oracle import exact as X
    return X.square_lattice_edges(W, H, j, rng)


def reserve_after_enqueued_tempering(capi, W=512, H=256, R=8, T=300):
    """pt_time_steps (enqueue only) -> append: reserve() regrows state, keys, thresholds and the measurement scratch while the
    sweeps may still run; a second container of the OLD size is created at once (it gets the freed blocks)."""
    ea, eb, ej = _lattice(W, H)
    g = capi.Graph(ea, eb, ej)
    seeds = capi.make_seeds(21, 2 * R + 1)
    betas = np.linspace(0.2, 0.9, R)
    a = capi.States(g, seeds[:R])
    a.pt_attach(betas, 0, R, 1, 5)
    a.pt_time_steps(T)                       # enqueued, not awaited
    a.set_betas(None)                        # (append refuses per-replica betas; the ladder stays attached)
    a.append(int(seeds[2 * R]))              # regrow: cap R -> R + R/2
    b = capi.States(g, seeds[R:2 * R])       # same block sizes as the ones `a` just released
    b.do_time_steps(3, 0.5)
    a.synchronize()
    return {"a": a.packed(), "b": b.packed(), "a_e": a.energies(), "b_e": b.energies()}


def halo_regrow_after_enqueued_strips(capi, L=1024, R=4, T=40):
    """strip geometry: pt_time_steps runs the persistent strip kernel (halo granules sized by the container's capacity);
    append grows the capacity, the next strip launch regrows the halo block while the previous launch may still use the old."""
    ea, eb, ej = _lattice(L, L)
    g = capi.Graph(ea, eb, ej)
    seeds = capi.make_seeds(22, R + 2)
    a = capi.States(g, seeds[:R])
    a.pt_attach(np.linspace(0.3, 0.6, R), 0, R, 1, 5)
    a.pt_time_steps(T)
    a.set_betas(None)
    a.append(int(seeds[R]))
    a.append(int(seeds[R + 1]))
    a.do_time_steps(T, 0.44)                 # synchronous: strips again, halo regrown for the new capacity
    return {"a": a.packed(), "a_e": a.energies()}


def packed_append_after_enqueued_tempering(capi, L=24, R=64, T=30):
    """bit-sliced packed container: pt rounds enqueued (sweep, measure, swap), then a replica that opens a NEW group: state, keys,
    counters and the per-group tables are regrown."""
    from oracle import exact as X
    os.environ["ISINGMC_FORCE_PACKED"] = "1"
    try:
        ea, eb, ej = X.cubic_lattice_edges(L, -1.0)
        g = capi.Graph(ea, eb, ej, force_general=True)
        seeds = capi.make_seeds(23, R + 1 + R)
        a = capi.States(g, seeds[:R])
        a.pt_attach(np.linspace(0.1, 0.4, R), 0, R, 1, 7)
        for _ in range(3):
            a.pt_time_steps(T // 3)
            a.pt_measure()
            a.pt_swap()
        perm, rounds, swaps = a.pt_state()   # (synchronises: the ladder state is part of the result)
        a.pt_detach()
        a.append(int(seeds[R]))
        b = capi.States(g, seeds[R + 1:])
        b.do_time_steps(2, 0.2)
        a.do_time_steps(2, 0.2)
    finally:
        del os.environ["ISINGMC_FORCE_PACKED"]
    return {"a": a.states(), "b": b.states(), "perm": perm, "swaps": np.array([rounds, swaps]), "a_e": a.energies()}


def real_path_set_betas_after_enqueued_tempering(capi, W=160, H=128, R=40, T=12):
    """real-coupling container: tempering rounds enqueued, detach, new per-replica betas (the scale table is rewritten), run."""
    ea, eb, _ = _lattice(W, H, 1.0)
    ej = np.random.default_rng(6).normal(size=len(ea))
    g = capi.Graph(ea, eb, ej, nvars=W * H, stable_path=True)
    seeds = capi.make_seeds(24, R)
    a = capi.States(g, seeds)
    a.pt_attach(np.linspace(0.3, 1.5, R), 0, R, 1, 3)
    for _ in range(4):
        a.pt_time_steps(T // 4)
        a.pt_measure()
        a.pt_swap()
    a.pt_detach()
    a.set_betas(np.linspace(1.5, 0.3, R))
    a.do_time_steps(3)
    return {"a": a.states(), "a_e": a.energies()}


def sampling_slabs_grow(capi, W=256, H=128, R=6):
    """a sampling call, then a larger one (sample ring, counters and the pinned host slabs are regrown), then a field lattice
    whose per-replica threshold table is swapped by set_betas between two runs."""
    ea, eb, ej = _lattice(W, H)
    g = capi.Graph(ea, eb, ej)
    a = capi.States(g, capi.make_seeds(25, R))
    e1, s1 = a.run_sampling(0.4, 3, 2, 2)
    e2, s2 = a.run_sampling(0.5, 1, 1, 9)
    b = capi.States(g, capi.make_seeds(26, R))      # takes what `a` released while growing
    e3, s3 = b.run_sampling(0.4, 3, 2, 2)
    gf = capi.Graph(ea, eb, ej, biases=np.full(W * H, 0.5))
    c = capi.States(gf, capi.make_seeds(27, R))
    e4, s4 = c.run_sampling(0.4, 2, 1, 3)
    c.set_betas(np.linspace(0.2, 0.8, R))
    c.do_time_steps(4)
    c.set_betas(np.linspace(0.8, 0.2, R))
    c.do_time_steps(4)
    return {"e1": e1, "s1": s1, "e2": e2, "s2": s2, "e3": e3, "s3": s3, "e4": e4, "s4": s4, "c": c.packed(), "c_e": c.energies()}


def two_threads_create_and_destroy(capi, W=256, H=128, R=5, rounds=12):
    """the device fan-out shape: two host threads create, run and destroy containers of EQUAL sizes on one graph, so each
    thread's freed blocks are the other's next allocation."""
    ea, eb, ej = _lattice(W, H)
    g = capi.Graph(ea, eb, ej)
    seeds = [capi.make_seeds(30 + k, R) for k in range(2)]
    out = [[None] * rounds for _ in range(2)]
    errors = []

    def work(k):
        try:
            for i in range(rounds):
                st = capi.States(g, seeds[k])
                st.do_time_steps(2 + i % 3, 0.3 + 0.1 * k)
                out[k][i] = (st.packed(), st.energies())
                del st
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    res = {}
    for k in range(2):
        for i in range(rounds):
            res[f"t{k}_{i}"] = out[k][i][0]
            res[f"t{k}_{i}_e"] = out[k][i][1]
    return res


SCENARIOS = {f.__name__: f for f in (reserve_after_enqueued_tempering, halo_regrow_after_enqueued_strips,
                                     packed_append_after_enqueued_tempering, real_path_set_betas_after_enqueued_tempering,
                                     sampling_slabs_grow, two_threads_create_and_destroy)}


def digests(result):
    return {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in result.items()}


if __name__ == "__main__":
    from pyisingmontecarlo_amd import _capi
    print(json.dumps({name: digests(fn(_capi)) for name, fn in SCENARIOS.items()}))
