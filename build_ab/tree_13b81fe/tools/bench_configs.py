#!/usr/bin/env python3
"""Single-GPU measurements of BASELINE.json's other configurations (c3, c4, c5) -- parity-test cases, not
bench lines; the numbers feed DESIGN.md section 5.  One JSON line per configuration.

  python tools/bench_configs.py [c3] [c4] [c5] [--steps K]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi  # noqa: E402
from pyisingmontecarlo_amd.tempering import ClassicalTempering  # noqa: E402


def _pmc(name):
    """HBM bytes per sweep launch from the committed counter passes (profiles/traffic_<name>.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", f"traffic_{name}.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def square(W, H, rng=None):
    ids = np.arange(W * H, dtype=np.uint64).reshape(H, W)
    ea = np.stack([ids, ids], axis=-1).reshape(-1)
    eb = np.stack([np.roll(ids, -1, axis=1), np.roll(ids, -1, axis=0)], axis=-1).reshape(-1)
    ej = np.full(ea.shape, -1.0) if rng is None else rng.choice(np.array([-1.0, 1.0]), size=ea.shape)
    return np.ascontiguousarray(ea), np.ascontiguousarray(eb), ej


def cubic(L):
    ids = np.arange(L ** 3, dtype=np.uint64).reshape(L, L, L)
    ea = np.stack([ids] * 3, axis=-1).reshape(-1)
    eb = np.stack([np.roll(ids, -1, axis=2), np.roll(ids, -1, axis=1), np.roll(ids, -1, axis=0)], axis=-1).reshape(-1)
    return np.ascontiguousarray(ea), np.ascontiguousarray(eb), np.full(ea.shape, -1.0)


def c4(steps):
    """2048^2 +-J glass, geometric beta schedule 0.1 -> 3.0, 128 replicas (the per-GPU share of 1024)."""
    L, R = 2048, 128
    ea, eb, ej = square(L, L, np.random.default_rng(2024))
    g = _capi.Graph(ea, eb, ej, nvars=L * L)
    assert g.kind == _capi.KIND_LATTICE2D and not g.info.uniform_sign
    st = _capi.States(g, _capi.make_seeds(1, R))
    betas = 0.1 * 30.0 ** (np.arange(steps) / max(steps - 1, 1))
    st.do_time_steps(10, 0.1)
    ms = st.do_time_steps_timed(steps, betas)
    e = st.energies().mean() / L ** 2
    rate = R * L * L * steps / (ms * 1e-3)
    out = {"config": "c4", "lattice": [L, L], "replicas": R, "steps": steps, "attempts_per_s": rate,
           "ms_per_step": ms / steps, "bytes_per_attempt": 0.875,
           "hbm_frac_algorithmic": rate * 0.875 / 8e12, "final_energy_per_site": e,
           "note": "0.875 B/attempt counts the 0.5 B of coupling-sign planes per replica; they are shared by all replicas and "
                   "stay in L2, so the HBM traffic is near 0.375 B/attempt: see hbm_frac_counters"}
    pmc = _pmc("c4")                     # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this script (profiles/)
    if pmc:
        out["pmc_bytes_per_attempt"] = pmc["hbm_bytes_per_launch"] / (R * L * L / 2)
        out["hbm_frac_counters"] = rate * out["pmc_bytes_per_attempt"] / 8e12
    return out


def c3(steps):
    """1024^2, 64 rungs (the per-GPU share of the 512-rung ladder beta = 0.1 .. 1.0), swap every 10 sweeps."""
    L, G = 1024, 64
    pt = ClassicalTempering(square(L, L), seed=1)
    for b in 0.1 + 0.9 * np.arange(G) / 511 * 8:      # every 8th rung of the 512-ladder: same span on one GPU
        pt.add_graph(float(b))
    pt.timesteps(20)
    pt.timesteps(40, replica_swap_freq=10)              # warm-up of the exchange path (its buffers are created on first use)
    t0 = time.perf_counter()
    n_blocks = steps // 10
    pt.timesteps(n_blocks * 10, replica_swap_freq=10)   # sweeps + exchange rounds, all on the engine's stream
    dt = time.perf_counter() - t0
    return {"config": "c3", "lattice": [L, L], "rungs": G, "steps": n_blocks * 10, "swap_every": 10,
            "attempts_per_s": G * L * L * n_blocks * 10 / dt, "ms_per_step": dt * 1e3 / (n_blocks * 10),
            "hbm_frac": G * L * L * n_blocks * 10 / dt * 0.375 / 8e12, "total_swaps": pt.get_total_swaps()}


def c5(steps):
    """256^3 cubic lattice through the general edge-list path (recogniser disabled), 64 replicas."""
    L, R = 256, 64
    ea, eb, ej = cubic(L)
    t0 = time.perf_counter()
    g = _capi.Graph(ea, eb, ej, nvars=L ** 3, force_general=True)
    ingest = time.perf_counter() - t0
    st = _capi.States(g, _capi.make_seeds(1, R))
    st.do_time_steps(2, 0.2217)
    ms = st.do_time_steps_timed(steps, 0.2217)
    rate = R * L ** 3 * steps / (ms * 1e-3)
    # the replica-packed path's OWN algorithmic bytes: per position and group of 32 replicas a class launch reads its word
    # (4 B), writes it (4 B), reads every word of the other class once (4 B) and 6 block headers per 64 positions (0.75 B)
    bpa = (4 + 4 + 4 + 0.75) / 32
    out = {"config": "c5", "lattice": [L, L, L], "replicas": R, "steps": steps, "n_colours": int(g.info.n_colours),
           "attempts_per_s": rate, "ms_per_step": ms / steps, "bytes_per_attempt": bpa, "hbm_frac": rate * bpa / 8e12,
           "csr_bytes_per_attempt": 48.375, "vs_per_replica_csr_ceiling": rate / (8e12 / 48.375),
           "graph_build_s": ingest, "energy_per_site": st.energies().mean() / L ** 3,
           "note": "hbm_frac uses the packed path's own bytes; SURVEY 8d's 48.375 B/attempt is the per-replica CSR stream this "
                   "path avoids (32 replicas share every index): vs_per_replica_csr_ceiling is the speed-up over THAT roofline"}
    pmc = _pmc("c5")
    if pmc:
        out["pmc_bytes_per_attempt"] = pmc["hbm_bytes_per_launch"] / (R * L ** 3 / 2)
        out["hbm_frac_counters"] = rate * out["pmc_bytes_per_attempt"] / 8e12
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a in ("c3", "c4", "c5")]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else None
    for name in args or ["c3", "c4", "c5"]:
        fn = {"c3": c3, "c4": c4, "c5": c5}[name]
        default = {"c3": 1000, "c4": 200, "c5": 20}[name]
        print(json.dumps(fn(steps or default)), flush=True)
