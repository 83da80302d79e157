#!/usr/bin/env python3
"""Where does a SMALL call spend its time?  c1 (16 x 16, 4 experiments, 1000 timesteps) through the C ABI, phase by phase, and through
the Python surface (Lattice.run_monte_carlo).  The kernel itself takes ~1.8 ms; everything else is fixed cost per call."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi  # noqa: E402
import py_monte_carlo  # noqa: E402

L, R, T = 16, 4, 1000
ids = np.arange(L * L, dtype=np.uint64).reshape(L, L)
ea = np.ascontiguousarray(np.stack([ids, ids], axis=-1).reshape(-1))
eb = np.ascontiguousarray(np.stack([np.roll(ids, -1, axis=1), np.roll(ids, -1, axis=0)], axis=-1).reshape(-1))
ej = -np.ones(len(ea))
g = _capi.Graph(ea, eb, ej)
seeds = _capi.make_seeds(1234, R)
for rep in range(3):
    t = [time.perf_counter()]
    st = _capi.States(g, seeds); t.append(time.perf_counter())
    st.do_time_steps(T, 0.3); t.append(time.perf_counter())
    e = st.energies(); t.append(time.perf_counter())
    s = st.states(); t.append(time.perf_counter())
    del st; t.append(time.perf_counter())
    names = ["create", "1000 timesteps", "energies", "states", "destroy"]
    print("C ABI   " + "  ".join(f"{n} {1e3 * (b - a):.3f} ms" for n, a, b in zip(names, t, t[1:])) + f"  total {1e3 * (t[-1] - t[0]):.3f} ms", flush=True)
lat = py_monte_carlo.Lattice.from_arrays(ea, eb, ej, seed_gen=1234)
lat.run_monte_carlo(0.3, 10, R)
for rep in range(3):
    t0 = time.perf_counter(); lat.run_monte_carlo(0.3, T, R); t1 = time.perf_counter()
    t2 = time.perf_counter(); lat.run_monte_carlo(0.3, 1, R); t3 = time.perf_counter()
    print(f"Python  run_monte_carlo(0.3, {T}, {R}): {1e3 * (t1 - t0):.3f} ms   run_monte_carlo(0.3, 1, {R}): {1e3 * (t3 - t2):.3f} ms", flush=True)
