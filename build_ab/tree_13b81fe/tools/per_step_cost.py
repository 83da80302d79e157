#!/usr/bin/env python3
"""What energies after every timestep cost on the lattice path (2048^2 +-J x 128 replicas, and 4096^2 x 64)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyisingmontecarlo_amd import _capi
from tools.bench_configs import square
for (L, R, rng) in ((2048, 128, np.random.default_rng(2024)), (4096, 64, None)):
    g = _capi.Graph(*square(L, L, rng), nvars=L * L)
    st = _capi.States(g, _capi.make_seeds(1, R))
    st.do_time_steps(10, 0.5)
    for per_step in (False, True):
        t0 = time.perf_counter(); st.do_time_steps(200, 0.5, per_step_energies=per_step); dt = time.perf_counter() - t0
        print(f"{L}^2 x {R} per_step_energies={per_step}: {dt / 200 * 1e6:.1f} us/step", flush=True)
