#!/usr/bin/env python3
"""Time of one energy measurement on the replica-packed path (256^3 cubic lattice through the general path, 64 replicas)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyisingmontecarlo_amd import _capi
from tools.bench_configs import cubic
L, R = 256, 64
g = _capi.Graph(*cubic(L), nvars=L ** 3, force_general=True)
st = _capi.States(g, _capi.make_seeds(1, R))
st.do_time_steps(2, 0.2217)
e = st.energies()
t0 = time.perf_counter()
for _ in range(10):
    e = st.energies()
dt = (time.perf_counter() - t0) / 10
print(f"energies() on the packed path: {dt*1e3:.2f} ms per call; E/N = {e.mean() / L**3:.5f}")
