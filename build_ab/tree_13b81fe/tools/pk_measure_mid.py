#!/usr/bin/env python3
"""energies() on the packed path at mid size (96^3 x 64) and at c5's size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyisingmontecarlo_amd import _capi
from tools.bench_configs import cubic
for L in (48, 96, 256):
    g = _capi.Graph(*cubic(L), nvars=L ** 3, force_general=True)
    st = _capi.States(g, _capi.make_seeds(1, 64)); st.do_time_steps(2, 0.2217); st.energies()
    t0 = time.perf_counter()
    for _ in range(20): e = st.energies()
    print(f"{L}^3 x 64: energies() {(time.perf_counter() - t0) / 20 * 1e6:.0f} us per call")
