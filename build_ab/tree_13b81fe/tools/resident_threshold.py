#!/usr/bin/env python3
"""us/step of the CSR path, LDS-resident kernel vs per-colour launches, by graph size (run twice: with and
without ISINGMC_DISABLE_RESIDENT=1)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyisingmontecarlo_amd import _capi
rng = np.random.default_rng(0)
tag = "streaming" if os.environ.get("ISINGMC_DISABLE_RESIDENT") else "resident"
for n in (1000, 4000, 16000, 64000, 200000):
    ea = rng.integers(0, n, 3 * n).astype(np.uint64); eb = rng.integers(0, n, 3 * n).astype(np.uint64); ej = rng.normal(size=3 * n)
    g = _capi.Graph(ea, eb, ej, nvars=n)
    for R in (4, 64):
        st = _capi.States(g, _capi.make_seeds(1, R))
        st.do_time_steps(5, 0.5)
        T = 200 if n <= 16000 else 40
        ms = st.do_time_steps_timed(T, 0.5)
        print(f"{tag} n={n} colours={int(g.info.n_colours)} R={R}: {ms / T * 1e3:9.1f} us/step", flush=True)
