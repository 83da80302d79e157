#!/usr/bin/env python3
"""General edge-list path (thread-per-site CSR kernels): Gaussian couplings + site-dependent biases on a 256^3 cubic and a
4096^2 square lattice, and BASELINE c2's lattice with a uniform field forced through it."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi  # noqa: E402
from tools.bench_configs import cubic, square  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(3)
for name, (ea, eb, ej), n in (("256^3 cubic, Gaussian J + biases", cubic(256), 256 ** 3), ("4096^2 square, Gaussian J + biases", square(4096, 4096), 4096 ** 2)):
    j = rng.normal(size=len(ej))
    g = _capi.Graph(ea, eb, j, nvars=n, biases=rng.normal(size=n) * 0.3)
    assert g.kind == _capi.KIND_GENERAL
    st = _capi.States(g, _capi.make_seeds(1, R))
    st.do_time_steps(2, 0.8)
    ms = min(st.do_time_steps_timed(steps, 0.8) for _ in range(2))
    print(f"{name:36s} R={R:4d}  {R * n * steps / (ms * 1e-3):.3e} attempts/s  {ms / steps * 1e3:9.1f} us/step "
          f"e/site={st.energies().mean() / n:.5f}", flush=True)
    del st, g
