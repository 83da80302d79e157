#!/usr/bin/env python3
"""3-d +-J (Edwards-Anderson) spin glass on the replica-packed path: L^3 sites, 64 replicas, one beta -- the one-degree
kernel with sign masks (packed_uni_kernels.hpp, PMJ) against the general packed kernel (ISINGMC_DISABLE_PACKED_UNIFORM=1)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi  # noqa: E402
from tools.bench_configs import cubic  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
R, BETA = 64, 0.9
ea, eb, ej = cubic(L)
for name, j in (("ferromagnet", ej), ("+-J glass", ej * np.random.default_rng(7).choice([-1.0, 1.0], len(ej)))):
    t0 = time.perf_counter()
    g = _capi.Graph(ea, eb, j, nvars=L ** 3)
    build = time.perf_counter() - t0
    st = _capi.States(g, _capi.make_seeds(1, R))
    st.do_time_steps(3, BETA)
    ms = min(st.do_time_steps_timed(steps, BETA) for _ in range(2))
    print(f"{L}^3 {name:12s} packed_degree={g.info.packed_degree} uniform_kernel={'ISINGMC_DISABLE_PACKED_UNIFORM' not in os.environ} "
          f"{R * L ** 3 * steps / (ms * 1e-3):.3e} attempts/s {ms / steps * 1e3:8.1f} us/step graph_build {build:.1f} s "
          f"e/site={st.energies().mean() / L ** 3:.4f}", flush=True)
