#!/usr/bin/env python3
"""Small Gaussian glasses (the size range of parallel-tempering studies): the LDS-resident f64 CSR kernel (one workgroup per
replica, all timesteps in one launch) against the replica-packed real-coupling kernels (one launch per colour class and
timestep) -- where does the selection rule's crossover sit?  usage: real_small.py [steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi  # noqa: E402
from tools.bench_configs import cubic, square  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(7)
cases = [("cubic", L, cubic(L)) for L in (6, 8, 12, 16, 20, 22)] + [("square", L, square(L, L)) for L in (32, 64, 96, 110)]
for kind, L, (ea, eb, _) in cases:
    n = int(max(ea.max(), eb.max())) + 1
    ej = rng.normal(size=len(ea))
    for R in (16, 64, 256, 1024):
        out = []
        for env in ({"ISINGMC_DISABLE_REAL": "1"}, {"ISINGMC_FORCE_REAL": "1"}):
            os.environ.update(env)
            g = _capi.Graph(ea, eb, ej, nvars=n)
            st = _capi.States(g, _capi.make_seeds(1, R))
            st.do_time_steps(5, 0.8)
            ms = min(st.do_time_steps_timed(steps, 0.8) for _ in range(2))
            out.append(R * n * steps / (ms * 1e-3))
            for k in env:
                del os.environ[k]
        print(f"{kind:6s} L={L:3d} sites={n:6d} R={R:5d}  CSR {out[0]:.3e}  real-coupling {out[1]:.3e}  ratio {out[1] / out[0]:5.2f}", flush=True)
