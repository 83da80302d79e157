#!/usr/bin/env python3
"""Few experiments on a big graph: the replica-packed kernels with a mostly empty 32-replica word against the per-replica f64 CSR
kernels -- from how many experiments on is packing worth it?  usage: few_replicas.py [steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi  # noqa: E402
from tools.bench_configs import cubic, square  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(3)
ea3, eb3, ej3 = cubic(128)
ea2, eb2, _ = square(2048, 2048)
cases = [("128^3 uniform J (bit-sliced packed path)", ea3, eb3, ej3, {"ISINGMC_FORCE_PACKED": "1"}, {"ISINGMC_DISABLE_PACKED": "1"}, True),
         ("2048^2 gaussian (real-coupling packed path)", ea2, eb2, rng.normal(size=len(ea2)), {"ISINGMC_FORCE_REAL": "1"}, {"ISINGMC_DISABLE_REAL": "1"}, False)]
for name, ea, eb, ej, on, off, force_general in cases:
    n = int(max(ea.max(), eb.max())) + 1
    for R in (1, 2, 3, 4, 6, 8, 15, 24, 33, 40):
        out = []
        for env in (off, on):
            os.environ.update(env)
            g = _capi.Graph(ea, eb, ej, nvars=n, force_general=force_general)
            st = _capi.States(g, _capi.make_seeds(1, R))
            st.do_time_steps(2, 0.5)
            ms = min(st.do_time_steps_timed(steps, 0.5) for _ in range(2))
            out.append(R * n * steps / (ms * 1e-3))
            for k in env:
                del os.environ[k]
        print(f"{name:46s} R={R:3d}  CSR {out[0]:.3e}  packed {out[1]:.3e}  ratio {out[1] / out[0]:6.2f}", flush=True)
