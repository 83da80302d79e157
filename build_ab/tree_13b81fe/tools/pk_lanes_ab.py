#!/usr/bin/env python3
"""Stream lanes over the replica groups of the packed paths (ISINGMC_PK_STREAMS=1|2|4): mid-size glasses.  usage: pk_lanes_ab.py [steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi  # noqa: E402
from tools.bench_configs import cubic, square  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(3)
for name, (ea, eb, _), pm in (("32^3 +-J", cubic(32), True), ("64^3 +-J", cubic(64), True), ("128^3 +-J", cubic(128), True), ("32^3 gauss", cubic(32), False),
                              ("64^3 gauss", cubic(64), False), ("512^2 gauss", square(512, 512), False), ("2048^2 gauss", square(2048, 2048), False)):
    n = int(max(ea.max(), eb.max())) + 1
    ej = rng.choice([-1.0, 1.0], size=len(ea)) if pm else rng.normal(size=len(ea))
    g = _capi.Graph(ea, eb, ej, nvars=n, force_general=True)
    for R in (64, 256):
        res = []
        for lanes in ("1", "2", "4"):
            os.environ["ISINGMC_PK_STREAMS"] = lanes
            st = _capi.States(g, _capi.make_seeds(1, R))
            st.do_time_steps(5, 0.6)
            ms = min(st.do_time_steps_timed(steps, 0.6) for _ in range(3))
            res.append(R * n * steps / (ms * 1e-3))
        del os.environ["ISINGMC_PK_STREAMS"]
        print(f"{name:13s} R={R:4d}  lanes 1: {res[0]:.3e}   2: {res[1]:.3e} ({res[1] / res[0]:.2f}x)   4: {res[2]:.3e} ({res[2] / res[0]:.2f}x)", flush=True)
