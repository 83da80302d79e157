#!/usr/bin/env python3
"""A/B of the LDS-resident CSR kernel with the graph in global memory / staged in LDS (ISINGMC_GEN_STAGE=0 / 1): device time per
timestep of small graphs.  Run once per setting:  ISINGMC_GEN_STAGE=0 python tools/small_graph_stage_ab.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi  # noqa: E402
from tools.bench_configs import cubic, square  # noqa: E402

os.environ["ISINGMC_DISABLE_REAL"] = "1"
os.environ["ISINGMC_DISABLE_PACKED"] = "1"
rng = np.random.default_rng(1)
for name, (ea, eb, ej), R in (("16x16 ferromagnet (c1)", square(16, 16), 4), ("16x16 ferromagnet", square(16, 16), 256), ("32x32 Gaussian", square(32, 32), 64),
                              ("8^3 Gaussian", cubic(8), 64), ("12^3 Gaussian", cubic(12), 64), ("64x64 Gaussian", square(64, 64), 8), ("64x64 Gaussian", square(64, 64), 512),
                              ("16x16 ferromagnet", square(16, 16), 4096), ("32x32 Gaussian", square(32, 32), 1024), ("8^3 Gaussian", cubic(8), 1024),
                              ("12^3 Gaussian", cubic(12), 512), ("64x64 Gaussian", square(64, 64), 1024)):
    if "Gaussian" in name:
        ej = rng.normal(size=len(ea))
    n = int(max(ea.max(), eb.max())) + 1
    g = _capi.Graph(ea, eb, ej, nvars=n, force_general=True)
    st = _capi.States(g, _capi.make_seeds(1, R))
    st.do_time_steps(50, 0.5)
    T = 2000 if R <= 512 else 400
    ms = min(st.do_time_steps_timed(T, 0.5) for _ in range(3))
    print(f"stage={os.environ.get('ISINGMC_GEN_STAGE', 'auto'):4s} {name:24s} R={R:4d}: {ms / T * 1e3:7.2f} us/step  {R * n * T / (ms * 1e-3):.3e} attempts/s", flush=True)
