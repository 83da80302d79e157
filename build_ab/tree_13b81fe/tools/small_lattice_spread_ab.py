#!/usr/bin/env python3
"""A/B of the LDS-resident lattice kernel: one lane per quad (ISINGMC_RESIDENT_SPREAD=0) against eight lanes per quad, one Philox
call each (default).  Run once per setting."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyisingmontecarlo_amd import _capi  # noqa: E402
from tools.bench_configs import square  # noqa: E402

for (W, H), R, glass in (((256, 256), 64, False), ((512, 256), 64, False), ((512, 512), 64, False), ((256, 256), 256, False), ((512, 512), 256, False), ((64, 16), 4, False), ((64, 64), 4, False), ((64, 64), 64, False), ((64, 64), 64, True), ((128, 128), 64, False), ((256, 128), 64, False),
                         ((256, 256), 64, False), ((64, 64), 4096, False), ((128, 128), 2048, False),
                         ((64, 64), 256, False), ((64, 64), 512, False), ((64, 64), 1024, False), ((64, 64), 2048, False), ((128, 128), 256, False), ((128, 128), 512, False), ((128, 128), 1024, False), ((256, 128), 256, False), ((256, 128), 512, False)):
    ea, eb, ej = square(W, H, np.random.default_rng(1) if glass else None)
    g = _capi.Graph(ea, eb, ej)
    st = _capi.States(g, _capi.make_seeds(1, R))
    st.do_time_steps(50, 0.44)
    T = 2000 if R <= 64 else 300
    ms = min(st.do_time_steps_timed(T, 0.44) for _ in range(3))
    print(f"spread={os.environ.get('ISINGMC_RESIDENT_SPREAD', '1')} {W:4d}x{H:<4d} x {R:5d} {'+-J' if glass else 'uni'}: {ms / T * 1e3:7.2f} us/step  {R * W * H * T / (ms * 1e-3):.3e} attempts/s", flush=True)
